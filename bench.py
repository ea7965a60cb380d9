#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native path-space differentiable renderer.

Workload (BASELINE.json configs[1]): cbox 512x512, spp = 64, PathTracer(max_depth=3);
one "step" = one renderC pass + one renderD pass w.r.t. the diffuse albedo of BSDF[0]
(forward mode, K = 3: d image / d (r, g, b) in a single pass) over synthetic data (the bundled
Cornell-box fixture, 12 triangles).  Metric: Mpath-samples/s = camera sample slots evaluated by
both passes / wall seconds / 1e6, inputs resident in HBM, device-synchronised.

Multi-GPU (`--gpus N` under torch.distributed.run): the spp of every pixel are sharded across the
ranks (weak scaling: per-GPU spp fixed at 64, global spp = 64*N) and the image / derivative-image
buffers are summed with ONE RCCL all-reduce per render call.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "psdr-cuda_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel PER GPU")
    ap.add_argument("--max-depth", type=int, default=3)
    ap.add_argument("--scene", default="cbox")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(slots, rays, derivative):
    """SURVEY.md 8(d) wavefront-stream model: 292 B per traced ray + 12 B splat per camera slot;
    renderD adds 88 B per ray + 12 B per slot."""
    b = 292.0 * rays + 12.0 * slots
    if derivative:
        b += 88.0 * rays + 12.0 * slots
    return b


def measured_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_traffic.json, written by tools/summarize_prof.py from separate rocprofv3 --pmc passes);
    bench.py itself cannot read PMC counters."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=os.path.getmtime)
    try:                      # the summary tools/summarize_prof.py wrote last (file times do not survive a checkout)
        latest = os.path.join(ROOT, "profiles", json.load(open(os.path.join(ROOT, "profiles", "latest.json")))["traffic"])
        files = [f for f in files if f != latest] + [latest]
    except Exception:
        pass
    for f in reversed(files):
        try:
            d = json.load(open(f))
            for name, v in d["kernels"].items():
                if "k_camera" in name and kernel_key in name and "rev" not in name:
                    return v["hbm_bytes_per_launch"], v.get("valu_wave_insts_per_launch"), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None, None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP render path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import psdr_cuda
    from psdr_cuda import _abi
    from psdr_cuda.fixtures import scene_path

    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(args.scene), False)
    sc.opts.width = sc.opts.height = args.res
    sc.opts.spp = args.spp * world              # global spp; each rank renders its 1/world share
    sc.opts.sppe = sc.opts.sppse = 0            # albedo has no boundary term
    sc.opts.log_level = 0
    sc.configure()
    integ = psdr_cuda.PathTracer(max_depth=args.max_depth)
    tb = sc.tables(0)
    opts = integ._opts(sc, with_edges=False)
    tangent_sets = []
    for c in range(3):
        t = torch.zeros_like(tb["texels"])
        t[c] = 1.0
        tangent_sets.append([None, t, None, None, None, None, None])

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    kern_ms = {"render_c": [], "render_d": []}
    rays = {}

    def step(record):
        ev[0].record()
        integ._render_c(sc, tb, opts, None)
        ev[1].record()
        rays_c = integ.last_counters
        ev[2].record()
        integ._render_fwd(sc, tb, opts, None, tangent_sets)
        ev[3].record()
        rays_d = integ.last_counters
        if record:
            torch.cuda.synchronize()
            kern_ms["render_c"].append(ev[0].elapsed_time(ev[1]))
            kern_ms["render_d"].append(ev[2].elapsed_time(ev[3]))
            rays["c"], rays["d"] = rays_c, rays_d

    # The first launch of a kernel loads its code object (8 ms) and the next three run 3-8 % slow while the clocks
    # ramp (tools/var_probe.py): a few untimed passes as part of the setup, so that a small --warmup still
    # measures the steady state.  The W warm-up steps of the contract follow.
    for _ in range(6):
        step(False)
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    slots_per_pass = args.res * args.res * args.spp * world          # whole job
    samples = 2.0 * slots_per_pass * args.steps
    value = samples / dt / 1e6

    # roofline of the dominant kernel (rank 0's launch; HIP events on the launch stream)
    local_slots = args.res * args.res * args.spp
    ms_c, ms_d = float(np.mean(kern_ms["render_c"])), float(np.mean(kern_ms["render_d"]))
    dom = "k_camera<Dual<3>> (renderD)" if ms_d >= ms_c else "k_camera<float> (renderC)"
    dom_ms = max(ms_c, ms_d)
    dom_rays = rays["d"][0] if ms_d >= ms_c else rays["c"][0]
    abytes = algorithmic_bytes(local_slots, dom_rays, ms_d >= ms_c)
    achieved = abytes / (dom_ms * 1e-3) / 1e9
    traffic, valu_insts, traffic_src = measured_traffic("Dual<3>" if ms_d >= ms_c else "float, float")
    # the fused kernel keeps the path state in registers, so the stream-model bytes are NOT moved (frac can
    # exceed 1); what binds it is VALU issue: wave-instructions x 4 cycles (wave64 on a 16-lane SIMD) over the
    # SIMD-cycles of the launch (1024 SIMDs at 2.4 GHz), instruction count from the committed PMC summary
    valu_frac = None if not valu_insts else round(valu_insts * 4.0 / (dom_ms * 1e-3 * 2.4e9 * 1024.0), 4)
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_src, "kernel": dom,
                "valu_issue_frac": valu_frac,
                "kernel_ms": round(dom_ms, 4), "rays_per_slot": round(dom_rays / local_slots, 4),
                "algorithmic_bytes_per_launch": abytes,
                "render_c_ms": round(ms_c, 4), "render_d_ms": round(ms_d, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        cspp = 2
        o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=args.max_depth, spp=cspp)
        tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
        tt = torch.zeros_like(tbc["texels"]); tt[0:3] = 1.0
        cores = os.cpu_count() or 1
        oracle.render(tbc, _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=args.max_depth, spp=1), nthreads=cores)
        c0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - c0 < 10.0:
            oracle.render(tbc, o, nthreads=cores)
            oracle.render(tbc, o, mode=1, tangents={"texels": tt}, nthreads=cores)
            reps += 1
        cdt = time.perf_counter() - c0
        cpu = {"value": round(2.0 * args.res * args.res * cspp * reps / cdt / 1e6, 4), "unit": "Mpath-samples/s",
               "cores": cores, "kind": "port",
               "sample": "%d x (renderC + renderD fwd K=1) of the same scene at %dx%d spp=%d, oracle fp32, %d threads"
                         % (reps, args.res, args.res, cspp, cores)}

    # the metric's parity half: gradient rel-L2 of the HIP path against the CPU oracle, same RNG streams,
    # at a size the oracle finishes in a second (tests/ hold the full parity suite)
    grad = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        gres, gspp = 64, 8
        sc2 = psdr_cuda.Scene()
        sc2.load_file(scene_path(args.scene), False)
        sc2.opts.width = sc2.opts.height = gres
        sc2.opts.spp, sc2.opts.sppe, sc2.opts.sppse, sc2.opts.log_level = gspp, 0, 0, 0
        sc2.configure()
        tb2 = sc2.tables(0)
        o2 = integ._opts(sc2, with_edges=False)
        ts2 = []
        for c in range(3):
            t = torch.zeros_like(tb2["texels"]); t[c] = 1.0
            ts2.append([None, t, None, None, None, None, None])
        _, dimgs = integ._render_fwd(sc2, tb2, o2, None, ts2)
        worst = 0.0
        for c in range(3):
            ref = oracle.render(tb2, o2, mode=1, tangents={"texels": ts2[c][1]})[1].reshape(-1)
            got = dimgs[c].cpu().numpy().astype(np.float64)
            worst = max(worst, float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)))
        grad = {"rel_l2": round(worst, 8), "bound": 1e-3,
                "check": "d image / d albedo(r,g,b), %s %dx%d spp=%d PathTracer(max_depth=%d), HIP vs CPU oracle on the same sample streams"
                         % (args.scene, gres, gres, gspp, args.max_depth)}

    if rank == 0:
        out = {
            "metric": "Mpath-samples/s renderC+renderD, cbox 512x512 spp=64; grad rel-L2 vs ref",
            "value": round(value, 3), "unit": "Mpath-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d spp=%d/GPU PathTracer(max_depth=%d) renderC + renderD(fwd, K=3) w.r.t. diffuse albedo"
                                   % (args.scene, args.res, args.res, args.spp, args.max_depth),
                       "triangles": int(tb["num_tris"]), "global_spp": args.spp * world,
                       "parallelism": "spp-shard x%d, one all-reduce per render call" % world},
            "roofline": roofline, "cpu_baseline": cpu, "grad_rel_l2": grad,
        }
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
