#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native path-space differentiable renderer.

Workload (BASELINE.json configs[1]): cbox 512x512, spp = 64, PathTracer(max_depth=3), derivative w.r.t. the diffuse
albedo of BSDF[0].  One "step" is what the reference's harness runs per pass (examples/run_test.py:44-147), THROUGH THE
DROP-IN SURFACE:
    run_orig pass   img  = integrator.renderC(scene)
    run_ad pass     P = FloatD(0); set_requires_gradient(P); reflectance.data = base + P; scene.configure()
                    imgD = integrator.renderD(scene); enoki.forward(P); d = enoki.gradient(imgD)
Metric: Mpath-samples/s = camera sample slots of the two passes / wall seconds / 1e6 (2 * W * H * spp per step),
scene tables resident in HBM, device-synchronised.  `value` is this surface-inclusive rate; the same work as bare
C-ABI launches (psdr_render_c + psdr_render_d_fwd, no Python / torch table chain) is reported beside it as
`kernel_only`, and the reverse-mode variant of the AD pass (loss.backward -> psdr_render_d_rev) as `reverse`.

Multi-GPU (`--gpus N` under torch.distributed.run): the spp of every pixel are sharded across the ranks (weak
scaling: per-GPU spp fixed at 64, global spp = 64 * N) and the image / derivative-image buffers are summed with ONE
RCCL all-reduce per render call.

`--pmc-child` (internal): a short run of the kernel-only launches under `rocprofv3 --pmc`, spawned by the parent to
MEASURE the VALU instruction count and the HBM traffic of the dominant kernel in this very run.

Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "psdr-cuda_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

# MI355X (MI355X_MICROARCH.md): 256 CUs x 4 SIMD-32 units, 2.4 GHz; a wave64 VALU instruction occupies its SIMD for 2 cycles
# (32 lanes per cycle; `v_fma_f32 (wave64) 2 cyc`), i.e. 1024 * 2.4e9 / 2 = 1228.8 G wave-instructions/s = the 157.3 TFLOP/s
# fp32 vector peak; measured by tools/micro/valu_rate.hip (profiles/r03_valu_rate.txt).  HBM3E 8 TB/s.
N_SIMD, CLOCK_HZ, HBM_BPS = 1024, 2.4e9, 8.0e12
VALU_PEAK_WAVE_INSTS_PER_S = N_SIMD * CLOCK_HZ / 2.0          # 1.2288e12 wave-instructions / s
# VALU lane-instructions one primitive test costs (Moeller-Trumbore with SGPR operands, csrc/psdr_device.h tiny_prim_test) and
# the rest of a traced ray's share of its path vertex (hit reconstruction, sampling, shading): DESIGN.md section 3
FLOOR_VALU_PER_PRIM_TEST, FLOOR_VALU_PER_RAY_REST = 31, 150          # plane-form primitive test as the ISA issues it: 20 arithmetic + 6 compare + 5 select
FLOOR_VALU_PER_SLAB_TEST = 16                                        # axis-aligned rectangle in slab form: 6 arithmetic + 4 compare + 5 select + 1 move



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel PER GPU")
    ap.add_argument("--max-depth", type=int, default=3)
    ap.add_argument("--scene", default="cbox")
    ap.add_argument("--config", default="c2", choices=("c2", "c4"),
                    help="c2 (default, the headline): cbox 512x512 spp 64 PER GPU, albedo derivative, weak scaling.  c4: cbox_bunny 1024x1024, "
                         "GLOBAL spp 512 sharded over the ranks (strong scaling), renderC + renderD + enoki.backward w.r.t. the bunny's vertex "
                         "positions and the albedo texels (gradient-buffer all-reduce)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-tree-scenes", action="store_true", help="skip the tree_scenes block (BASELINE configs 3-5 at one GPU's share)")
    ap.add_argument("--tree-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--native-option", action="append", default=[], help=argparse.SUPPRESS)          # developer: name=value, psdr_scene_set_option on the headline scene's handle (A/B runs)
    ap.add_argument("--hip-lib", default=None, help=argparse.SUPPRESS)          # developer: another build of libpsdr_hip.so (tools/build_variant_lib.sh)
    ap.add_argument("--no-c4-strong", action="store_true", help="skip the c4_strong block (BASELINE configs[3]: cbox_bunny 1024^2, global spp 512 sharded over the ranks)")
    return ap.parse_args()


def stream_model_bytes(slots, rays, derivative):
    """SURVEY.md 8(d) wavefront-stream model: 292 B per traced ray + 12 B splat per camera slot; renderD adds 88 B per
    ray + 12 B per slot.  The fused kernels keep the path state in registers and do NOT move these bytes: reported as
    `stream_model_equiv` only."""
    b = 292.0 * rays + 12.0 * slots
    if derivative:
        b += 88.0 * rays + 12.0 * slots
    return b


class Workload:
    """Scene + integrator + the two call sequences (surface / bare C ABI)."""

    def __init__(self, args, world):
        import enoki as ek
        import psdr_cuda
        from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
        from psdr_cuda.fixtures import scene_path
        self.ek, self.FloatD, self.Vector3fD = ek, FloatD, Vector3fD
        sc = psdr_cuda.Scene()
        sc.load_file(scene_path(args.scene), False)
        for kv in getattr(args, "native_option", []):
            k, v = kv.split("=")
            sc.native_options[k] = float(v)
        sc.opts.width = sc.opts.height = args.res
        sc.opts.spp = args.spp * world              # global spp; each rank renders its 1/world share
        sc.opts.sppe = sc.opts.sppse = 0            # albedo has no boundary term
        sc.opts.log_level = 0
        self.sc = sc
        self.refl = sc.param_map["BSDF[0]"].reflectance
        self.base = ek.detach(self.refl.data)
        self.integ = psdr_cuda.PathTracer(max_depth=args.max_depth)
        sc.configure()

    # ---- through the drop-in surface (examples/run_test.py run_orig + run_ad, one pass each)
    def surface_step(self):
        ek = self.ek
        img = self.integ.renderC(self.sc)
        P = self.FloatD(0.)
        ek.set_requires_gradient(P)
        self.refl.data = self.Vector3fD(self.base) + P
        self.sc.configure()
        imgD = self.integ.renderD(self.sc)
        ek.forward(P, free_graph=True)
        return img, imgD, ek.gradient(imgD)

    def surface_reverse_step(self):
        """renderD + a torch loss + enoki.backward (docs/inverse_diff_render.rst): the gradient scatter-add path."""
        ek = self.ek
        r = self.Vector3fD(self.base)
        ek.set_requires_gradient(r)
        self.refl.data = r
        self.sc.configure()
        imgD = self.integ.renderD(self.sc)
        ek.backward(self.FloatD._wrap(imgD.t.sum().reshape(1)))
        return ek.gradient(r)

    def surface_reverse_all_step(self):
        """the same with EVERY gradient the C2 scene offers: albedo, the light's radiance, the vertex positions of a wall (triangle rows through
        the native table chain) and the camera pose -- configure + renderD + enoki.backward, psdr_render_d_rev with all tables requested"""
        ek, sc = self.ek, self.sc
        r = self.Vector3fD(self.base)
        ek.set_requires_gradient(r)
        self.refl.data = r
        mesh = sc.param_map["Mesh[0]"]
        v = self.Vector3fD(ek.detach(mesh.vertex_positions))
        ek.set_requires_gradient(v)
        mesh.vertex_positions = v
        em = sc.m_emitters[0]
        rad = self.Vector3fD(ek.detach(em.radiance))
        ek.set_requires_gradient(rad)
        em.radiance = rad
        cam = sc.m_sensors[0]
        tw = cam._to_world.detach().clone().requires_grad_(True)
        cam._to_world = tw
        sc.configure()
        imgD = self.integ.renderD(sc)
        ek.backward(self.FloatD._wrap(imgD.t.sum().reshape(1)))
        out = (ek.gradient(r), ek.gradient(v), ek.gradient(rad), tw.grad)
        # back to the albedo-only scene of the other steps
        mesh.vertex_positions = self.Vector3fD(ek.detach(v)); em.radiance = self.Vector3fD(ek.detach(rad)); cam._to_world = tw.detach()
        return out

    # ---- the same work as bare C-ABI launches
    def kernel_setup(self, K):
        self.refl.data = self.Vector3fD(self.base)
        self.sc.configure()
        self.tb = self.sc.tables(0)
        self.opts = self.integ._opts(self.sc, with_edges=False)
        self.tsets = []
        for c in range(K):
            t = torch.zeros_like(self.tb["texels"])
            if K == 1:
                t[0:3] = 1.0
            else:
                t[c] = 1.0
            self.tsets.append([None, t, None, None, None, None, None])
        self.adj = torch.ones(self.tb["width"] * self.tb["height"] * 3, device="cuda")

    def kernel_c(self):
        return self.integ._render_c(self.sc, self.tb, self.opts, None)

    def kernel_d(self):
        return self.integ._render_fwd(self.sc, self.tb, self.opts, None, self.tsets)

    def kernel_rev(self, names=("texels",)):
        tb = dict(self.tb)
        for n in names:
            tb[n] = tb[n].detach().requires_grad_(True)
        return self.integ._render_rev(self.sc, tb, self.opts, None, self.adj)

    def kernel_rev_all(self):
        """every gradient table of the interior term at once: texels, emitter radiance, triangle rows (geometry), camera pose"""
        return self.kernel_rev(("texels", "emitter_rad", "tri_info", "cam_to_world"))

    def kernel_c_keep(self):
        """psdr_render_c with PSDR_FLAG_KEEP_RECORDS: the primal render as the value kernel of the reverse launch that follows (what renderD does when a geometry gradient is attached)"""
        return self.integ._render_c(self.sc, self.tb, self.opts, None, keep_records=True)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(n):
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.mean(ms))


TORCHRUN_VARS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                 "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
                 "PSDR_FORCE_COLLECTIVES", "PSDR_BENCH_ONE_GPU")


def child_env(local_rank=0):
    """Environment of a counter-pass child (rocprofv3 -- python bench.py --*-child): ONE process on this rank's GPU, whatever launched the parent -- the
    torchrun variables are stripped (the child must not try to join the job's process group) and the device is pinned."""
    env = dict(os.environ, TMPDIR="/tmp")
    for k in TORCHRUN_VARS:
        env.pop(k, None)
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    if vis:
        ids = [x for x in vis.split(",") if x != ""]
        env["HIP_VISIBLE_DEVICES"] = ids[local_rank] if local_rank < len(ids) else ids[0]
    else:
        env["HIP_VISIBLE_DEVICES"] = str(local_rank)
    env.pop("CUDA_VISIBLE_DEVICES", None)
    return env


def pmc_passes(args):
    """VALU wave-instructions and HBM bytes per launch of the dominant kernels, measured NOW: this script re-runs
    itself (--pmc-child: a few kernel-only launches) under `rocprofv3 --pmc`, one pass per counter group, and parses
    the counter CSV.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128 B request, so
    reads are doubled (MI355X_MICROARCH.md, HBM section).  Returns {} when rocprofv3 is unavailable."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return {}
    out = {}
    tmp = tempfile.mkdtemp(prefix="psdr_pmc_", dir="/tmp")
    env = child_env(getattr(args, "local_rank", 0))
    try:
        for group in ("SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, group.split()[0])
            cmd = [exe, "--pmc"] + group.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--res", str(args.res), "--spp", str(args.spp), "--max-depth", str(args.max_depth), "--scene", args.scene] + lib_arg(args)
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            except Exception:
                return {}
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {}
            acc = {}
            for r in csv.DictReader(open(files[0])):
                name = r.get("Kernel_Name", "")
                if "k_camera" not in name or r.get("Counter_Name") not in group.split():
                    continue
                # renderD forward = the dual-number kernel or the log-derivative kernel: both are launched behind one gate and one of them returns at once
                key = "rev" if "k_camera_rev" in name else ("d" if ("Dual<" in name or "k_camera_logd" in name) else "c")
                acc.setdefault(key, {}).setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            lead = group.split()[0]
            for key, by_name in acc.items():
                name = max(by_name, key=lambda nm: float(np.mean(by_name[nm].get(lead, [0.0]))))          # the one that did the work
                for cname, vals in by_name[name].items():
                    out.setdefault(key, {})[cname] = float(np.mean(vals))
                out[key]["_kernel"] = name.replace("void (anonymous namespace)::", "").split("(")[0]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def pmc_passes_c4(args, res, spp):
    """Stream-traffic accounting of the wavefront stages of --config c4 (the one place the north star's HBM roofline applies): one renderC of
    the workload per rocprofv3 pass (--kernel-trace for the durations), FETCH_SIZE / WRITE_SIZE per kernel summed over its launches."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tmp = tempfile.mkdtemp(prefix="psdr_pmc4_", dir="/tmp")
    env = child_env(getattr(args, "local_rank", 0))
    cnt, dur, launches = {}, {}, {}
    try:
        for group in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, group)
            cmd = [exe, "--pmc", group, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--config", "c4", "--pmc-child", "--res", str(res), "--spp", str(spp), "--max-depth", str(args.max_depth)] + lib_arg(args)
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, check=True)
            except Exception:
                return None
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    n = r.get("Kernel_Name", "")
                    if "k_wf_" in n and r.get("Counter_Name") == group:
                        k = "k_wf_camera" if "k_wf_camera" in n else ("k_wf_trace" if "k_wf_trace" in n else "k_wf_bounce")
                        cnt.setdefault(k, {}).setdefault(group, 0.0); cnt[k][group] += float(r["Counter_Value"])
                        if group == "FETCH_SIZE":
                            launches[k] = launches.get(k, 0) + 1
            if group == "FETCH_SIZE":
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        n = r.get("Kernel_Name", "")
                        if "k_wf_" in n:
                            k = "k_wf_camera" if "k_wf_camera" in n else ("k_wf_trace" if "k_wf_trace" in n else "k_wf_bounce")
                            dur[k] = dur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k, c in cnt.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and dur.get(k):
            b = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            out[k] = {"launches": launches.get(k), "seconds": round(dur[k], 6), "hbm_bytes": b, "GBps": round(b / dur[k] / 1e9, 1), "hbm_measured_frac": round(b / dur[k] / HBM_BPS, 4)}
    if out:
        tb, ts = sum(v["hbm_bytes"] for v in out.values()), sum(v["seconds"] for v in out.values())
        out["all_wavefront_kernels"] = {"seconds": round(ts, 6), "hbm_bytes": tb, "GBps": round(tb / ts / 1e9, 1), "hbm_measured_frac": round(tb / ts / HBM_BPS, 4)}
    return out or None


# ---------------------------------------------------------------------------------------------------------------- tree scenes
# BASELINE configs 3-5 (the scenes WITH a tree) at one GPU's share, through the same C-ABI entry points as the headline's kernel_only block:
#   c4_shard_path3_renderC      cbox_bunny 1024^2, 64 of the 512 spp (one of eight GPUs), PathTracer(3) renderC
#   c4_shard_direct_rev3        the same shard, DirectIntegrator(1,1) renderD + backward with all three terms (spp = sppe = sppse share), triangle rows + texels
#   c4_shard_path3_rev          the same shard, PathTracer(3) renderD + backward (interior term), triangle rows + texels
#   c4_shard_path3_primal_then_rev   psdr_render_c(PSDR_FLAG_KEEP_RECORDS) + psdr_render_d_rev: one optimisation step's kernels (the reverse call reuses the records; its rays = 0)
#   c4_shard_path3_fwd_geo, c5_path3_fwd_geo   PathTracer(3) renderD forward, K = 1 geometry tangents (a translation of one object): the traced wavefront with dual-number stages
#   c5_path3_renderC            50 k-triangle interior with rough conductors, 512^2 spp 16, PathTracer(3) renderC
#   c3_direct_fwd3              cbox_bunny 512^2 spp = sppe = sppse = 16, renderD forward (K = 1: a translation of the bunny), three terms -- 1/8 of configs[2]'s sample count, kept for continuity
#   c3_bunny_rev3               BASELINE configs[2] AS WORDED: bunny_light 512^2, spp = sppe = sppse = 128, THROUGH THE SURFACE: configure + DirectIntegrator(1,1).renderD +
#                               enoki.backward to ALL 2 503 x 3 vertex positions of the bunny (primal launch + three-term reverse launch + table-chain backward)
#   c3_bunny_fwd3_translation   the same scene and counts, the harness' forward mode (run_test.py:126-129): P = FloatD(0), set_transform(translate(x) * P), configure, renderD, enoki.forward
#   c5_rev_rough_vertices       BASELINE configs[4]'s gradient: the 50 k-triangle interior 512^2 spp 16, PathTracer(3) reverse w.r.t. the texel pool (every roughness / albedo texel)
#                               and the triangle rows (vertices), psdr_render_d_rev
class TreeScenes:
    def __init__(self):
        import psdr_cuda
        from psdr_cuda import _abi
        from psdr_cuda.fixtures import make_interior_scene, scene_path
        self._abi = _abi
        self.integ = psdr_cuda.DirectIntegrator(1, 1)          # only its native plumbing is used: the options below name the integrator
        self.cases = []
        self.size = {}                                         # row -> the workload's size in words (printed in the row)
        self.counters_of = {}                                  # row -> the integrator object whose counters the row reads (default self.integ)

        def bunny(res, spp, sppe, sppse):
            sc = psdr_cuda.Scene()
            sc.load_file(scene_path("cbox_bunny"), False)
            sc.opts.width = sc.opts.height = res
            sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
            sc.configure()
            return sc
        sc4 = bunny(1024, 512, 512, 512)
        tb4 = sc4.tables(0)
        n4 = 1024 * 1024 * 64
        o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=512, spp_range=(0, 64))
        self.cases.append(("c4_shard_path3_renderC", n4, lambda sc=sc4, tb=tb4, o=o: self.integ._render_c(sc, tb, o, None), sc4))
        od = _abi.make_opts(spp=512, sppe=512, sppse=512, spp_range=(0, 64), sppe_range=(0, 64), sppse_range=(0, 64))
        adj4 = torch.ones(1024 * 1024 * 3, device="cuda")
        tb4g = dict(tb4)
        for k in ("tri_info", "texels", "prim_edge", "sec_edge"):
            if tb4g.get(k) is not None:
                tb4g[k] = tb4g[k].detach().requires_grad_(True)
        self.cases.append(("c4_shard_direct_rev3", 3 * n4, lambda sc=sc4, tb=tb4g, o=od, a=adj4: self.integ._render_rev(sc, tb, o, None, a), sc4))
        tb4p = dict(tb4)
        for k in ("tri_info", "texels"):
            tb4p[k] = tb4p[k].detach().requires_grad_(True)
        self.cases.append(("c4_shard_path3_rev", n4, lambda sc=sc4, tb=tb4p, o=o, a=adj4: self.integ._render_rev(sc, tb, o, None, a), sc4))
        # what renderD + enoki.backward costs per optimisation step: the primal render (the loss needs the image) with PSDR_FLAG_KEEP_RECORDS, then the reverse
        # call, which finds the value sweep's records on the handle and runs its adjoint kernel only (round 5)
        self.path_integ = psdr_cuda.PathTracer(3)
        self.cases.append(("c4_shard_path3_primal_then_rev", n4, lambda sc=sc4, tb=tb4p, o=o, a=adj4: (self.path_integ._render_c(sc, tb, o, None, keep_records=True),
                                                                                                      self.path_integ._render_rev(sc, tb, o, None, a)), sc4))
        # forward mode with GEOMETRY tangents (a translation of the bunny: the reference harness' AD mode, run_test.py:126-129) through the PathTracer
        tan4 = self._translation_tangents(sc4, tb4)
        tb4t = sc4.tables(0)
        self.cases.append(("c4_shard_path3_fwd_geo", n4, lambda sc=sc4, tb=tb4t, o=o, t=tan4: self.integ._render_fwd(sc, tb, o, None, [t]), sc4))
        sc5 = make_interior_scene(seed=0, n_objects=10, res=512, spp=16)
        sc5.configure()
        tb5 = sc5.tables(0)
        o5 = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16)
        self.cases.append(("c5_path3_renderC", 512 * 512 * 16, lambda sc=sc5, tb=tb5, o=o5: self.integ._render_c(sc, tb, o, None), sc5))
        tan5 = self._translation_tangents(sc5, tb5, mesh=sc5.m_meshes[8])
        tb5t = sc5.tables(0)
        self.cases.append(("c5_path3_fwd_geo", 512 * 512 * 16, lambda sc=sc5, tb=tb5t, o=o5, t=tan5: self.integ._render_fwd(sc, tb, o, None, [t]), sc5))
        # C3: forward mode, the tangent tables of a unit translation of the bunny along x (every table row that moves with the mesh)
        sc3 = bunny(512, 16, 16, 16)
        tb3 = sc3.tables(0)
        o3 = _abi.make_opts(spp=16, sppe=16, sppse=16)
        tan3 = self._translation_tangents(sc3, tb3)
        tb3 = sc3.tables(0)                                     # the tables of the configure() that carries P (same values)
        self.cases.append(("c3_direct_fwd3", 3 * 512 * 512 * 16, lambda sc=sc3, tb=tb3, o=o3, t=tan3: self.integ._render_fwd(sc, tb, o, None, [t]), sc3))
        self.size["c3_direct_fwd3"] = "cbox_bunny 512x512 spp = sppe = sppse = 16 (1/8 of BASELINE configs[2]'s sample count; another scene), forward K = 1, C-ABI launch"
        self.size["c4_shard_path3_renderC"] = self.size["c4_shard_path3_rev"] = self.size["c4_shard_path3_primal_then_rev"] = self.size["c4_shard_path3_fwd_geo"] = \
            "cbox_bunny 1024x1024, samples [0, 64) of the global 512 spp (one of eight GPUs), PathTracer(3), C-ABI launch"
        self.size["c4_shard_direct_rev3"] = "cbox_bunny 1024x1024, samples [0, 64) of the global spp = sppe = sppse = 512, DirectIntegrator(1,1) three terms, C-ABI launch"
        self.size["c5_path3_renderC"] = self.size["c5_path3_fwd_geo"] = "50 k-triangle interior (10 bunnies, rough conductors) 512x512 spp 16, PathTracer(3), C-ABI launch"
        # ---- BASELINE configs[2] as worded, through the surface (examples/config.py:111-126 sizes; docs/inverse_diff_render.rst reverse mode)
        import enoki as ek
        from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
        sc3f = psdr_cuda.Scene()
        sc3f.load_file(scene_path("bunny_light"), False)
        sc3f.opts.width = sc3f.opts.height = 512
        sc3f.opts.spp, sc3f.opts.sppe, sc3f.opts.sppse, sc3f.opts.log_level = 128, 128, 128, 0
        m3 = sc3f.param_map["Mesh[0]"]
        v3 = ek.detach(m3.vertex_positions)
        sc3f.configure()
        self.integ_c3 = psdr_cuda.DirectIntegrator(1, 1)
        n3 = 512 * 512 * 128

        def c3_rev():
            v = Vector3fD(v3); ek.set_requires_gradient(v); m3.vertex_positions = v
            sc3f.configure()
            imgD = self.integ_c3.renderD(sc3f)
            ek.backward(FloatD._wrap(imgD.t.sum().reshape(1)))
            g = ek.gradient(v)
            m3.vertex_positions = Vector3fD(v3)
            return g
        self.cases.append(("c3_bunny_rev3", 3 * n3, c3_rev, sc3f))
        self.counters_of["c3_bunny_rev3"] = self.integ_c3
        self.size["c3_bunny_rev3"] = ("bunny_light 512x512 spp = sppe = sppse = 128 (BASELINE configs[2]), through the surface: configure + DirectIntegrator(1,1).renderD + enoki.backward "
                                      "to all %d x 3 vertex positions of Mesh[0] (primal launch, three-term psdr_render_d_rev, table-chain backward)" % int(v3.t.shape[0]))

        def c3_fwd():
            P = FloatD(0.); ek.set_requires_gradient(P)
            m3.set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P))
            sc3f.configure()
            imgD = self.integ_c3.renderD(sc3f)
            ek.forward(P, free_graph=True)
            g = ek.gradient(imgD)
            m3.set_transform(np.eye(4, dtype=np.float32))          # the other row's configure() must not see P
            return g
        self.cases.append(("c3_bunny_fwd3_translation", 3 * n3, c3_fwd, sc3f))
        self.counters_of["c3_bunny_fwd3_translation"] = self.integ_c3
        self.size["c3_bunny_fwd3_translation"] = ("bunny_light 512x512 spp = sppe = sppse = 128, through the surface: the harness' forward mode (run_test.py:126-129) -- P = FloatD(0), "
                                                  "Mesh[0].set_transform(translate(x) * P), configure, renderD, enoki.forward: ONE three-term forward-mode launch (K = 1)")
        # ---- BASELINE configs[4]'s gradient: roughness + vertices, reverse mode
        tb5g = dict(sc5.tables(0))
        for k in ("tri_info", "texels"):
            tb5g[k] = tb5g[k].detach().requires_grad_(True)
        adj5 = torch.ones(512 * 512 * 3, device="cuda")
        self.cases.append(("c5_rev_rough_vertices", 512 * 512 * 16, lambda sc=sc5, tb=tb5g, o=o5, a=adj5: self.integ._render_rev(sc, tb, o, None, a), sc5))
        self.size["c5_rev_rough_vertices"] = ("50 k-triangle interior 512x512 spp 16, PathTracer(3) renderD reverse (psdr_render_d_rev) w.r.t. the texel pool (%d words: every roughness and "
                                              "albedo) and all %d triangle rows (vertices)" % (int(tb5g["texels"].numel()), int(tb5g["num_tris"])))

    def _translation_tangents(self, sc, tb, mesh=None):
        """d table / d P for Mesh[1] (the bunny; or `mesh`) translated by P along x: JVP of the table chain through a second configure."""
        import enoki as ek
        from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
        P = FloatD(0.)
        ek.set_requires_gradient(P)
        (mesh if mesh is not None else sc.param_map["Mesh[1]"]).set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P))
        sc.configure()
        tbd = sc.tables(0)
        from enoki._array import _jvp_wrt
        from psdr_cuda import _abi
        return [None if t is None else t.detach() for t in _jvp_wrt([tbd.get(k) for k in _abi.TANGENT_FIELDS], P.t)]

    def run(self, name, fn, scene):
        fn(); torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        rays = self.counters_of.get(name, self.integ).last_counters
        if "primal_then_rev" in name:
            rays = self.path_integ.last_counters              # of the reverse call: 0 rays when it reused the primal render's records
        return sorted(ms)[1], int(rays[0])


def tree_child():
    """--tree-child (under rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace): every workload once, a spin kernel between them as a separator."""
    ts = TreeScenes()
    for name, slots, fn, sc in ts.cases:
        fn(); torch.cuda.synchronize()
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    for name, slots, fn, sc in ts.cases:
        fn(); torch.cuda.synchronize()
        torch.cuda._sleep(1000); torch.cuda.synchronize()


def tree_scenes(args):
    ts = TreeScenes()
    out = {}
    for name, slots, fn, sc in ts.cases:
        ms, rays = ts.run(name, fn, sc)
        out[name] = {"size": ts.size.get(name), "ms": round(ms, 3), "slots": slots, "rays": rays, "Grays_per_s": round(rays / ms / 1e6, 2), "Mslots_per_s": round(slots / ms / 1e3, 1)}
        try:
            # floor of the call's arithmetic: every ray tests the scene's kernel-argument primitives once and pays its share of a path vertex; the tree walks are NOT
            # in the floor (no closed form), so floor_time_frac is a LOWER bound of (time the arithmetic needs at the VALU peak) / (time taken)
            st = ts._abi.scene_stats(sc._native)
            n_prims, n_slab = int(st.get("n_tiny", 0)), int(st.get("n_slab", 0))
            floor_lane = (n_slab * FLOOR_VALU_PER_SLAB_TEST + (n_prims - n_slab) * FLOOR_VALU_PER_PRIM_TEST + FLOOR_VALU_PER_RAY_REST) * float(rays)
            out[name]["floor_time_frac_lower_bound"] = round(floor_lane / 64.0 / VALU_PEAK_WAVE_INSTS_PER_S / (ms * 1e-3), 4) if rays else None
        except Exception:
            out[name]["floor_time_frac_lower_bound"] = None
    exe = shutil.which("rocprofv3")
    if exe and not args.no_pmc:
        tmp = tempfile.mkdtemp(prefix="psdr_tree_", dir="/tmp")
        try:
            per_case = {}                                         # case -> kernel -> [duration ns, {counter: value}, launches]
            for group in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, group)
                cmd = [exe, "--pmc", group, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--tree-child"] + lib_arg(args)
                subprocess.run(cmd, cwd="/tmp", env=child_env(getattr(args, "local_rank", 0)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, check=True)
                disp = {}
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        disp[r["Dispatch_Id"]] = [r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), 0.0]
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r.get("Counter_Name") == group and r["Dispatch_Id"] in disp:
                            disp[r["Dispatch_Id"]][3] += float(r["Counter_Value"])
                rows = sorted(disp.values(), key=lambda x: x[1])
                seg, segs = [], []
                for r in rows:
                    if "spin_kernel" in r[0]:
                        segs.append(seg); seg = []
                    else:
                        seg.append(r)
                segs = segs[1:]                                     # [0] = the warm-up calls in front of the first separator
                for (name, _, _, _), sg in zip(ts.cases, segs):
                    agg = per_case.setdefault(name, {})
                    for kn, _, dd, v in sg:
                        if "k_" in kn and "at::native" not in kn and "rocprim" not in kn:
                            a = agg.setdefault(kn, [0.0, {}, 0])
                            a[1][group] = a[1].get(group, 0.0) + v
                            if group == "SQ_INSTS_VALU":
                                a[0] += dd; a[2] += 1
            for name, agg in per_case.items():
                if not agg:
                    continue
                kn, (dd, c, n) = max(agg.items(), key=lambda kv: kv[1][0])
                short = kn.replace("void (anonymous namespace)::", "").split("(")[0]
                v = c.get("SQ_INSTS_VALU", 0.0)
                out[name]["dominant_kernel"] = {"name": short, "launches": n, "ms_under_profiler": round(dd / 1e6, 3), "valu_wave_insts": v,
                                                "valu_issue_frac": round(v / (dd * 1e-9) / VALU_PEAK_WAVE_INSTS_PER_S, 4) if dd else None}
                # measured HBM traffic of ALL library kernels of the call against 8 TB/s: the north star's roofline for the wavefront launches
                tot_ns = sum(a[0] for a in agg.values())
                if tot_ns and all("FETCH_SIZE" in a[1] and "WRITE_SIZE" in a[1] for a in agg.values()):
                    tot_b = sum((2.0 * a[1]["FETCH_SIZE"] + a[1]["WRITE_SIZE"]) * 1024.0 for a in agg.values())
                    out[name]["hbm"] = {"bytes_per_call": tot_b, "kernel_ms_under_profiler": round(tot_ns / 1e6, 3), "GBps": round(tot_b / (tot_ns * 1e-9) / 1e9, 1),
                                        "hbm_measured_frac": round(tot_b / (tot_ns * 1e-9) / HBM_BPS, 4),
                                        "per_kernel": {k.replace("void (anonymous namespace)::", "").split("(")[0]: {"launches": a[2], "ms": round(a[0] / 1e6, 3),
                                                       "hbm_measured_frac": round((2.0 * a[1]["FETCH_SIZE"] + a[1]["WRITE_SIZE"]) * 1024.0 / (a[0] * 1e-9) / HBM_BPS, 4) if a[0] else None}
                                                       for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:6]}}
        except Exception as e:
            out["pmc_error"] = repr(e)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    out["note"] = ("ONE GPU, median of 3 (HIP events); rows are C-ABI launches unless `size` says 'through the surface'; Grays_per_s = rays traced / time; floor_time_frac_lower_bound = "
                   "(primitive tests + 150 VALU per ray, tree walks not counted) / 64 / VALU peak / time; dominant_kernel = the kernel with the largest summed duration of "
                   "the workload, its SQ_INSTS_VALU over its duration against 1228.8 G wave-instructions/s; hbm = (2 * FETCH_SIZE + WRITE_SIZE) KiB of all library kernels of the call over their "
                   "summed duration against 8 TB/s (rocprofv3 --pmc passes of this script, one per counter)")
    return out


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one process per GPU (what the driver's own
    command line does for N > 1).  PSDR_BENCH_ONE_GPU=1 (developer switch) lets the N ranks share cuda:0 over gloo."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("PSDR_BENCH_ONE_GPU") != "1":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, n_dev))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def rank_devices(dist, local_rank):
    """[(rank, device index, device name, PCI bus id)] of every rank + the RCCL version (rank 0 prints them in `config`)."""
    props = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": dist.get_rank() if dist else 0, "device": local_rank, "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
    if not dist:
        return [mine], None
    allr = [None] * dist.get_world_size()
    dist.all_gather_object(allr, mine)
    try:
        v = torch.cuda.nccl.version()
        v = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        v = None
    return allr, v


def run_c4(args, world, rank, local_rank, dist, devices, rccl):
    """BASELINE configs[3]: cbox_bunny 1024x1024, GLOBAL spp 512 sharded over the ranks (strong scaling), PathTracer(3).  Step = renderC +
    [configure + renderD + enoki.backward] w.r.t. the bunny's vertex positions and the albedo texels: per step one image all-reduce
    (renderC), one of the primal image of renderD and one of the flat gradient buffer [triangle rows || texels]."""
    import enoki as ek
    import psdr_cuda
    from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
    from psdr_cuda.fixtures import scene_path
    res, spp = (args.res if args.res != 512 else 1024), (args.spp if args.spp != 64 else 512)
    if args.pmc_child:
        res, spp = args.res, args.spp
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("cbox_bunny"), False)
    sc.opts.width = sc.opts.height = res
    # geometry gradients with all three terms (interior + primary-edge + secondary-edge boundary integrals) come from the DirectIntegrator, the
    # reference's own configuration for them (the PathTracer has no secondary-edge term); renderC is the PathTracer
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, spp, spp, 0
    integ = psdr_cuda.PathTracer(max_depth=args.max_depth)
    integ_d = psdr_cuda.DirectIntegrator(1, 1)
    refl = sc.param_map["BSDF[0]"].reflectance
    base = ek.detach(refl.data)
    mesh = sc.param_map["Mesh[1]"]                      # the bunny (cbox_bunny.xml)
    v0 = ek.detach(mesh.vertex_positions)
    sc.configure()
    if args.pmc_child:                                  # one renderC of this rank's share under rocprofv3 (pmc_passes_c4)
        integ.renderC(sc); integ.renderC(sc)
        torch.cuda.synchronize()
        return

    def step():
        img = integ.renderC(sc)
        r = Vector3fD(base); ek.set_requires_gradient(r); refl.data = r
        v = Vector3fD(v0); ek.set_requires_gradient(v); mesh.vertex_positions = v
        sc.configure()
        imgD = integ_d.renderD(sc)
        ek.backward(FloatD._wrap(imgD.t.sum().reshape(1)))
        return img, ek.gradient(v), ek.gradient(r)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = 2.0 * res * res * spp * args.steps / dt / 1e6
    T = int(sc.tables(0)["num_tris"])
    # the wavefront stages' stream traffic against the HBM roofline: measured on one GPU's share of the 8-GPU job (64 spp: the launch size at
    # which the library runs the class-binned wavefront), record size 44 + 12 K bytes + 3 (1 + K) accumulator words per live path and stage
    wf = pmc_passes_c4(args, res, max(spp // 8, 1)) if (rank == 0 and not args.no_pmc) else None
    if dist and world > 1:
        dist.barrier()
    grad_words = T * 24 + int(sc.tables(0)["texels"].numel())
    if rank == 0:
        gv = out[1].numpy()
        print(json.dumps({
            "metric": "Mpath-samples/s renderC+renderD, cbox_bunny 1024x1024 spp=512 sharded (BASELINE configs[3])",
            "value": round(value, 3), "unit": "Mpath-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cbox_bunny %dx%d GLOBAL spp = sppe = sppse = %d (%d per GPU): PathTracer(max_depth=%d).renderC + configure + DirectIntegrator(1,1).renderD "
                                   "(interior + primary-edge + secondary-edge terms) + enoki.backward w.r.t. the bunny's vertex positions and the albedo texels, through the psdr_cuda surface"
                                   % (res, res, spp, spp // world, args.max_depth),
                       "triangles": T, "global_spp": spp, "world_size": world, "devices": devices, "rccl_version": rccl,
                       "collectives_forced": bool(dist is not None and world == 1),
                       "allreduce_bytes_per_step": 0 if (world == 1 and dist is None) else int(2 * res * res * 3 * 4 + grad_words * 4),
                       "parallelism": "spp-shard x%d; all-reduces per step: [image] (renderC), [image] (renderD primal), [triangle-row || texel gradients]" % world},
            "grad_check": {"finite": bool(np.isfinite(gv).all()), "abs_max_vertex_grad": float(np.abs(gv).max())},
            "wavefront_traffic": None if wf is None else {"workload": "renderC of one rank's share of the 8-GPU job (%d spp), two calls" % max(spp // 8, 1), "kernels": wf,
                                                           "stream_record_bytes": 44 + 12, "note": "(2 * FETCH_SIZE + WRITE_SIZE) KiB per kernel summed over its launches / its summed duration / 8 TB/s (rocprofv3 --pmc, one pass per counter)"},
        }))


def c4_strong(args, world, rank, dist, rccl, wait_all, one_integrator=False):
    """BASELINE configs[3] beside the headline at EVERY world size (north star: "Mpath-samples/s reported at 1/2/4/8 GPUs with achieved-HBM-fraction"):
    cbox_bunny 1024x1024, GLOBAL spp = sppe = sppse = 512 sharded over the ranks (strong scaling), step = PathTracer(3).renderC + configure +
    renderD + enoki.backward w.r.t. the bunny's vertex positions and the albedo texels, through the surface; the wavefront kernels' measured HBM traffic
    (rank 0's share under rocprofv3 --pmc) against 8 TB/s.  renderD's integrator: DirectIntegrator(1,1) with all three terms (the reference's own
    configuration for geometry gradients), or -- one_integrator, configs[3] read literally: ONE integrator -- the same PathTracer(3)
    (interior + primary-edge terms; the PathTracer has no secondary-edge term).  2 warm-up steps, 5 timed."""
    import enoki as ek
    import psdr_cuda
    from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
    from psdr_cuda.fixtures import scene_path
    res, spp, steps, warm = 1024, 512, 5, 2
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("cbox_bunny"), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, spp, spp, 0
    integ = psdr_cuda.PathTracer(max_depth=3)
    integ_d = integ if one_integrator else psdr_cuda.DirectIntegrator(1, 1)
    refl = sc.param_map["BSDF[0]"].reflectance
    base = ek.detach(refl.data)
    mesh = sc.param_map["Mesh[1]"]
    v0 = ek.detach(mesh.vertex_positions)
    sc.configure()

    def step():
        img = integ.renderC(sc)
        r = Vector3fD(base); ek.set_requires_gradient(r); refl.data = r
        v = Vector3fD(v0); ek.set_requires_gradient(v); mesh.vertex_positions = v
        sc.configure()
        imgD = integ_d.renderD(sc)
        ek.backward(FloatD._wrap(imgD.t.sum().reshape(1)))
        return img, ek.gradient(v), ek.gradient(r)

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    per_step = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        per_step.append((time.perf_counter() - t1) * 1e3)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tb = sc.tables(0)
    T = int(tb["num_tris"])
    grad_words = T * 24 + int(tb["texels"].numel())
    finite = bool(np.isfinite(out[1].numpy()).all())
    wf = None
    if rank == 0 and not args.no_pmc and not one_integrator:
        wf = pmc_passes_c4(args, res, max(spp // world, 1))
    wait_all()
    if rank != 0:
        return None
    d_words = ("PathTracer(3).renderD (interior + primary-edge terms; it has no secondary-edge term)" if one_integrator else
               "DirectIntegrator(1,1).renderD (interior + primary-edge + secondary-edge terms)")
    return {"workload": "cbox_bunny %dx%d GLOBAL spp = sppe = sppse = %d sharded over %d rank(s) (%d per GPU): PathTracer(3).renderC + configure + %s "
                        "+ enoki.backward w.r.t. the bunny's vertex positions and the albedo texels" % (res, res, spp, world, spp // world, d_words),
            "scaling": "strong", "steps": steps, "warmup": warm, "ms_per_step": round(dt / steps * 1e3, 3),
            "ms_per_step_min_max_this_rank": [round(min(per_step), 3), round(max(per_step), 3)], "value": round(2.0 * res * res * spp * steps / dt / 1e6, 2),
            "unit": "Mpath-samples/s (2 W H spp camera slots per step; the step also evaluates the W H sppe (+ sppse) boundary slots)", "world_size": world, "global_spp": spp,
            "triangles": T, "allreduce_bytes_per_step": 0 if dist is None else int(2 * res * res * 3 * 4 + grad_words * 4),
            "allreduces_per_step": "[image] (renderC), [image] (renderD primal), [triangle-row || texel gradients]", "rccl_version": rccl, "grad_finite": finite,
            "wavefront_hbm": None if wf is None else {"workload": "PathTracer(3).renderC of rank 0's share (%d spp), two calls under rocprofv3 --pmc" % max(spp // world, 1), "kernels": wf,
                                                      "note": "(2 * FETCH_SIZE + WRITE_SIZE) KiB per kernel summed over its launches / its summed duration / 8 TB/s"}}


def lib_arg(args):
    """the counter children measure the build the parent measures (developer runs with --hip-lib)"""
    return (["--hip-lib", args.hip_lib] if getattr(args, "hip_lib", None) else []) + [x for kv in getattr(args, "native_option", []) for x in ("--native-option", kv)]


def main():
    args = parse()
    if args.hip_lib:
        from psdr_cuda import _abi as _abi0
        _abi0.use_library(args.hip_lib)
    if args.tree_child:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU")
        torch.cuda.set_device(0)
        tree_child()
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        relaunch_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP render path has no CPU fallback)")
    # PSDR_BENCH_ONE_GPU=1 (developer switch, not a measurement): every rank on cuda:0 over gloo, so that the multi-rank code of this
    # script -- sharded spp, the all-reduces inside the render calls, the max-over-ranks clock -- can be EXECUTED on a one-GPU box
    one_gpu = os.environ.get("PSDR_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # PSDR_FORCE_COLLECTIVES=1 (developer switch): a one-rank job builds its nccl (= RCCL) process group as well and every collective of the
    # render calls EXECUTES (psdr_cuda/integrator.py _dist) -- the RCCL path on a one-GPU box (tests/test_rccl_single_rank_gpu.py)
    forced = world == 1 and os.environ.get("PSDR_FORCE_COLLECTIVES") == "1" and not args.pmc_child
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif forced:
        import socket
        import torch.distributed as dist
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

    if forced:
        from psdr_cuda.integrator import force_collectives
        force_collectives(True)
    args.local_rank = local_rank
    side = None
    if dist and world > 1:
        # host-side barrier (gloo): the other ranks wait here -- no spinning collective on their GPUs, no RCCL watchdog -- while rank 0 runs the counter
        # passes (child processes under rocprofv3) and the single-GPU side blocks
        from datetime import timedelta
        side = dist.new_group(backend="gloo", timeout=timedelta(hours=2))

    def wait_all():
        if side is not None:
            dist.barrier(group=side)

    devices, rccl = rank_devices(dist, local_rank)
    if args.config == "c4":
        run_c4(args, world, rank, local_rank, dist, devices, rccl)
        if dist:
            dist.destroy_process_group()
        return
    from psdr_cuda import _abi
    w = Workload(args, world)

    if args.pmc_child:                      # counted launches only: 2 x (renderC, renderD K=1, reverse)
        w.kernel_setup(1)
        for _ in range(2):
            w.kernel_c(); w.kernel_d(); w.kernel_rev()
        torch.cuda.synchronize()
        return

    # ---- the timed region: K surface steps (W warm-up steps before it; nothing else runs untimed in front)
    for _ in range(args.warmup):
        w.surface_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.surface_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    local_slots = args.res * args.res * args.spp
    slots_per_pass = local_slots * world                             # whole job
    value = 2.0 * slots_per_pass * args.steps / dt / 1e6

    # ---- beside it: the surface with reverse mode, the bare kernels (HIP events on the launch stream), host shares
    n_side = max(5, min(20, args.steps))
    ms_rev_surface = timed(w.surface_reverse_step, n_side)
    w.refl.data = w.Vector3fD(w.base); w.sc.configure()
    t1 = time.perf_counter()
    for _ in range(n_side):
        w.refl.data = w.Vector3fD(w.base); w.sc.configure()
    torch.cuda.synchronize()
    ms_configure = (time.perf_counter() - t1) / n_side * 1e3
    try:
        ms_rev_all_surface = timed(w.surface_reverse_all_step, n_side)
    except Exception as e:                                   # a scene without these parameters: reported as missing, never as a number
        import traceback
        print("surface reverse-all step skipped: %r\n%s" % (e, traceback.format_exc()), file=sys.stderr)
        ms_rev_all_surface = None
    w.kernel_setup(1)
    for _ in range(3):
        w.kernel_c(); w.kernel_d(); w.kernel_rev()
    ms_c = timed(w.kernel_c, n_side); rays_c = w.integ.last_counters[0]
    ms_d1 = timed(w.kernel_d, n_side); rays_d = w.integ.last_counters[0]
    ms_rev = timed(w.kernel_rev, n_side)
    w.kernel_rev_all()
    ms_rev_all = timed(w.kernel_rev_all, n_side)
    # the pair an optimisation step with a geometry gradient launches: recording primal render, then the adjoint kernel alone on its records
    w.kernel_c_keep(); w.kernel_rev_all()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    keep_ms = [[], []]
    for _ in range(n_side):
        e[0].record(); w.kernel_c_keep(); e[1].record(); w.kernel_rev_all(); e[2].record()
        torch.cuda.synchronize()
        keep_ms[0].append(e[0].elapsed_time(e[1])); keep_ms[1].append(e[1].elapsed_time(e[2]))
    ms_c_keep, ms_rev_all_kept = float(np.mean(keep_ms[0])), float(np.mean(keep_ms[1]))
    rays_rev_kept = w.integ.last_counters[0]
    w.kernel_setup(3)
    w.kernel_d()
    ms_d3 = timed(w.kernel_d, n_side)
    kernel_only = {"value": round(2.0 * slots_per_pass / ((ms_c + ms_d1) * 1e-3) / 1e6, 1), "unit": "Mpath-samples/s",
                   "render_c_ms": round(ms_c, 4), "render_d_fwd_k1_ms": round(ms_d1, 4), "render_d_fwd_k3_ms": round(ms_d3, 4),
                   "render_d_rev_ms": round(ms_rev, 4), "render_d_rev_all_ms": round(ms_rev_all, 4),
                   "render_c_keep_records_ms": round(ms_c_keep, 4), "render_d_rev_all_on_kept_records_ms": round(ms_rev_all_kept, 4), "rays_of_rev_on_kept_records": int(rays_rev_kept),
                   "note": "psdr_render_c + psdr_render_d_fwd (K=1) launches of the same samples; K=3 = d/d(r,g,b) in one pass (round-1 headline form); "
                           "rev = psdr_render_d_rev, texel gradient; rev_all = texels + emitter radiance + triangle rows (geometry) + camera pose; "
                           "keep_records pair = psdr_render_c(PSDR_FLAG_KEEP_RECORDS) as the value kernel + psdr_render_d_rev (all tables) as the adjoint kernel alone"}
    surface = {"ms_per_step": round(dt / args.steps * 1e3, 4), "configure_ms": round(ms_configure, 4),
               "reverse_step_ms": round(ms_rev_surface, 4),
               "reverse_all_step_ms": round(ms_rev_all_surface, 4) if ms_rev_all_surface is not None else None,
               "note": "step = renderC + [configure + renderD + enoki.forward] (one launch each: renderD is rendered by the forward-mode kernel); "
                       "reverse_step = configure + renderD + enoki.backward (primal launch + psdr_render_d_rev); reverse_all_step = the same with gradients of the albedo, the light's radiance, a wall's vertices and the camera pose"}

    # ---- roofline of the dominant kernel: VALU issue (not HBM: the path state lives in registers)
    # (rank 0 at ANY world size: the child is one process on rank 0's GPU with this rank's per-GPU workload; the other ranks wait at wait_all() below)
    pmc = {} if (args.no_pmc or rank != 0) else pmc_passes(args)
    dom_key, dom_ms, dom_name = ("d", ms_d1, "k_camera<float, Dual<1>, PATH> (renderD fwd)") if ms_d1 >= ms_c else ("c", ms_c, "k_camera<float, float, PATH> (renderC)")
    if dom_key == "d" and "k_camera_logd" in str(pmc.get("d", {}).get("_kernel", "")):
        dom_name = "%s (renderD fwd: the log-derivative kernel; kernel_ms also holds its gate kernels and the dual-number kernel that returns at once)" % pmc["d"]["_kernel"]
    dom_rays = rays_d if dom_key == "d" else rays_c
    valu = pmc.get(dom_key, {}).get("SQ_INSTS_VALU")
    fetch, write = pmc.get(dom_key, {}).get("FETCH_SIZE"), pmc.get(dom_key, {}).get("WRITE_SIZE")
    traffic = None if fetch is None or write is None else (2.0 * fetch + write) * 1024.0
    achieved = None if valu is None else valu / (dom_ms * 1e-3)
    # algorithmic floor: VALU lane-instructions a traced ray NEEDS on this scene -- one test per PRIMITIVE the kernel holds (the 12 wall
    # triangles of the Cornell box are 6 parallelograms, psdr_bvh_build.h pack_tiny_prims) x 27 (Moeller-Trumbore with SGPR operands) +
    # ~150 for hit reconstruction, sampling and shading of its path vertex -- over the lane-instructions issued (64 per wave-instruction)
    st = _abi.scene_stats(w.sc._native)
    n_prims = int(st.get("n_tiny", 0)) or int(w.tb["num_tris"])
    n_slab = int(st.get("n_slab", 0))
    floor_lane_insts = (n_slab * FLOOR_VALU_PER_SLAB_TEST + (n_prims - n_slab) * FLOOR_VALU_PER_PRIM_TEST + FLOOR_VALU_PER_RAY_REST) * float(dom_rays)
    dpm = pmc.get(dom_key, {})
    wave_cycles = dpm.get("SQ_WAVE_CYCLES")
    roofline = {
        "bound": "valu", "kernel": dom_name, "kernel_ms": round(dom_ms, 4),
        "achieved": None if achieved is None else round(achieved / 1e9, 3), "peak": round(VALU_PEAK_WAVE_INSTS_PER_S / 1e9, 3),
        "unit": "G wave-instructions/s", "frac": None if achieved is None else round(achieved / VALU_PEAK_WAVE_INSTS_PER_S, 4),
        "peak_note": "1024 SIMD-32 units x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)",
        "valu_wave_insts_per_launch": valu, "algorithmic_floor_frac": None if valu is None else round(floor_lane_insts / (valu * 64.0), 4),
        "floor_time_frac": None if valu is None else round(floor_lane_insts / 64.0 / VALU_PEAK_WAVE_INSTS_PER_S / (dom_ms * 1e-3), 4),
        "floor_time_note": "floor_time_frac = (time the algorithmic floor needs at the VALU peak) / kernel time = algorithmic_floor_frac x frac: the useful-work fraction; `frac` alone is an ISSUE rate "
                           "(a kernel that issues fewer instructions for the same result lowers it)",
        "primitives_tested_per_ray": n_prims, "slab_form_primitives": n_slab,
        "wait_any_frac": None if not wave_cycles or "SQ_WAIT_ANY" not in dpm else round(dpm["SQ_WAIT_ANY"] / wave_cycles, 4),
        "wait_inst_any_frac": None if not wave_cycles or "SQ_WAIT_INST_ANY" not in dpm else round(dpm["SQ_WAIT_INST_ANY"] / wave_cycles, 4),
        "valu_active_frac_of_wave_cycles": None if not wave_cycles or "SQ_ACTIVE_INST_VALU" not in dpm else round(dpm["SQ_ACTIVE_INST_VALU"] / wave_cycles, 4),
        "traffic": traffic, "hbm_measured_frac": None if traffic is None else round(traffic / (dom_ms * 1e-3) / HBM_BPS, 5),
        "traffic_note": "(2 * FETCH_SIZE + WRITE_SIZE) KiB per launch from rocprofv3 --pmc passes run by this script; image + derivative image = %.1f MB"
                        % (local_slots / args.spp * 12 * 2 / 1e6),
        "rays_per_slot": round(dom_rays / local_slots, 4),
        "stream_model_equiv": {"bytes_per_launch": stream_model_bytes(local_slots, dom_rays, dom_key == "d"),
                               "GBps": round(stream_model_bytes(local_slots, dom_rays, dom_key == "d") / (dom_ms * 1e-3) / 1e9, 1),
                               "note": "SURVEY 8(d) wavefront-stream bytes; NOT moved by the fused kernel, not a roofline"},
        "other_kernels": {k: {"valu_wave_insts": v.get("SQ_INSTS_VALU"), "hbm_bytes": None if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v else (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0}
                          for k, v in pmc.items() if k != dom_key},
    }

    from psdr_cuda.integrator import solo
    cpu, grad = None, None
    if rank == 0 and not args.no_cpu_baseline:
      with solo():                                       # rank 0 alone: no sharding, no collective in the render calls of this block
          sys.path.insert(0, os.path.join(ROOT, "oracle"))
          import oracle
          import psdr_cuda
          from psdr_cuda.fixtures import scene_path
          # ---- CPU baseline: the oracle (a port of the reference's estimator) on the host cores, bounded sample
          w.kernel_setup(1)
          tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in w.tb.items()}
          cspp = 2
          o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=args.max_depth, spp=cspp)
          cores = os.cpu_count() or 1
          tt = torch.zeros_like(tbc["texels"]); tt[0:3] = 1.0
          oracle.render(tbc, _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=args.max_depth, spp=1), nthreads=cores)
          c0 = time.perf_counter()
          reps = 0
          while time.perf_counter() - c0 < 10.0:
              oracle.render(tbc, o, nthreads=cores)
              oracle.render(tbc, o, mode=1, tangents={"texels": tt}, nthreads=cores)
              reps += 1
          cdt = time.perf_counter() - c0
          cpu = {"value": round(2.0 * args.res * args.res * cspp * reps / cdt / 1e6, 4), "unit": "Mpath-samples/s", "cores": cores, "kind": "port",
                 "sample": "%d x (renderC + renderD fwd K=1) of the same scene at %dx%d spp=%d, oracle fp32, %d threads" % (reps, args.res, args.res, cspp, cores)}
          # ---- the metric's parity half: d image / d albedo, HIP vs the oracle on the same sample streams, at a size the
          # oracle finishes in a second.  Three references: fp32 in the product's forms, fp32 in the reference's literal
          # forms, fp64 literal (= the exact value of the reference's estimator).  tests/ hold the full parity suite.
          gres, gspp = 64, 8
          sc2 = psdr_cuda.Scene()
          sc2.load_file(scene_path(args.scene), False)
          sc2.opts.width = sc2.opts.height = gres
          sc2.opts.spp, sc2.opts.sppe, sc2.opts.sppse, sc2.opts.log_level = gspp, 0, 0, 0
          sc2.configure()
          tb2 = sc2.tables(0)
          o2 = w.integ._opts(sc2, with_edges=False)
          ts2 = []
          for c in range(3):
              t = torch.zeros_like(tb2["texels"]); t[c] = 1.0
              ts2.append([None, t, None, None, None, None, None])
          _, dimgs = w.integ._render_fwd(sc2, tb2, o2, None, ts2)

          def worst(**kw):
              r = 0.0
              for c in range(3):
                  ref = oracle.render(tb2, o2, mode=1, tangents={"texels": ts2[c][1]}, **kw)[1].reshape(-1).astype(np.float64)
                  got = dimgs[c].cpu().numpy().astype(np.float64)
                  r = max(r, float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)))
              return round(r, 8)
          grad = {"rel_l2": worst(precision=1, reference_form=True), "bound": 1e-3,
                  "rel_l2_vs_fp32_same_forms": worst(precision=0), "rel_l2_vs_fp32_reference_forms": worst(precision=0, reference_form=True),
                  "check": "d image / d albedo(r,g,b), %s %dx%d spp=%d PathTracer(max_depth=%d), HIP vs CPU oracle on the same sample streams; rel_l2 = against "
                           "fp64 in the reference's literal forms" % (args.scene, gres, gres, gspp, args.max_depth)}

    trees = None
    if rank == 0 and not args.no_tree_scenes:
        try:
            with solo():
                trees = tree_scenes(args)
        except Exception as e:                                  # reported as missing, never as a number
            trees = {"error": repr(e)}
    wait_all()
    # ---- BASELINE configs[3] as a strong-scaling block, every rank takes part
    strong, strong_one = None, None
    if not args.no_c4_strong:
        try:
            strong = c4_strong(args, world, rank, dist, rccl, wait_all)
        except Exception as e:
            strong = {"error": repr(e)}
            wait_all()
        try:
            strong_one = c4_strong(args, world, rank, dist, rccl, wait_all, one_integrator=True)
        except Exception as e:
            strong_one = {"error": repr(e)}
            wait_all()
    if rank == 0:
        out = {
            "metric": "Mpath-samples/s renderC+renderD, cbox 512x512 spp=64; grad rel-L2 vs ref",
            "value": round(value, 3), "unit": "Mpath-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d spp=%d/GPU PathTracer(max_depth=%d): renderC + configure + renderD + enoki.forward w.r.t. diffuse albedo, through the psdr_cuda surface"
                                   % (args.scene, args.res, args.res, args.spp, args.max_depth),
                       "triangles": int(w.tb["num_tris"]), "global_spp": args.spp * world, "world_size": world,
                       "device": torch.cuda.get_device_name(local_rank), "local_rank": local_rank, "devices": devices, "rccl_version": rccl,
                       "allreduce_bytes_per_step": 0 if world == 1 else int(args.res * args.res * 3 * 4 * 3),
                       "parallelism": "spp-shard x%d, one all-reduce per render call ([image] for renderC, [image || derivative image] for renderD)" % world},
            "surface": surface, "kernel_only": kernel_only,
            "roofline": roofline, "cpu_baseline": cpu, "grad_rel_l2": grad, "tree_scenes": trees, "c4_strong": strong, "c4_strong_one_integrator": strong_one,
        }
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
