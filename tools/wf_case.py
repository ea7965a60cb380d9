#!/usr/bin/env python
"""One tree workload of BASELINE's configs through the C ABI, a few calls (developer tool: run under rocprofv3 --kernel-trace --stats or --pmc).
usage: wf_case.py c4|c5|c3b [fused|wavefront|default] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
case = sys.argv[1] if len(sys.argv) > 1 else "c4"
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flags = {"fused": _abi.FLAG_FUSED, "wavefront": _abi.FLAG_WAVEFRONT, "default": 0}[mode]
if case == "c4":
    sc, _ = load_scene("cbox_bunny", res=1024, spp=512, sppe=0, sppse=0)
    o = _abi.make_opts(spp=512, spp_range=(0, 64), integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 1024 * 1024 * 64
elif case == "c5":
    from psdr_cuda.fixtures import make_interior_scene
    sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure()
    o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 512 * 512 * 16
elif case in ("c4pr", "c5pr"):
    # PathTracer(3) reverse (triangle rows + texels): C4 shard / the 50 k-triangle interior
    if case == "c4pr":
        sc, _ = load_scene("cbox_bunny", res=1024, spp=512, sppe=0, sppse=0)
        o = _abi.make_opts(spp=512, spp_range=(0, 64), integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 1024 * 1024 * 64
    else:
        from psdr_cuda.fixtures import make_interior_scene
        sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure()
        o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 512 * 512 * 16
elif case in ("c4fg", "c5fg"):
    # PathTracer(3) renderD forward, K = 1 geometry tangents (a translation of one object): C4 shard / the 50 k-triangle interior
    from helpers import tangents_wrt
    if case == "c4fg":
        sc, P = load_scene("cbox_bunny", res=1024, spp=512, sppe=0, sppse=0, translate=(1, (1.0, 0.0, 0.0)))
        o = _abi.make_opts(spp=512, spp_range=(0, 64), integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 1024 * 1024 * 64
    else:
        import enoki as ek
        from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
        from psdr_cuda.fixtures import make_interior_scene
        sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16)
        P = FloatD(0.); ek.set_requires_gradient(P)
        sc.m_meshes[8].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.4, -0.3]) * P)); sc.configure()
        o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 512 * 512 * 16
elif case in ("c2ra", "c2rt", "c2keep"):
    # C2 (cbox 512^2 spp 64, no tree) PathTracer(3) reverse: every gradient table / the texels only
    sc, _ = load_scene("cbox", res=512, spp=64)
    o = _abi.make_opts(spp=64, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 512 * 512 * 64
elif case in ("blp", "blpr"):
    # bunny_light 512^2 spp 128, PathTracer(3): renderC / reverse (rows + texels)
    sc, _ = load_scene("bunny_light", res=512, spp=128)
    o = _abi.make_opts(spp=128, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 512 * 512 * 128
elif case in ("blr", "blf", "blc"):
    # BASELINE configs[2] as worded: bunny_light 512^2 spp = sppe = sppse = 128, DirectIntegrator(1,1): reverse (all tables) / forward K = 1 (a translation of Mesh[0]) / renderC
    from helpers import tangents_wrt
    sc, P = load_scene("bunny_light", res=512, spp=128, sppe=128, sppse=128, translate=(0, (1.0, 0.0, 0.0)))
    o = _abi.make_opts(spp=128, sppe=128, sppse=128) if case != "blc" else _abi.make_opts(spp=128); n = 512 * 512 * 128 * (1 if case == "blc" else 3)
elif case in ("c3f", "c3r", "c4r3"):
    # the three-term DirectIntegrator workloads: C3 forward (K = 1, a translation of the bunny) / reverse at 512^2 spp 16; C4 shard reverse
    from helpers import tangents_wrt
    big = case == "c4r3"
    res, spp = (1024, 512) if big else (512, 16)
    sc, P = load_scene("cbox_bunny", res=res, spp=spp, sppe=spp, sppse=spp, translate=(1, (1.0, 0.0, 0.0)))
    rng = dict(spp_range=(0, 64), sppe_range=(0, 64), sppse_range=(0, 64)) if big else {}
    o = _abi.make_opts(spp=spp, sppe=spp, sppse=spp, **rng); n = res * res * (64 if big else spp) * 3
else:
    sc, _ = load_scene("cbox_bunny", res=256, spp=64)
    o = _abi.make_opts(spp=64, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 256 * 256 * 64
tb = sc.tables(0)
g = GpuScene(tb)
if case in ("c3f", "c4fg", "c5fg", "blf"):
    tan = tangents_wrt(tb, P)
    run = lambda: g.render_d_fwd(o, [tan])
elif case in ("c3r", "c4r3", "blr"):
    adj = np.random.default_rng(0).random((tb["width"] * tb["height"], 3)).astype(np.float32)
    run = lambda: g.render_d_rev(o, adj, with_image=False)
elif case in ("c4pr", "c5pr", "blpr"):
    adj = np.random.default_rng(0).random((tb["width"] * tb["height"], 3)).astype(np.float32)
    run = lambda: g.render_d_rev(o, adj, want=os.environ.get("WF_WANT", "tri_info,texels").split(","), with_image=False)          # WF_WANT: which gradient tables (A/B of what a table's adds cost)
elif case in ("c2ra", "c2rt"):
    adj = np.random.default_rng(0).random((tb["width"] * tb["height"], 3)).astype(np.float32)
    run = lambda: g.render_d_rev(o, adj, want=["texels", "emitter_rad", "tri_info", "cam_to_world"] if case == "c2ra" else ["texels"], with_image=False)
elif case == "c2keep":
    # the pair of an optimisation step with geometry gradients on a scene without a tree: recording primal render + the adjoint kernel on its records
    adj = np.random.default_rng(0).random((tb["width"] * tb["height"], 3)).astype(np.float32)
    ok = _abi.make_opts(spp=64, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags | _abi.FLAG_KEEP_RECORDS)
    def run():
        t0 = time.perf_counter(); g.render_c(ok); t1 = time.perf_counter()
        g.render_d_rev(o, adj, want=["texels", "emitter_rad", "tri_info", "cam_to_world"], with_image=False)
        run.parts = ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3)
else:
    run = lambda: g.render_c(o)
run(); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("%s %s: %s ms (median %.2f), rays/slot %.2f%s" % (case, mode, " ".join("%.2f" % t for t in ts), sorted(ts)[len(ts) // 2], g.counters()[0] / n,
                                                         "  last call: render_c %.2f + render_d_rev %.2f ms" % run.parts if case == "c2keep" else ""))
