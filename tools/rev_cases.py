#!/usr/bin/env python
"""Reverse-mode timings only (developer tool): C2 direct / path3, C3 bunny 3 terms, C5 interior path3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi


def timeit(fn, reps=3):
    """median of `reps` individually timed calls after one warm-up call (a host-side stall -- the caching allocator returning blocks, a
    page-in -- lands in one repetition, not in the figure)"""
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(max(reps, 3)):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


sc, _ = load_scene("cbox", res=512, spp=64)
tb = sc.tables(0); g = GpuScene(tb)
adj = np.random.default_rng(0).random((512 * 512, 3)).astype(np.float32)
for name, kw in (("direct11", dict(bsdf_samples=1, light_samples=1)), ("path3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3))):
    o = _abi.make_opts(spp=64, **kw)
    print("C2 %-8s rev texels %7.2f ms   rev all %7.2f ms" % (name, timeit(lambda: g.render_d_rev(o, adj, want=["texels"], with_image=False)),
          timeit(lambda: g.render_d_rev(o, adj, want=["texels", "emitter_rad", "tri_info", "cam_to_world"], with_image=False))))
sc, _ = load_scene("cbox_rough", res=512, spp=64)
tb = sc.tables(0); g = GpuScene(tb)
o = _abi.make_opts(spp=64, integrator=_abi.INTEGRATOR_PATH, max_depth=3)
print("C2 rough path3 rev all %7.2f ms" % timeit(lambda: g.render_d_rev(o, adj, want=["texels", "tri_info"], with_image=False)))
sc, _ = load_scene("bunny_light", res=512, spp=16, sppe=16, sppse=16)
tb = sc.tables(0); g = GpuScene(tb)
o = _abi.make_opts(spp=16, sppe=16, sppse=16, bsdf_samples=1, light_samples=1)
print("C3 bunny direct11 3 terms rev %7.2f ms" % timeit(lambda: g.render_d_rev(o, adj, want=["tri_info", "sec_edge", "prim_edge"], with_image=False)))
from psdr_cuda.fixtures import make_interior_scene
sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure()
tb = sc.tables(0); g = GpuScene(tb)
for name, kw in (("direct11", dict(bsdf_samples=1, light_samples=1)), ("path3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3))):
    o = _abi.make_opts(spp=16, **kw)
    print("C5 %-8s rev tri+texels %7.2f ms" % (name, timeit(lambda: g.render_d_rev(o, adj, want=["tri_info", "texels"], with_image=False))))
