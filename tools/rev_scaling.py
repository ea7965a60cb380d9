#!/usr/bin/env python
"""reverse-mode time vs problem size (developer tool): exposes the fixed cost of the gradient-sink flush"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for scene in ("cbox", "cbox_bunny"):
    for res, spp in ((64, 16), (128, 16), (256, 16), (512, 16), (512, 64)):
        sc, _ = load_scene(scene, res=res, spp=spp); tb = sc.tables(0); g = GpuScene(tb)
        adj = np.random.default_rng(0).random((res * res, 3)).astype(np.float32)
        o = _abi.make_opts(spp=spp, bsdf_samples=1, light_samples=1)
        a = timeit(lambda: g.render_d_rev(o, adj, want=["texels", "tri_info", "cam_to_world"], with_image=False))
        b = timeit(lambda: g.render_d_rev(o, adj, want=["texels"], with_image=False))
        c = timeit(lambda: g.render_c(o))
        print("%-10s %4d^2 x %2d = %8d slots: rev all %6.2f ms  rev texels %6.2f ms  renderC %6.2f ms" % (scene, res, spp, res * res * spp, a, b, c))
