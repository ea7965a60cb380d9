#!/usr/bin/env python
"""quick traversal-bound timings (developer tool): C3 bunny path3, C5 path3, C2 bench kernels"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
out = []
for scene, res, spp in (("cbox", 512, 64), ("bunny_light", 512, 16), ("cbox_bunny", 512, 16)):
    sc, _ = load_scene(scene, res=res, spp=spp); tb = sc.tables(0); g = GpuScene(tb)
    for name, kw in (("direct11", dict(bsdf_samples=1, light_samples=1)), ("path3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=_abi.FLAG_FUSED))):
        o = _abi.make_opts(spp=spp, **kw)
        out.append("%s %s %.2f" % (scene, name, timeit(lambda: g.render_c(o))))
print(" | ".join(out))
