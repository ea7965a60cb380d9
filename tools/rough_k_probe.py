#!/usr/bin/env python
"""Forward-mode renderD on a rough-conductor scene: K = 3 tangent sets in one pass against three K = 1 passes (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
from psdr_cuda.fixtures import make_interior_scene


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


for name in ("cbox_rough", "interior", "cbox"):
    if name == "interior":
        sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure(); tb = sc.tables(0); spp = 16
    else:
        tb = load_scene(name, res=512, spp=64)[0].tables(0); spp = 64
    g = GpuScene(tb)
    sets = [{"texels": torch.eye(tb["texels"].numel())[c]} for c in range(3)]
    for kind, kw in (("direct11", dict(bsdf_samples=1, light_samples=1)), ("path3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3))):
        o = _abi.make_opts(spp=spp, **kw)
        t3 = timeit(lambda: g.render_d_fwd(o, sets))
        t1 = timeit(lambda: g.render_d_fwd(o, sets[:1]))
        print("%-10s %-8s K=3 %6.2f ms   K=1 %6.2f ms (x3 = %.2f)   renderC %6.2f ms" % (name, kind, t3, t1, 3 * t1, timeit(lambda: g.render_c(o))))
