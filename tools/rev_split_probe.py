#!/usr/bin/env python
"""Reverse mode, one kernel against the split launch (value kernel + adjoint kernel), per scene and integrator (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
from psdr_cuda.fixtures import make_interior_scene


def timeit(fn, reps=3):
    """median of individually timed calls after one warm-up call"""
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(max(reps, 3)):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


res, spp = 512, 16
adj = np.random.default_rng(0).random((res * res, 3)).astype(np.float32)
for name in sys.argv[1:] or ["cbox_bunny", "bunny_light", "interior", "cbox"]:
    if name == "interior":
        sc = make_interior_scene(seed=0, n_objects=10, res=res, spp=spp); sc.configure(); tb = sc.tables(0)
    else:
        tb = load_scene(name, res=res, spp=spp)[0].tables(0)
    g = GpuScene(tb)
    for kind, kw in (("direct11", dict(bsdf_samples=1, light_samples=1)), ("path3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)),
                     ("path6", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=6))):
        o = _abi.make_opts(spp=spp, **kw)
        t = {}
        for mode in ("0", "1"):
            os.environ["PSDR_OPTIONS"] = "rev_split=%s" % mode
            t[mode] = timeit(lambda: g.render_d_rev(o, adj, want=["tri_info", "texels"], with_image=False))
        print("%-12s %-8s one kernel %6.2f ms   split %6.2f ms" % (name, kind, t["0"], t["1"]))
