#!/usr/bin/env python
"""Cost of the guiding-grid build (DirectIntegrator.preprocess_secondary_edges) next to the renderD it serves (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for name in ("cbox_bunny", "bunny_light"):
    sc, _ = load_scene(name, res=512, spp=16, sppe=16, sppse=16)
    tb = sc.tables(0); g = GpuScene(tb)
    o = _abi.make_opts(spp=16, sppe=16, sppse=16, bsdf_samples=1, light_samples=1)
    for reso, rounds in (((200, 4, 4, 2), 2), ((1000, 8, 8, 2), 4), ((5000, 10, 10, 4), 1)):
        cells = reso[0] * reso[1] * reso[2]
        print("%-12s guide grid %s x %d samples x %d rounds (%d cells): %.2f ms" % (name, reso[:3], reso[3], rounds, cells, timeit(lambda: g.guide_build(o, reso, rounds))))
