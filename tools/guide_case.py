#!/usr/bin/env python
"""Guiding-grid build at the reference's size (examples/config.py cbox_MIS: (40000, 5, 5, 2), 32 rounds) on cbox_bunny through the C ABI: ms per build.
usage: guide_case.py [rounds] [reps]   (developer tool; run under rocprofv3 --kernel-trace --stats for the per-kernel times)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc, _ = load_scene("cbox_bunny", res=512, spp=4, sppe=4, sppse=4)
tb = sc.tables(0)
g = GpuScene(tb)
o = _abi.make_opts(spp=4, sppe=4, sppse=4)
reso = (40000, 5, 5, 2)
g.guide_build(o, reso, 1)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); m = g.guide_build(o, reso, rounds); ts.append((time.perf_counter() - t0) * 1e3)
print("guiding (40000, 5, 5, 2) x %d rounds on cbox_bunny: %s ms (median %.2f), rays %d, non-zero cells %d, mass sum %.6e" % (
    rounds, " ".join("%.2f" % t for t in ts), sorted(ts)[len(ts) // 2], g.counters()[0], (m > 0).sum(), m.sum()))
