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
else:
    sc, _ = load_scene("cbox_bunny", res=256, spp=64)
    o = _abi.make_opts(spp=64, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=flags); n = 256 * 256 * 64
g = GpuScene(sc.tables(0))
g.render_c(o); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); g.render_c(o); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("%s %s path3 renderC: %s ms (median %.2f), rays/slot %.2f" % (case, mode, " ".join("%.2f" % t for t in ts), sorted(ts)[len(ts) // 2], g.counters()[0] / n))
