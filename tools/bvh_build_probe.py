#!/usr/bin/env python
"""Host SAH tree against the device-built radix tree (developer tool): build time and PathTracer(3) renderC time per scene.
the options travel through PSDR_OPTIONS (tests/helpers.py GpuScene -> psdr_scene_set_option)."""
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


def tables(name):
    if name == "interior":
        sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure(); return sc.tables(0)
    return load_scene(name, res=512, spp=16)[0].tables(0)


for name in sys.argv[1:] or ["cbox_bunny", "bunny_light", "interior"]:
    tb = tables(name)
    o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    for mode, two in (("host", "1"), ("host", "0"), ("device", "0")):
        os.environ["PSDR_OPTIONS"] = "bvh_build=%d,two_level=%s" % (1 if mode == "device" else 0, two)
        torch.cuda.synchronize(); t0 = time.perf_counter(); g = GpuScene(tb); torch.cuda.synchronize(); tb_ms = (time.perf_counter() - t0) * 1e3
        print("%-12s T=%6d  %-6s two-level=%s  handle+build %7.2f ms   path3 renderC %6.2f ms" % (name, tb["tri_info"].shape[0], mode, two, tb_ms, timeit(lambda: g.render_c(o))))
