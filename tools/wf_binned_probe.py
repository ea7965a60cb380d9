#!/usr/bin/env python
"""Developer tool (GPU box): PathTracer renderC on the tree scenes -- fused kernel against the wavefront, whose streams are
binned by cost class on two-level scenes (PSDR_WF_BINNED=0: plain streams, PSDR_TWO_LEVEL=0: one tree)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene, rel_l2
from psdr_cuda import _abi


def timeit(fn, reps=3):
    """median of `reps` individually timed calls after one warm-up call (a host-side stall -- the caching allocator returning blocks, a
    page-in -- lands in one repetition, not in the figure)"""
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(max(reps, 3)):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def run(name, tb, spp, spp_range, depths=(3,)):
    g = GpuScene(tb)
    n = tb["width"] * tb["height"] * (spp_range[1] - spp_range[0])
    for depth in depths:
        imgs = {}
        for fl, fn in ((_abi.FLAG_FUSED, "fused"), (_abi.FLAG_WAVEFRONT, "wavefront")):
            o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=spp, spp_range=spp_range, flags=fl)
            ms = timeit(lambda: g.render_c(o), reps=2)
            imgs[fn] = g.render_c(o)
            print("%-22s path%d %-9s %8.2f ms  %7.0f Msamples/s  rays/slot %.2f" % (name, depth, fn, ms, n / ms / 1e3, g.counters()[0] / n), flush=True)
        print("%-22s path%d wavefront vs fused rel-L2 %.1e" % (name, depth, rel_l2(imgs["wavefront"], imgs["fused"])))


which = sys.argv[1:] or ["c3", "c4", "c5"]
if "c3" in which:
    sc, _ = load_scene("cbox_bunny", res=512, spp=16)
    run("cbox_bunny 512 spp16", sc.tables(0), 16, (0, 16), (3, 6))
if "c4" in which:
    sc, _ = load_scene("cbox_bunny", res=1024, spp=512)
    run("C4 shard (67M slots)", sc.tables(0), 512, (0, 64))
if "c5" in which:
    from psdr_cuda.fixtures import make_interior_scene
    sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure()
    run("C5 interior 512 spp16", sc.tables(0), 16, (0, 16), (3, 6))
if "open" in which:
    sc, _ = load_scene("bunny_light", res=512, spp=32)
    run("bunny_light 512 spp32", sc.tables(0), 32, (0, 32), (3, 6))
