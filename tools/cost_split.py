#!/usr/bin/env python
"""where a camera sample's time goes (developer tool): primary ray only / + BSDF sample / + light sample / per bounce"""
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
for scene in sys.argv[1:] or ["cbox", "cbox_bunny"]:
    sc, _ = load_scene(scene, res=512, spp=64 if scene == "cbox" else 16); tb = sc.tables(0); g = GpuScene(tb)
    spp = sc.opts.spp
    cases = [("field depth (primary ray)", dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["depth"])),
             ("direct B=1 L=0", dict(bsdf_samples=1, light_samples=0)), ("direct B=0 L=1", dict(bsdf_samples=0, light_samples=1)),
             ("direct B=1 L=1", dict(bsdf_samples=1, light_samples=1)), ("direct B=2 L=2", dict(bsdf_samples=2, light_samples=2))]
    cases += [("path depth %d" % d, dict(integrator=_abi.INTEGRATOR_PATH, max_depth=d, flags=_abi.FLAG_FUSED)) for d in (1, 2, 3, 4)]
    for name, kw in cases:
        o = _abi.make_opts(spp=spp, **kw)
        ms = timeit(lambda: g.render_c(o)); r = g.counters()[0]
        print("%-10s %-28s %7.3f ms  rays %6.1f M  %6.2f ns/slot" % (scene, name, ms, r / 1e6, ms * 1e6 / (512 * 512 * spp)))
