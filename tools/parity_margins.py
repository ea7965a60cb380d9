#!/usr/bin/env python
"""How far the HIP build sits from the oracle on the parity-test workloads (developer tool): the rel-L2 the tests
bound by 1e-4, for several RNG offsets -- a build change that moves these towards the bound is a warning."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import oracle
from helpers import GpuScene, load_scene, rel_l2
from psdr_cuda import _abi
OPTS = {"direct11": dict(bsdf_samples=1, light_samples=1), "path3": dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)}
for scene in ("cbox", "cbox_rough", "cbox_occluder", "cbox_env"):
    sc, _ = load_scene(scene, res=48, spp=16)
    tb = sc.tables(0); g = GpuScene(tb)
    for kind, kw in OPTS.items():
        vals = []
        for off in (7, 1007, 2007, 3007, 4007, 5007):
            o = _abi.make_opts(spp=16, rng_offset=(off, 0, 0), **kw)
            vals.append(rel_l2(g.render_c(o), oracle.render(tb, o)))
        print("%-14s %-9s rel-L2 vs oracle: max %.2e  median %.2e   (test bound 1e-4)" % (scene, kind, max(vals), float(np.median(vals))), flush=True)
