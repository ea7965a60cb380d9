#!/usr/bin/env python
"""Reverse camera kernel time by requested gradient table (developer tool): which sink traffic costs what."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi


def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


for scene in sys.argv[1:] or ["cbox", "cbox_rough"]:
    sc, _ = load_scene(scene, res=512, spp=64)
    tb = sc.tables(0); g = GpuScene(tb)
    adj = np.random.default_rng(0).random((512 * 512, 3)).astype(np.float32)
    for name, kw in (("direct11", dict(bsdf_samples=1, light_samples=1)), ("path3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3))):
        o = _abi.make_opts(spp=64, **kw)
        row = ["renderC %.2f" % t(lambda: g.render_c(o))]
        for want in (["emitter_rad"], ["texels"], ["cam_to_world"], ["tri_info"], ["tri_info", "texels", "emitter_rad", "cam_to_world"]):
            row.append("%s %.2f" % ("+".join(w[:3] for w in want), t(lambda: g.render_d_rev(o, adj, want=want, with_image=False))))
        print("%-10s %-8s" % (scene, name), "  ".join(row), flush=True)
