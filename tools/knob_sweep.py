#!/usr/bin/env python
"""Environment-knob sweep of the forward PathTracer(3) renderC on the tree scenes (developer tool):
   python tools/knob_sweep.py PSDR_LDS_BUDGET 20480 28672 36864 49152"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "child":
    for p in ("psdr-cuda_amd", "oracle", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import numpy as np, torch
    from helpers import GpuScene, load_scene
    from psdr_cuda import _abi
    from psdr_cuda.fixtures import make_interior_scene

    def timeit(fn, reps=3):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]
    out = []
    for name in ("cbox_bunny", "interior", "bunny_light"):
        if name == "interior":
            sc = make_interior_scene(seed=0, n_objects=10, res=512, spp=16); sc.configure(); tb = sc.tables(0)
        else:
            tb = load_scene(name, res=512, spp=16)[0].tables(0)
        g = GpuScene(tb)
        for kind, kw in (("d11", dict(bsdf_samples=1, light_samples=1)), ("p3", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=_abi.FLAG_FUSED))):
            o = _abi.make_opts(spp=16, **kw)
            out.append("%s/%s %.2f" % (name, kind, timeit(lambda: g.render_c(o))))
    print("  ".join(out))
else:
    knob = sys.argv[1]
    for v in sys.argv[2:]:
        env = dict(os.environ); env[knob] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print("%s=%-8s %s" % (knob, v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
