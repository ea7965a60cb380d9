#!/usr/bin/env python
"""Leaf size / traversal-cost knobs of the host SAH builder against PathTracer(3) renderC time (developer tool)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
        o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=_abi.FLAG_FUSED)
        out.append("%s %.2f" % (name, timeit(lambda: g.render_c(o))))
    print("leaf=%s tcost=%s  " % (os.environ.get("PSDR_OPTIONS", ""), "") + "  ".join(out))
else:
    combos = [tuple(a.split(":")) for a in sys.argv[1:]] or [(l, t) for l in ("2", "4", "6", "8") for t in ("0.5", "1.0", "2.0")]
    for leaf, tc in combos:
        if True:
            env = dict(os.environ, PSDR_OPTIONS="bvh_maxleaf=%s,bvh_tcost=%s" % (leaf, tc))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
