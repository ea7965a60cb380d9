#!/usr/bin/env python
"""psdr_trace throughput (developer tool): coherent camera rays and incoherent bounce rays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, camera_rays, load_scene
for scene in sys.argv[1:] or ["cbox", "cbox_bunny"]:
    sc, _ = load_scene(scene, res=64)
    tb = sc.tables(0); g = GpuScene(tb)
    n = 4_000_000
    o, d = camera_rays(tb, n, seed=1)
    shape, tri, u, v = g.trace(o, d)
    info = tb["tri_info"].cpu().numpy()
    hit = tri >= 0
    p = info[tri[hit], 0:3] + u[hit, None] * info[tri[hit], 3:6] + v[hit, None] * info[tri[hit], 6:9]
    rng = np.random.default_rng(2)
    d2 = rng.normal(size=p.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    for name, (oo, dd) in (("camera", (o, d)), ("bounce", (p.astype(np.float32), d2))):
        m = oo.shape[0]
        t = [torch.tensor(np.ascontiguousarray(x), device="cuda") for x in (oo[:, 0], oo[:, 1], oo[:, 2], dd[:, 0], dd[:, 1], dd[:, 2])]
        tmax = torch.full((m,), float("inf"), device="cuda")
        outs = [torch.empty(m, dtype=torch.int32, device="cuda") for _ in range(2)] + [torch.empty(m, device="cuda") for _ in range(2)]
        from psdr_cuda import _abi
        def run():
            _abi.check(g.lib, g.lib.psdr_trace(g.h, m, *[c.data_ptr() for c in t], tmax.data_ptr(), *[x.data_ptr() for x in outs], None))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%-12s %-7s %8.3f ms for %d rays = %7.1f Grays/s" % (scene, name, dt * 1e3, m, m / dt / 1e9))
