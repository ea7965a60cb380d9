"""round 6: is the occluder-row table active, and what does it buy?  C2 renderC / renderD K = 1 with occ_rows 1 / 0 inside one call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
sc, _ = load_scene("cbox", res=512, spp=64)
tb = sc.tables(0)
o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=64)
for rows in (1, 0, 1, 0):
    g = GpuScene(tb, options={"occ_rows": rows})
    g.render_c(o)
    print("occ_rows", rows, _abi.scene_stats(g.h))
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.render_c(o); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("   render_c ms (incl. the image copy back)", sorted(ts)[2], "rays", g.counters()[0])
    g.close()
