import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for scene, spp in (("cbox", 64), ("cbox_bunny", 16), ("bunny_light", 16), ("bunny_env", 16), ("cbox_env", 16)):
    sc, _ = load_scene(scene, res=512, spp=spp); tb = sc.tables(0); g = GpuScene(tb)
    for d in (3, 6):
        row = []
        for name, fl in (("fused", _abi.FLAG_FUSED), ("wavefront", _abi.FLAG_WAVEFRONT)):
            o = _abi.make_opts(spp=spp, integrator=_abi.INTEGRATOR_PATH, max_depth=d, flags=fl)
            ms = t(lambda: g.render_c(o)); r = g.counters()[0]
            row.append("%s %6.2f ms" % (name, ms))
        print("%-12s depth %d  rays/slot %.2f of %d   %s" % (scene, d, r / (512 * 512 * spp), 1 + 2 * d, "   ".join(row)), flush=True)
