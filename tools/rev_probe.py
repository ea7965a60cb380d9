import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene
from psdr_cuda import _abi
from psdr_cuda.fixtures import make_interior_scene
res, spp = 512, 16
sc = make_interior_scene(seed=0, n_objects=10, res=res, spp=spp); sc.configure()
tb = sc.tables(0); g = GpuScene(tb)
adj = np.random.default_rng(0).random((res * res, 3)).astype(np.float32)
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for depth in (1, 2, 3):
    o = _abi.make_opts(spp=spp, integrator=_abi.INTEGRATOR_PATH, max_depth=depth)
    for want in (["texels"], ["tri_info"], ["tri_info", "texels"], ["emitter_rad"]):
        print("depth", depth, want, "%.2f ms" % t(lambda: g.render_d_rev(o, adj, want=want, with_image=False)))
o = _abi.make_opts(spp=spp, bsdf_samples=1, light_samples=1)
print("direct", "%.2f ms" % t(lambda: g.render_d_rev(o, adj, want=["tri_info", "texels"], with_image=False)))
