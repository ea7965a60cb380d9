#!/usr/bin/env python
"""Runs ONE workload a few times (for rocprofv3): python tools/prof_case.py <scene> <integrator> <mode> [res spp depth]
scene: cbox | cbox_bunny | interior ; integrator: direct | path ; mode: c | fwd | rev"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene, random_tangents
from psdr_cuda import _abi
scene, integ, mode = sys.argv[1:4]
res = int(sys.argv[4]) if len(sys.argv) > 4 else 512
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 16
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 3
if scene == "interior":
    from psdr_cuda.fixtures import make_interior_scene
    sc = make_interior_scene(seed=0, n_objects=10, res=res, spp=spp); sc.configure()
else:
    sc, _ = load_scene(scene, res=res, spp=spp, sppe=spp if mode == "edges" else 0, sppse=spp if mode == "edges" else 0)
tb = sc.tables(0); g = GpuScene(tb)
if mode == "trace":          # k_trace on incoherent rays: cosine-distributed bounce rays from the primary hit points
    from helpers import camera_rays
    n = res * res * spp
    o0, d0 = camera_rays(tb, n, seed=1)
    _, tri, u, v = g.trace(o0, d0)
    info = tb["tri_info"].cpu().numpy()
    idx = np.nonzero(tri >= 0)[0]
    p = (info[tri[idx], 0:3] + u[idx, None] * info[tri[idx], 3:6] + v[idx, None] * info[tri[idx], 6:9]).astype(np.float32)
    d = np.random.default_rng(2).normal(size=p.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    for _ in range(3):
        g.trace(p, d)
    print("done trace", len(p))
    sys.exit(0)
if mode == "edges":          # DirectIntegrator renderD forward with both edge terms: k_primary_edge, k_secondary_edge_filter, k_secondary_edge
    o = _abi.make_opts(spp=spp, sppe=spp, sppse=spp)
    tan = random_tangents(tb, ["tri_info", "sec_edge", "prim_edge"])
    for _ in range(3):
        g.render_d_fwd(o, [tan])
    print("done edges", g.counters())
    sys.exit(0)
kw = dict(bsdf_samples=1, light_samples=1) if integ == "direct" else dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth)
if os.environ.get("PSDR_PROF_WAVEFRONT"):
    kw["flags"] = _abi.FLAG_WAVEFRONT if os.environ["PSDR_PROF_WAVEFRONT"] == "1" else _abi.FLAG_FUSED
o = _abi.make_opts(spp=spp, **kw)
adj = np.random.default_rng(0).random((res * res, 3)).astype(np.float32)
tan = random_tangents(tb, ["tri_info", "texels"])
for _ in range(3):
    if mode == "c": g.render_c(o)
    elif mode == "fwd": g.render_d_fwd(o, [tan])
    else: g.render_d_rev(o, adj, want=["tri_info", "texels"], with_image=False)
print("done", g.counters())
