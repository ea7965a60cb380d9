#!/usr/bin/env python
"""fused vs wavefront on the interior scene: how far apart, where (developer probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from helpers import GpuScene, load_scene, rel_l2
from psdr_cuda import _abi
from psdr_cuda.fixtures import make_interior_scene
import oracle
sc = make_interior_scene(seed=0, n_objects=10, res=96, spp=16); sc.configure()
tb = sc.tables(0); g = GpuScene(tb)
for depth in (1, 2, 3):
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=16, rng_offset=(5, 0, 0))
    a = g.render_c(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw)); b = g.render_c(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw))
    r = oracle.render(tb, _abi.make_opts(**kw))
    rel = np.abs(a - b).max(1) / (1.0 + np.abs(a).max(1))
    print("depth %d: fused-wf rel-L2 %.2e; per-pixel rel diff quantiles 50/90/99/99.9/max: %s; fused-oracle %.2e wf-oracle %.2e"
          % (depth, rel_l2(b, a), np.quantile(rel, [.5, .9, .99, .999, 1.0]), rel_l2(a, r), rel_l2(b, r)))
    ro = np.abs(a - r).max(1) / (1.0 + np.abs(r).max(1)); rw = np.abs(b - r).max(1) / (1.0 + np.abs(r).max(1))
    print("   pixels > 1e-5: fused-wf %d  fused-oracle %d  wf-oracle %d of %d" % ((rel > 1e-5).sum(), (ro > 1e-5).sum(), (rw > 1e-5).sum(), rel.size))
