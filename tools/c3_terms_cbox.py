#!/usr/bin/env python
"""C3-style workload on cbox_bunny (512x512, spp = sppe = sppse = 16, translation of the bunny): forward and reverse passes; run under
`rocprofv3 --kernel-trace --stats` for the cost of each term's kernel (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene, tangents_wrt
from psdr_cuda import _abi
sc, P = load_scene("cbox_bunny", res=512, spp=16, sppe=16, sppse=16, translate=(1, (1.0, 0.0, 0.0)))
tb = sc.tables(0); g = GpuScene(tb)
o = _abi.make_opts(spp=16, sppe=16, sppse=16, bsdf_samples=1, light_samples=1)
tan = tangents_wrt(tb, P)
adj = np.random.default_rng(0).random((512 * 512, 3)).astype(np.float32)
for _ in range(3):
    g.render_d_fwd(o, [tan])
    g.render_d_rev(o, adj, with_image=False)
print("done")
