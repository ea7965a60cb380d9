#!/usr/bin/env python
"""Is a render call reproducible?  cbox_bunny PathTracer(3) through the C ABI, every launch form twice on the same handle and on a fresh handle:
number of pixels that differ by more than 1e-5 relative, largest difference (developer tool, round 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene
from psdr_cuda import _abi
res, spp = 256, 32
sc, _ = load_scene("cbox_bunny", res=res, spp=spp, sppe=0, sppse=0)
tb = sc.tables(0)
def cmp(tag, a, b):
    d = np.abs(a - b); s = np.maximum(np.abs(a), 1e-3)
    bad = (d / s > 1e-5)
    i = np.unravel_index(np.argmax(d), d.shape)
    print("%-46s pixels*channels off by > 1e-5 rel: %6d   max |diff| %.3e at %s (value %.4f)" % (tag, int(bad.sum()), d[i], i, a[i]), flush=True)
forms = [("fused", _abi.FLAG_FUSED, {}), ("wavefront traced", _abi.FLAG_WAVEFRONT, {}), ("wavefront traced, one trace workgroup per CU", _abi.FLAG_WAVEFRONT, {"trace_wg2": 0}),
         ("wavefront binned (no trace kernel)", _abi.FLAG_WAVEFRONT, {"wf_traced": 0})]
ref = None
for name, fl, opts in forms:
    try:
        g = GpuScene(tb, options=opts)
    except Exception as e:
        print(name, "option not available:", e); continue
    o = _abi.make_opts(spp=spp, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=fl)
    a = g.render_c(o); b = g.render_c(o)
    cmp(name + ": two calls, one handle", a, b)
    g2 = GpuScene(tb, options=opts)
    c = g2.render_c(o)
    cmp(name + ": fresh handle", a, c)
    if ref is None: ref = a
    else: cmp(name + " vs fused", a, ref)
