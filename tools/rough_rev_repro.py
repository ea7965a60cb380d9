#!/usr/bin/env python
"""Reproducer of the order-dependent camera-pose gradient: cbox_rough, PathTracer(3), reverse mode with every gradient, against forward mode.
usage: PSDR_HIP_LIB=variants/lib_dppall.so python tools/rough_rev_repro.py [history] [poison pattern hex] [what]
  history: none | bvh (a device-built tree + a trace, as tests/test_device_bvh_gpu.py does) | big (a large-scratch kernel: C5 reverse)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import GpuScene, load_scene, dot_tables, camera_rays
from psdr_cuda import _abi
import test_reverse_mode as trm
history = sys.argv[1] if len(sys.argv) > 1 else "none"
pattern = int(sys.argv[2], 16) if len(sys.argv) > 2 else None
what = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if history == "bvh":
    os.environ["PSDR_OPTIONS"] = "bvh_build=1"
    sc, _ = load_scene("cbox_bunny", res=64)
    g = GpuScene(sc.tables(0))
    o, d = camera_rays(sc.tables(0), 100000, seed=1)
    g.trace(o, d)
    os.environ.pop("PSDR_OPTIONS")
elif history == "big":
    from psdr_cuda.fixtures import make_interior_scene
    sc = make_interior_scene(seed=0, n_objects=4, res=64, spp=4); sc.configure()
    g = GpuScene(sc.tables(0))
    adj = np.ones((64 * 64, 3), np.float32)
    g.render_d_rev(_abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=4), adj, want=["tri_info", "texels", "cam_to_world"], with_image=False)
if pattern is not None:
    P = C.CDLL(os.path.join(ROOT, "tests", "poison", "libpoison.so"))
    rc = P.poison_gpu(C.c_uint32(pattern), what)
    assert rc == 0, rc
tb, o, adj = trm._setup("cbox_rough", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3), 0, 0, res=32, spp=8)
g = GpuScene(tb)
names = ["texels", "emitter_rad", "tri_info", "cam_to_world"]
if len(sys.argv) > 4 and sys.argv[4] == "diff":
    # which gradient words move with the scratch pattern: zero-filled scratch against the pattern, same process
    P = C.CDLL(os.path.join(ROOT, "tests", "poison", "libpoison.so"))
    assert P.poison_gpu(C.c_uint32(0), 1) == 0
    _, ga = g.render_d_rev(o, adj, want=names)
    assert P.poison_gpu(C.c_uint32(pattern), 1) == 0
    _, gb = g.render_d_rev(o, adj, want=names)
    for n in names:
        a_, b_ = np.asarray(ga[n], np.float64).ravel(), np.asarray(gb[n], np.float64).ravel()
        bad = np.nonzero(~np.isclose(a_, b_, rtol=1e-3, atol=1e-3 * np.abs(a_).max()))[0]
        print("%-14s words %6d  moved %5d  first %s  |clean| max %.3e  max diff %.3e" % (n, a_.size, bad.size, bad[:12].tolist(), np.abs(a_).max(), np.nanmax(np.abs(a_ - b_)) if bad.size else 0.0))
        if n == "cam_to_world":
            print("   clean ", a_); print("   poison", b_)
    sys.exit(0)
_, grads = g.render_d_rev(o, adj, want=names)
tan = trm._tangents(tb, "cam_to_world")
_, dimg = g.render_d_fwd(o, [tan])
lhs, rhs = float((adj.astype(np.float64) * dimg[0]).sum()), dot_tables(grads, tan)
print("history=%s poison=%s what=%d: forward %.6e reverse %.6e -> %s" % (history, sys.argv[2] if pattern is not None else "-", what, lhs, rhs, "OK" if abs(lhs - rhs) < 1e-3 * abs(lhs) + 1 else "WRONG"))
print("   cam gradient", np.asarray(grads["cam_to_world"]).ravel()[:12])
