#!/usr/bin/env python
"""Developer tool (CPU only): statistics of the 4-wide tree walk on the product's forest (node counts, stack depth, LDS coverage).
usage: python tools/walk_stats/run.py [cbox_bunny|interior|...] [npix]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import oracle
from helpers import load_scene
from psdr_cuda.scene import make_desc
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libwalk_stats.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(here, "walk_stats.cpp"), "-o", so])
L = C.CDLL(so)
scene = sys.argv[1] if len(sys.argv) > 1 else "cbox_bunny"
npix = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if scene == "interior":
    from psdr_cuda.fixtures import make_interior_scene
    sc = make_interior_scene(seed=0, n_objects=10, res=256, spp=1); sc.configure()
else:
    sc, _ = load_scene(scene, res=256)
tb = sc.tables(0)
tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
desc, keep = make_desc(tbc, None, device="cpu")
rng = np.random.default_rng(0)
cam = tbc["cam"].numpy().astype(np.float64)
s2c, tw = cam[0:16].reshape(4, 4), cam[16:32].reshape(4, 4)
pix = rng.integers(0, 256 * 256, npix)
px = np.repeat(pix % 256, 64); py = np.repeat(pix // 256, 64)
n = npix * 64
s = np.stack([(px + rng.random(n)) / 256, (py + rng.random(n)) / 256], 1)
v = np.concatenate([s, np.zeros((n, 1)), np.ones((n, 1))], 1) @ s2c.T
d = v[:, :3] / v[:, 3:4]; d /= np.linalg.norm(d, axis=1, keepdims=True)
d0 = (d @ tw[:3, :3].T).astype(np.float32); o0 = np.broadcast_to(tw[:3, 3], (n, 3)).astype(np.float32)
info = tbc["tri_info"].numpy()

def hits(o, d):
    shape, tri, u, vv = oracle.trace(tbc, o, d)
    ok = tri >= 0
    t = np.where(ok, tri, 0)
    p = info[t, 0:3] + u[:, None] * info[t, 3:6] + vv[:, None] * info[t, 6:9]
    return ok, p.astype(np.float32), info[t, 18:21]

def cos_dirs(nrm, win):
    nrm = np.where((np.sum(nrm * win, 1) > 0)[:, None], -nrm, nrm)
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t1 = np.cross(nrm, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True); t2 = np.cross(nrm, t1)
    r1, r2 = rng.random(len(nrm)), rng.random(len(nrm))
    r, ph = np.sqrt(r1), 2 * np.pi * r2
    loc = np.stack([r * np.cos(ph), r * np.sin(ph), np.sqrt(np.maximum(0, 1 - r1))], 1)
    return (t1 * loc[:, 0:1] + t2 * loc[:, 1:2] + nrm * loc[:, 2:3]).astype(np.float32)

L.walk_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
def stats(name, o, d, sorted_=1):
    out = (C.c_double * 64)()
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    rc = L.walk_stats(C.byref(desc), len(o), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), sorted_, out)
    assert rc == 0
    print("%-14s %s: BVH2 nodes %d, 4-wide nodes %d (%.0f KB), worst-case stack %d, leaf triangles %d" % (name, "sorted" if sorted_ else "nearest-first", out[0], out[1], out[1] * 64 / 1024, out[2], out[3]))
    print("   rays into a tree %d of %d: node visits %.1f, leaf visits %.1f, triangle tests %.1f per ray" % (out[4], len(o), out[5], out[6], out[7]))
    h = np.array(out[8:40]); cum = np.cumsum(h)
    print("   deepest stack per ray: " + " ".join("%d:%.3f" % (i, cum[i]) for i in range(32) if h[i] > 0 or i < 4))
    print("   share of node visits in the first N nodes: " + " ".join("%s:%.3f" % (n_, out[40 + i]) for i, n_ in enumerate(("128", "256", "512", "1k", "2k", "4k", "8k", "all"))))

ok0, p0, n0 = hits(o0, d0)
d1 = cos_dirs(n0, d0)
ok1, p1, n1 = hits(p0, d1)
d2 = cos_dirs(n1, d1)
for srt in (1, 0):
    stats("camera", o0, d0, srt); stats("bounce1", p0, d1, srt); stats("bounce2", p1, d2, srt)
