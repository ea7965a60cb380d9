#!/usr/bin/env python
"""Developer tool (CPU only): SIMD efficiency of the BVH walk under different wave execution models.
usage: python tools/simd_sim/run.py [scene] [npix]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import oracle
from helpers import load_scene
from psdr_cuda.scene import make_desc
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libsimd_sim.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(here, "simd_sim.cpp"), "-o", so])
L = C.CDLL(so)
scene = sys.argv[1] if len(sys.argv) > 1 else "cbox_bunny"
npix = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
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
    nrm = info[t, 18:21]
    return ok, p.astype(np.float32), nrm

def cos_dirs(nrm, win):
    nrm = np.where((np.sum(nrm * win, 1) > 0)[:, None], -nrm, nrm)       # face the arriving ray
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t1 = np.cross(nrm, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True); t2 = np.cross(nrm, t1)
    r1, r2 = rng.random(len(nrm)), rng.random(len(nrm))
    r, ph = np.sqrt(r1), 2 * np.pi * r2
    loc = np.stack([r * np.cos(ph), r * np.sin(ph), np.sqrt(np.maximum(0, 1 - r1))], 1)
    return (t1 * loc[:, 0:1] + t2 * loc[:, 1:2] + nrm * loc[:, 2:3]).astype(np.float32)

def light_dirs(p):
    # emitter 0: uniform point on its first triangles
    ei = tbc["emitter_i"].numpy().reshape(-1, 4)[0]
    f = rng.integers(0, ei[2], len(p)) + ei[1]
    a, b = rng.random(len(p)), rng.random(len(p)); t = np.sqrt(a)
    q = info[f, 0:3] + (1 - t)[:, None] * info[f, 3:6] + (t * b)[:, None] * info[f, 6:9]
    dd = q - p; dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    return dd.astype(np.float32)

def sim(name, o, d, R=2, mib=0):
    out = (C.c_double * 16)()
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    L.simd_sim(C.byref(desc), len(o), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_double(46.0), C.c_double(40.0), C.c_double(12.0), R, mib, out)
    print("%-22s vote: R=0 %6.0f  R=%d %6.0f | thresh .5: R=0 %6.0f R=%d %6.0f" % (name, out[9], R, out[10], out[11], R, out[12]))
    print("%-22s steps/ray %5.1f leaf-tris/ray %4.1f outer-iters/ray %4.1f useful/ray %6.0f | A: eff %.3f cost/64rays %7.0f | B(R=%d): cost/64rays %7.0f (%.2fx)"
          % (name, out[1], out[2], out[3], out[0], out[4], out[5], R, out[6], out[5] / max(out[6], 1)))

def sim2(name, o, d, bsz=512):
    out = (C.c_double * 16)()
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
    L.simd_sim_two_level(C.byref(desc), len(o), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_double(46.0), C.c_double(40.0), C.c_double(12.0),
                         bsz, 64, C.c_double(27.0), C.c_double(16.0), C.c_double(30.0), C.c_double(60.0), out)
    print("%-22s two-level bsz %4d: cost/64rays %6.0f  (phase1 %4.0f per wave; deferred items/ray %.2f; phase-2 waves per batch %.2f; inline tris %d, BLAS %d)"
          % (name, bsz, out[0], out[1], out[2], out[3], out[4], out[5]))

L.simd_sim_two_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
L.simd_sim.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
desc.num_guide_cells = -12345 if os.environ.get("DUMP") else desc.num_guide_cells
ok0, p0, n0 = hits(o0, d0)
p0 = p0 + 0  # primary hit points
d1 = cos_dirs(n0, d0); l1 = light_dirs(p0)
ok1, p1, n1 = hits(p0, d1)
d2 = cos_dirs(n1, d1); l2 = light_dirs(p1)
if os.environ.get("TWO"):
    ok0, p0, n0 = hits(o0, d0); d1 = cos_dirs(n0, d0); l1 = light_dirs(p0); ok1, p1, n1 = hits(p0, d1); d2 = cos_dirs(n1, d1); l2 = light_dirs(p1)
    for name, (oo, dd) in (("camera", (o0, d0)), ("bounce1 bsdf", (p0, d1)), ("bounce1 light", (p0, l1)), ("bounce2 bsdf", (p1, d2)), ("bounce2 light", (p1, l2))):
        sim(name, oo, dd, 4)
        for bsz in (256, 512, 1024):
            sim2(name, oo, dd, bsz)
    sys.exit(0)
for R in (4,):
    sim("camera", o0, d0, R); sim("bounce1 bsdf", p0, d1, R); sim("bounce1 light", p0, l1, R)
    sim("bounce2 light", p1, l2, R); sim("bounce2 bsdf", p1, d2, R)
    if os.environ.get("DUMP"):
        c = np.fromfile("/tmp/raycost.bin", np.float32)
        print("bounce2 bsdf ray cost percentiles", np.percentile(c, [10, 25, 50, 75, 90, 95, 99, 100]).round(0), "mean", c.mean())
        w = c.reshape(-1, 64)
        print("per-wave max: mean %.0f; mean of per-wave mean %.0f" % (w.max(1).mean(), w.mean(1).mean()))
        # sort each 512-ray batch by cost (oracle binning) and re-price as max per wave
        b = np.sort(c.reshape(-1, 512), axis=1).reshape(-1, 64)
        print("oracle-sorted 512 batches: mean per-wave max %.0f" % b.max(1).mean())
        b = np.sort(c.reshape(-1, 256), axis=1).reshape(-1, 64)
        print("oracle-sorted 256 batches: mean per-wave max %.0f" % b.max(1).mean())
        sys.exit(0)
    # both rays of a vertex in one batch (interleaved per 256 lanes)
    def inter(a, b):
        x = np.empty((len(a) * 2, 3), np.float32)
        x.reshape(-1, 2, 256, 3)[:, 0] = a.reshape(-1, 256, 3); x.reshape(-1, 2, 256, 3)[:, 1] = b.reshape(-1, 256, 3)
        return x
    sim("b1 bsdf+light", inter(p0, p0), inter(d1, l1), R); sim("b2 bsdf+light", inter(p1, p1), inter(d2, l2), R)

# ---- ray sorting experiment (SORT=1): the same bounce rays, ordered by a key of (origin cell, direction) before they are cut into waves
if os.environ.get("SORT"):
    def morton3(x, y, z, bits):
        k = np.zeros(len(x), np.uint64)
        for b in range(bits):
            k |= ((x >> b) & 1).astype(np.uint64) << np.uint64(3 * b) | ((y >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 1) | ((z >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 2)
        return k
    def sort_rays(o, d, ob, db, dir_first=False):
        lo, hi = o.min(0), o.max(0)
        q = np.minimum(((o - lo) / np.maximum(hi - lo, 1e-6) * (1 << ob)).astype(np.int64), (1 << ob) - 1)
        ko = morton3(q[:, 0], q[:, 1], q[:, 2], ob)
        qd = np.minimum(((d * 0.5 + 0.5) * (1 << db)).astype(np.int64), (1 << db) - 1)
        kd = morton3(qd[:, 0], qd[:, 1], qd[:, 2], db)
        key = (kd << np.uint64(3 * ob)) | ko if dir_first else (ko << np.uint64(3 * db)) | kd
        idx = np.argsort(key, kind="stable")
        return o[idx], d[idx]
    for name, (oo, dd) in (("bounce1 bsdf", (p0, d1)), ("bounce2 bsdf", (p1, d2)), ("bounce2 light", (p1, l2))):
        for ob, db, df in ((0, 0, False), (1, 1, False), (2, 1, False), (2, 2, False), (3, 1, False), (3, 2, False), (2, 1, True), (1, 2, True)):
            if ob == 0:
                sim(name + " unsorted", oo, dd, 4)
            else:
                so_, sd_ = sort_rays(oo, dd, ob, db, df)
                sim("%s o%d d%d %s" % (name, ob, db, "dir-first" if df else "org-first"), so_, sd_, 4)

# ---- the same within the EXPENSIVE class only (SORT2=1): rays that enter the box of the large mesh (what the class-binned wavefront already separates)
if os.environ.get("SORT2"):
    tm = tbc["tri_mesh"].numpy() & ~0x40000000
    big = np.bincount(tm).argmax()
    rows = info[tm == big]
    vv = np.concatenate([rows[:, 0:3], rows[:, 0:3] + rows[:, 3:6], rows[:, 0:3] + rows[:, 6:9]])
    blo, bhi = vv.min(0), vv.max(0)
    def enters(o, d):
        inv = 1.0 / np.where(d == 0, 1e-30, d)
        t0, t1 = (blo - o) * inv, (bhi - o) * inv
        tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
        return (tf >= np.maximum(tn, 0))
    for name, (oo, dd) in (("bounce1 bsdf", (p0, d1)), ("bounce2 bsdf", (p1, d2)), ("bounce2 light", (p1, l2))):
        m = enters(oo, dd)
        oo, dd = oo[m], dd[m]
        n64 = len(oo) // 64 * 64
        oo, dd = oo[:n64], dd[:n64]
        print("%s: %d rays enter the box (%.1f %%)" % (name, n64, 100.0 * m.mean()))
        for ob, db, df in ((0, 0, False), (1, 1, False), (2, 1, False), (2, 2, False), (3, 2, False), (4, 3, False), (1, 2, True), (2, 3, True)):
            if ob == 0:
                sim(name + " unsorted", oo, dd, 4)
            else:
                so_, sd_ = sort_rays(oo, dd, ob, db, df)
                sim("%s o%d d%d %s" % (name, ob, db, "dir-first" if df else "org-first"), so_, sd_, 4)
