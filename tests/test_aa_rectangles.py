"""Axis-aligned rectangles among the kernel-argument primitives travel in SLAB FORM (psdr_bvh_build.h tiny_plane_form, psdr_device.h
aa_prim_test: three slots per axis, the hit's in-plane offsets mapped to the hit triangle's barycentrics by the rows setup_lds /
tiny_hit_row derive from the pair's codes and the rectangle's scale).  Every orientation a mesh can present -- normal along +-x / y / z,
either winding, any starting corner, either diagonal, any rotation of a face's index triple, a fourth rectangle of an axis (stays in plane
form), rectangles next to lone triangles -- must give the oracle's triangle and barycentrics (the replacement of
cuda/psdr_cuda.cu:9-45 for such scenes)."""
import ctypes as C

import numpy as np
import pytest

import oracle
import psdr_cuda
from helpers import hostcheck_lib, make_desc
from psdr_cuda.scene import look_at


def rect_scene(seed, n_rect, extra_tris=0, same_axis=None):
    rng = np.random.default_rng(seed)
    verts, faces = [], []
    for r in range(n_rect):
        axis = int(rng.integers(0, 3)) if same_axis is None else same_axis
        a, b = [k for k in range(3) if k != axis]
        # corners on a grid of 1/32: a pair becomes ONE primitive only when its fourth corner is exact in fp32 (pack_tiny_prims)
        c = np.round(rng.uniform(-1, 1, 3) * 32) / 32
        ha, hb = np.round(rng.uniform(0.2, 1.5, 2) * 32) / 32
        corners = []
        for sa, sb in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            p = c.copy(); p[a] += sa * ha; p[b] += sb * hb
            corners.append(p)
        start, flip = int(rng.integers(0, 4)), bool(rng.integers(0, 2))
        order = [(start + k) % 4 for k in range(4)]
        if flip:
            order = order[::-1]
        q = [corners[k] for k in order]
        base = len(verts)
        verts += q
        tris = [(0, 1, 2), (0, 2, 3)] if rng.integers(0, 2) else [(0, 1, 3), (1, 2, 3)]
        for t in tris:
            rot = int(rng.integers(0, 3))
            faces.append([base + t[(rot + k) % 3] for k in range(3)])
    for _ in range(extra_tris):
        c = rng.uniform(-1, 1, (1, 3))
        base = len(verts)
        verts += list(c + 0.7 * rng.normal(size=(3, 3)))
        faces.append([base, base + 1, base + 2])
    return np.asarray(verts, np.float32), np.asarray(faces, np.int32)


def make_scene(verts, faces):
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = 8
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 1, 0, 0, 0
    cam = psdr_cuda.PerspectiveCamera(40.0, 0.1, 1e3)
    cam.to_world = look_at([0, 0, 5], [0, 0, 0], [0, 1, 0])
    sc.add_sensor(cam)
    b = psdr_cuda.Diffuse([0.5, 0.6, 0.7]); b.id = "b"
    sc.add_bsdf(b)
    m = psdr_cuda.Mesh()
    m.use_face_normals = True
    m.set_geometry(verts, faces)
    sc.add_mesh(m, b, emitter_radiance=[3.0, 2.0, 1.0])
    sc.finalize()
    sc.configure()
    return sc


def rays(seed, m):
    rng = np.random.default_rng(seed + 100)
    o = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    target = rng.uniform(-1.5, 1.5, (m, 3)).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # rays along an axis (a zero direction component: the reciprocal is infinite) and rays that start ON a rectangle's plane
    d[:50] = np.eye(3, dtype=np.float32)[np.arange(50) % 3] * np.where(np.arange(50) % 2, 1.0, -1.0)[:, None].astype(np.float32)
    return o, np.ascontiguousarray(d, dtype=np.float32)


def host_trace(tb, o, d):
    H = hostcheck_lib()
    import torch
    tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
    desc, keep = make_desc(tbc, None, device="cpu")
    m = o.shape[0]
    tri = np.zeros(m, np.int32); u = np.zeros(m, np.float32); v = np.zeros(m, np.float32)
    oo, dd = np.ascontiguousarray(o), np.ascontiguousarray(d)
    lay = (C.c_int * 4)()
    assert H.hostcheck_tiny_layout(C.byref(desc), lay) == 0
    rc = H.hostcheck_trace(C.byref(desc), m, C.c_void_p(oo.ctypes.data), C.c_void_p(dd.ctypes.data), C.c_void_p(tri.ctypes.data),
                           C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data))
    assert rc == 0
    return tri, u, v, list(lay)


def compare(name, tri, u, v, tri_ref, u_ref, v_ref, frac=0.998):
    same = tri == tri_ref
    assert same.mean() > frac, (name, same.mean())              # rays through the shared diagonal / a shared edge may pick the neighbour
    hit = same & (tri_ref >= 0)
    assert hit.sum() > 500, (name, hit.sum())
    du, dv = np.abs(u[hit] - u_ref[hit]).max(), np.abs(v[hit] - v_ref[hit]).max()
    assert du < 2e-5 and dv < 2e-5, (name, du, dv)
    assert set(np.unique(tri[hit])) == set(np.unique(tri_ref[hit]))


CASES = [(0, 6, 0, None), (1, 8, 0, None), (2, 3, 4, None), (3, 4, 0, 0), (4, 4, 2, 1), (5, 5, 1, 2), (6, 1, 0, None), (7, 7, 2, None)]


@pytest.mark.parametrize("seed,n_rect,extra,axis", CASES)
def test_slab_form_on_the_host_matches_the_oracle(seed, n_rect, extra, axis):
    """the product's closest_hit compiled for the host (tests/hostcheck): slab rows, plane rows and the hit-row decode, against the oracle's
    independent Moeller-Trumbore traversal"""
    verts, faces = rect_scene(seed, n_rect, extra, axis)
    tb = make_scene(verts, faces).tables(0)
    assert tb["num_tris"] <= 16
    o, d = rays(seed, 20000)
    tri, u, v, lay = host_trace(tb, o, d)
    # every rectangle sits in a slab slot, up to three per axis; what is left (a fourth of an axis, lone triangles) follows from row 9 on
    per_axis = [0, 0, 0]
    for r in range(n_rect):
        e = verts[faces[2 * r]]
        per_axis[int(np.argmin(np.ptp(e, axis=0)))] += 1
    want = [min(c, 3) for c in per_axis]
    assert lay[1:] == want and lay[0] == 9 + (n_rect - sum(want)) + extra, (lay, per_axis)
    _, tri_ref, u_ref, v_ref = oracle.trace(tb, o, d)
    compare("host", tri, u, v, tri_ref, u_ref, v_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rect,extra,axis", CASES)
def test_slab_form_on_the_gpu_matches_the_oracle_and_the_plane_form(seed, n_rect, extra, axis):
    from helpers import GpuScene
    verts, faces = rect_scene(seed, n_rect, extra, axis)
    tb = make_scene(verts, faces).tables(0)
    o, d = rays(seed, 50000)
    _, tri_ref, u_ref, v_ref = oracle.trace(tb, o, d)
    out = {}
    for aa in (1, 0):
        g = GpuScene(tb, options={"aa_prims": aa})
        _, tri, u, v = g.trace(o, d)
        compare("gpu aa=%d" % aa, tri, u, v, tri_ref, u_ref, v_ref)
        out[aa] = (tri, u, v)
    compare("gpu slab vs plane", *out[1], *out[0], frac=0.999)
