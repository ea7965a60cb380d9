"""Occluder rows of a scene without a tree (round 6; csrc/psdr_bvh_build.h tiny_occluder_rows, psdr_device.h closest_hit MASKED).

On a scene whose primitives all travel in the kernel arguments (the Cornell box of BASELINE configs 1 / 2) a light ray -- from a path vertex on
triangle r to a sampled point of emitter triangle e -- tests only the rows of the primitive table that CAN lie between the two: e's own primitive
plus every primitive whose plane separates a corner of r's primitive from a corner of e's.  The table is conservative by construction; these tests
check it against brute force on the host: whatever the FULL closest-hit search (every primitive, tests/hostcheck = the product's closest_hit) finds
along such a segment lies in a row the table names.  The GPU half (same image with the table and with all-ones rows) is tests/test_options_gpu.py.
"""
import ctypes as C

import numpy as np

from helpers import hostcheck_lib, load_scene
from psdr_cuda.scene import make_desc


def occluder_rows(tb):
    H = hostcheck_lib()
    tbc = {k: (v.detach().cpu() if hasattr(v, "detach") else v) for k, v in tb.items()}
    desc, keep = make_desc(tbc, None, device="cpu")
    T = int(tb["num_tris"])
    occ = np.zeros(T * T, np.uint32)
    row = np.full(T, -1, np.int32)
    rc = H.hostcheck_occluder_rows(C.byref(desc), C.c_void_p(occ.ctypes.data), C.c_void_p(row.ctypes.data))
    assert rc == 0
    return occ.reshape(T, T), row, (desc, keep)


def emitter_tris(tb):
    ei = tb["emitter_i"].detach().cpu().numpy().reshape(-1, 4)
    out = []
    for e in ei:
        out += list(range(int(e[1]), int(e[1]) + int(e[2])))
    return out


def tri_points(tb, t, n, rng):
    r = tb["tri_info"].detach().cpu().numpy().reshape(-1, 24)[t]
    a, b = rng.random(n), rng.random(n)
    flip = a + b > 1
    a[flip], b[flip] = 1 - a[flip], 1 - b[flip]
    return r[0:3] + a[:, None] * r[3:6] + b[:, None] * r[6:9]


def test_convex_room_light_rays_test_the_light_alone():
    """cbox: five walls and the light, six axis-aligned parallelograms.  No wall's plane separates another wall from the light: every entry of an
    emitter column is the light's row alone; the columns of non-emitter triangles stay all ones (never looked at)."""
    sc, _ = load_scene("cbox", res=8, spp=1)
    tb = sc.tables(0)
    occ, row, _ = occluder_rows(tb)
    em = emitter_tris(tb)
    assert len(em) == 2 and row[em[0]] == row[em[1]] >= 0
    light = np.uint32(1 << int(row[em[0]]))
    T = int(tb["num_tris"])
    for r in range(T):
        for e in range(T):
            # (also from the light's own triangles -- rays IN the light's plane meet nothing of it, and with one emitter primitive nothing behind can count)
            assert occ[r, e] == (light if e in em else np.uint32(0xffffffff)), (r, e, hex(int(occ[r, e])))


def test_an_occluder_appears_exactly_where_its_plane_separates():
    """cbox_occluder: a quad hangs between the light and the floor.  Rows: its plane separates the floor (and the lower parts of the walls) from the light,
    so those entries carry its bit; it can never hide the light from the ceiling, which lies on the light's side of it."""
    sc, _ = load_scene("cbox_occluder", res=8, spp=1)
    tb = sc.tables(0)
    occ, row, _ = occluder_rows(tb)
    em = emitter_tris(tb)
    rows_seen = {int(occ[r, e]) for r in range(int(tb["num_tris"])) for e in em}
    assert len(rows_seen) > 1                                   # not one mask for the whole room any more
    assert all(m & (1 << int(row[em[0]])) for m in rows_seen)  # the light's own row is always tested


def _trace(H, desc, o, d, rows=None):
    n = len(o)
    tri = np.zeros(n, np.int32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); t = np.zeros(n, np.float32)
    rows = np.full(n, 0xffffffff, np.uint32) if rows is None else np.ascontiguousarray(rows, np.uint32)
    rc = H.hostcheck_trace_rows(C.byref(desc), n, C.c_void_p(o.ctypes.data), C.c_void_p(d.ctypes.data), C.c_void_p(rows.ctypes.data), C.c_void_p(tri.ctypes.data),
                                C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data), C.c_void_p(t.ctypes.data))
    assert rc == 0
    return tri, u, v, t


def test_masked_search_gives_the_light_rays_the_outcome_of_the_full_search():
    """Brute force: 200 random segments per (triangle, emitter triangle) pair, traced by the product's closest_hit on the host with every row and with the
    rows the table names.  What a light ray's hit is used for (direct.cpp:138-141: the hit must be an emitter at the sample's distance or beyond): wherever
    the full search ends on an EMITTER triangle the masked search returns the same hit bit for bit; wherever something lies IN FRONT of the sampled point the
    masked search finds it too (same triangle); a full-search hit on a non-emitter BEHIND the point (the ray missed the light) may be anything in the masked
    search but an emitter.  On the convex box, the box with an occluder, and from the light's own triangles (rays in the light's plane)."""
    H = hostcheck_lib()
    rng = np.random.default_rng(7)
    for name in ("cbox", "cbox_occluder"):
        sc, _ = load_scene(name, res=8, spp=1)
        tb = sc.tables(0)
        occ, row, (desc, keep) = occluder_rows(tb)
        T = int(tb["num_tris"])
        em = set(emitter_tris(tb))
        checked = in_front = 0
        for e in em:
            for r in range(T):
                n = 200
                o, p = tri_points(tb, r, n, rng).astype(np.float32), tri_points(tb, e, n, rng).astype(np.float32)
                d = p - o
                ln = np.linalg.norm(d, axis=1)
                ok = ln > 1e-2
                d = (d / np.maximum(ln, 1e-30)[:, None]).astype(np.float32)
                full = _trace(H, desc, o, d)
                part = _trace(H, desc, o, d, np.full(n, occ[r, e], np.uint32))
                for i in np.nonzero(ok)[0]:
                    ft, pt = int(full[0][i]), int(part[0][i])
                    if ft in em:
                        assert pt == ft and part[1][i] == full[1][i] and part[2][i] == full[2][i] and part[3][i] == full[3][i], (name, r, e, ft, pt)
                    elif ft >= 0 and full[3][i] < ln[i] - 1e-2:
                        assert pt == ft, (name, r, e, ft, pt)          # an occluder in front of the sample: found either way
                        in_front += 1
                    else:
                        assert pt not in em, (name, r, e, ft, pt)
                    checked += 1
        assert checked > 1000 and (name == "cbox" or in_front > 100), (name, checked, in_front)
