"""Edge cases of the C ABI on the GPU: degenerate and minimal inputs, limits and error paths
(the reference asserts on most of these: integrator.cpp:65,74; scene.cpp:57,427)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
import psdr_cuda
from helpers import GpuScene, load_scene, rel_l2
from psdr_cuda import _abi
from psdr_cuda.scene import look_at

pytestmark = pytest.mark.gpu


def tiny_scene(verts, faces, emitter=True, res=16, spp=4, extra=None):
    """one mesh (optionally emissive) + optional second mesh, seen from z = +5"""
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
    cam = psdr_cuda.PerspectiveCamera(40.0, 0.1, 1e3)
    cam.to_world = look_at([0, 0, 5], [0, 0, 0], [0, 1, 0])
    sc.add_sensor(cam)
    b = psdr_cuda.Diffuse([0.5, 0.6, 0.7]); b.id = "b"
    sc.add_bsdf(b)
    m = psdr_cuda.Mesh()
    m.use_face_normals = True
    m.set_geometry(np.asarray(verts, np.float32), np.asarray(faces, np.int32))
    sc.add_mesh(m, b, emitter_radiance=[3.0, 2.0, 1.0] if emitter else None)
    if extra is not None:
        m2 = psdr_cuda.Mesh()
        m2.use_face_normals = True
        m2.set_geometry(np.asarray(extra[0], np.float32), np.asarray(extra[1], np.int32))
        sc.add_mesh(m2, b)
    sc.finalize()
    sc.configure()
    return sc


TRI = ([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[0, 1, 2]])


def test_single_triangle_scene_root_is_a_leaf():
    tb = tiny_scene(*TRI).tables(0)
    assert tb["num_tris"] == 1
    for kw in (dict(bsdf_samples=1, light_samples=1), dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3),
               dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["depth"])):
        o = _abi.make_opts(spp=4, **kw)
        img, ref = GpuScene(tb).render_c(o), oracle.render(tb, o)
        assert np.isfinite(img).all() and rel_l2(img, ref) < 1e-5 and img.max() > 0


def test_two_meshes_emitter_facing_a_receiver():
    quad = ([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]], [[0, 1, 2], [0, 2, 3]])
    light = ([[-0.5, -0.5, 1], [0.5, -0.5, 1], [0.5, 0.5, 1], [-0.5, 0.5, 1]], [[0, 2, 1], [0, 3, 2]])     # faces -z, towards the quad
    sc = tiny_scene(light[0], light[1], True, extra=quad)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=4, bsdf_samples=1, light_samples=1)
    img, ref = GpuScene(tb).render_c(o), oracle.render(tb, o)
    assert rel_l2(img, ref) < 1e-5 and img.mean() > 0


def test_degenerate_triangle_is_never_hit_and_never_sampled():
    verts = [[-1, -1, 0], [1, -1, 0], [0, 1, 0], [2, 2, 0], [2, 2, 0], [2, 2, 0]]          # second face: zero area
    tb = tiny_scene(verts, [[0, 1, 2], [3, 4, 5]]).tables(0)
    assert float(tb["tri_info"][1, 21]) == 0.0
    g = GpuScene(tb)
    o = _abi.make_opts(spp=8, bsdf_samples=1, light_samples=1)
    img = g.render_c(o)
    assert np.isfinite(img).all() and rel_l2(img, oracle.render(tb, o)) < 1e-5
    rays_o = np.tile(np.array([[2.0, 2.0, 5.0]], np.float32), (64, 1))
    rays_d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (64, 1))
    _, tri, _, _ = g.trace(rays_o, rays_d)
    assert (tri == -1).all()


def test_scene_without_emitter_fails_like_the_reference():
    tb = tiny_scene(*TRI, emitter=False).tables(0)
    g = GpuScene(tb)
    with pytest.raises(RuntimeError, match="No Emitter!"):
        g.render_c(_abi.make_opts(spp=4, bsdf_samples=1, light_samples=1))
    img = g.render_c(_abi.make_opts(spp=4, integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["silhouette"]))     # needs no emitter
    assert 0.0 < img.mean() < 1.0


def test_limits_and_empty_work():
    sc, _ = load_scene("cbox", res=16, spp=4)
    tb = sc.tables(0)
    g = GpuScene(tb)
    with pytest.raises(RuntimeError, match="Too many samples"):
        g.render_c(_abi.make_opts(spp=2 ** 24))                   # 16*16*2^24 > INT_MAX, integrator.cpp:74
    with pytest.raises(RuntimeError, match="Invalid spp shard range"):
        g.render_c(_abi.make_opts(spp=4, spp_range=(2, 9)))
    with pytest.raises(RuntimeError, match="bsdf_samples \\+ light_samples"):
        g.render_c(_abi.make_opts(spp=4, bsdf_samples=0, light_samples=0))
    empty = g.render_c(_abi.make_opts(spp=4, spp_range=(2, 2)))   # a rank that owns no sample: zero image, no error
    assert empty.shape == (256, 3) and not empty.any()
    assert not g.render_c(_abi.make_opts(spp=0)).any()
    t = {"texels": torch.ones_like(tb["texels"])}
    with pytest.raises(RuntimeError, match="K must be 1 or 3"):
        g.render_d_fwd(_abi.make_opts(spp=4), [t, t])
    with pytest.raises(RuntimeError, match="max_depth > 250"):
        g.render_d_rev(_abi.make_opts(spp=4, integrator=_abi.INTEGRATOR_PATH, max_depth=251), np.ones((256, 3), np.float32), want=["texels"])


def test_null_arguments_and_unconfigured_handle():
    lib = _abi.load_hip()
    assert lib.psdr_render_c(None, None, None, None) != 0 and b"null argument" in lib.psdr_last_error()
    h = C.c_void_p()
    assert lib.psdr_scene_create(C.byref(h)) == 0
    o = _abi.make_opts(spp=1)
    buf = torch.zeros(3, device="cuda")
    assert lib.psdr_render_c(h, C.byref(o), buf.data_ptr(), None) != 0 and b"Scene not loaded yet!" in lib.psdr_last_error()
    assert lib.psdr_bvh_build(h, None) != 0
    assert lib.psdr_scene_destroy(h) == 0


def test_bvh_refit_between_configures_matches_a_rebuild():
    """an optimisation loop moves vertices, keeps the topology: the tree is refitted on the device
    (psdr_bvh_build) -- same hits as the oracle, and a rebuild once the boxes have degraded"""
    from helpers import camera_rays
    import enoki as ek
    from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
    lib = _abi.load_hip()
    sc, _ = load_scene("cbox_bunny", res=32, spp=4)
    integ = psdr_cuda.DirectIntegrator(1, 1)
    integ.renderC(sc)
    stats = (C.c_int32 * 4)()
    lib.psdr_bvh_stats(sc._native, stats)
    assert list(stats)[:2] == [1, 0] and stats[2] > 1000
    mesh = sc.param_map["Mesh[1]"]
    for step, shift in enumerate((2.0, 5.0, -3.0)):
        mesh.set_transform(Matrix4fD.translate(Vector3fD([shift, 0.5 * shift, 0.0])))
        sc.configure()
        img = integ.renderC(sc).numpy()
        lib.psdr_bvh_stats(sc._native, stats)
        assert stats[0] == 1 and stats[1] == step + 1                       # refitted, not rebuilt
        tb = sc.tables(0)
        o, d = camera_rays(tb, 50_000, seed=step)
        _, tri, u, v = GpuScene(tb).trace(o, d)                               # a fresh handle = a full build
        ref_tri = oracle.trace(tb, o, d)[1]
        assert (tri == ref_tri).mean() > 0.999
        # the refitted handle renders the same image as the oracle on the same streams
        ref = oracle.render(tb, _abi.make_opts(spp=4, bsdf_samples=1, light_samples=1, rng_offset=(7 * (step + 1), 0, 0)))
        bad = (np.abs(img - ref).max(1) > 1e-3 * (1 + np.abs(ref).max(1))).mean()
        assert bad < 0.01
    # blow the bunny up: the refitted boxes grow far beyond the build -> the next GEOMETRY configure rebuilds (the refit's area is read
    # back one call later, without a stall); a material-only configure in between keeps the tree as it is (no psdr_bvh_build at all)
    mesh.set_transform(Matrix4fD.scale(Vector3fD([3.0, 3.0, 3.0])))
    sc.configure(); integ.renderC(sc)
    lib.psdr_bvh_stats(sc._native, stats)
    before = list(stats)[:2]
    sc.configure(); integ.renderC(sc)
    lib.psdr_bvh_stats(sc._native, stats)
    assert list(stats)[:2] == before                                        # nothing but (unchanged) materials: neither rebuilt nor refitted
    mesh.set_transform(Matrix4fD.scale(Vector3fD([3.0, 3.0, 3.01])))
    sc.configure(); integ.renderC(sc)
    lib.psdr_bvh_stats(sc._native, stats)
    assert stats[0] == 2


@pytest.mark.parametrize("n_tris,seed", [(1, 0), (2, 1), (3, 2), (5, 3), (12, 8), (16, 9), (17, 4), (100, 5), (1000, 6), (5000, 7)])
def test_trace_fuzz_random_triangle_soups(n_tris, seed):
    """closest hit on random triangle soups (overlapping, sliver and tiny triangles, all leaf sizes of the
    SAH-terminated tree) against the oracle's independent traversal: same triangle, same barycentrics"""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1, 1, (n_tris, 1, 3))
    size = rng.choice([0.02, 0.2, 1.0], (n_tris, 1, 1))
    tri = centres + size * rng.normal(size=(n_tris, 3, 3))
    if n_tris > 10:
        tri[3, 2] = tri[3, 0] + 1e-7 * (tri[3, 1] - tri[3, 0])          # a sliver
        tri[4] = tri[4, :1]                                              # a degenerate point triangle
    verts = tri.reshape(-1, 3).astype(np.float32)
    faces = np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3)
    tb = tiny_scene(verts, faces, res=8, spp=1).tables(0)
    g = GpuScene(tb)
    m = 20000
    o = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    target = rng.uniform(-1, 1, (m, 3)).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    _, t_gpu, u_gpu, v_gpu = g.trace(o, d)
    _, t_ref, u_ref, v_ref = oracle.trace(tb, o, d)
    same = t_gpu == t_ref
    assert same.mean() > 0.998, (n_tris, same.mean())                   # rays through shared edges / coincident hits may differ
    hit = same & (t_ref >= 0)
    if n_tris >= 17:
        assert hit.sum() > 100
    if hit.any():
        assert np.abs(u_gpu[hit] - u_ref[hit]).max() < 2e-3 and np.abs(v_gpu[hit] - v_ref[hit]).max() < 2e-3
    assert ((t_gpu < 0) == (t_ref < 0))[same].all()


def test_tiny_scene_all_triangles_path_equals_the_tree_walk(monkeypatch):
    """<= 16 triangles: the geometry travels in the kernel arguments and closest_hit tests all of it (SceneView::tiny),
    two triangles of a wall as ONE parallelogram whose plane coordinates are mapped to the hit triangle's barycentrics
    (pack_tiny_prims); psdr_scene_set_option("tiny_scene", 0) walks the tree instead -- same triangles, barycentrics equal to rounding, same
    image, same gradients; a 17-triangle scene takes the tree"""
    from helpers import camera_rays
    sc, _ = load_scene("cbox", res=48, spp=8)
    tb = sc.tables(0)
    assert tb["num_tris"] == 12
    o_, d_ = camera_rays(tb, 100_000, seed=3)
    opts = _abi.make_opts(spp=8, integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    adj = np.random.default_rng(0).random((48 * 48, 3)).astype(np.float32)
    out = {}
    for mode in ("1", "0"):
        g = GpuScene(tb, options={"tiny_scene": int(mode)})
        out[mode] = (g.trace(o_, d_), g.render_c(opts), g.render_d_rev(opts, adj, want=["tri_info", "texels"], with_image=False)[1])
    (sa, ta, ua, va), (sb, tb_, ub, vb) = out["1"][0], out["0"][0]
    same = ta == tb_
    assert same.mean() > 0.9999                            # only rays through an edge shared by two triangles may differ
    print("tiny vs tree barycentrics: max |du| %.1e |dv| %.1e, triangles differ on %.1e of the rays" % (np.abs(ua[same] - ub[same]).max(), np.abs(va[same] - vb[same]).max(), 1 - same.mean()))
    assert np.abs(ua[same] - ub[same]).max() < 1e-5 and np.abs(va[same] - vb[same]).max() < 1e-5 and (sa[same] == sb[same]).all()
    assert rel_l2(out["1"][1], out["0"][1]) < 1e-5
    for k in ("tri_info", "texels"):
        assert rel_l2(out["1"][2][k], out["0"][2][k]) < 1e-4, k


def test_deep_paths_reverse_on_the_large_scene():
    """PathTracer depth 4 and 8 in reverse mode on the ~50 k-triangle interior: traversal stacks + 8-vertex path
    records + the gradient cache exceed the default 64 KB of dynamic LDS (hipFuncSetAttribute); depth 12: the path
    record moves to HBM (one column per thread of the grid); dot-product identity against forward mode"""
    from helpers import dot_tables, random_tangents
    from psdr_cuda.fixtures import make_interior_scene
    sc = make_interior_scene(seed=0, n_objects=10, res=48, spp=4); sc.configure()
    tb = sc.tables(0)
    g = GpuScene(tb)
    adj = np.random.default_rng(0).random((48 * 48, 3)).astype(np.float32)
    for depth in (4, 8, 12):
        o = _abi.make_opts(spp=4, integrator=_abi.INTEGRATOR_PATH, max_depth=depth)
        for names in (["tri_info"], ["texels"]):
            tan = random_tangents(tb, names, seed=3)
            _, d = g.render_d_fwd(o, [tan])
            _, grads = g.render_d_rev(o, adj, want=names, with_image=False)
            lhs, rhs = float((adj.astype(np.float64) * d[0]).sum()), dot_tables(grads, tan)
            assert abs(lhs - rhs) < 3e-3 * np.abs(adj * d[0]).sum(), (depth, names, lhs, rhs)
    # the two homes of the record give the same gradient: depth 8 (LDS) against depth 9 with an albedo so dark that a ninth vertex adds < 1e-6
    import os
    o8, o12 = (_abi.make_opts(spp=4, integrator=_abi.INTEGRATOR_PATH, max_depth=d) for d in (8, 12))
    for mode in (0, 1):                              # one kernel / value kernel + adjoint kernel: both read the HBM record
        g.set_option("rev_split", mode)
        _, g12 = g.render_d_rev(o12, adj, want=["texels"], with_image=False)
        tan = random_tangents(tb, ["texels"], seed=3)
        _, d12 = g.render_d_fwd(o12, [tan])
        lhs, rhs = float((adj.astype(np.float64) * d12[0]).sum()), dot_tables(g12, tan)
        assert abs(lhs - rhs) < 3e-3 * np.abs(adj * d12[0]).sum(), (mode, lhs, rhs)


@pytest.mark.parametrize("scene", ["cbox", "cbox_bunny"])
def test_trace_respects_tmax(scene):
    """psdr_trace: hits with t in [RayEpsilon, tmax] only -- a tmax short of the wall misses, a generous one hits,
    tmax = 0 / negative never hits; same answers as the oracle (all-triangles path and tree walk)"""
    from helpers import camera_rays
    sc, _ = load_scene(scene, res=16, spp=1)
    tb = sc.tables(0)
    g = GpuScene(tb)
    o, d = camera_rays(tb, 20_000, seed=5)
    _, tri, u, v = g.trace(o, d)
    info = tb["tri_info"].cpu().numpy()
    hit = tri >= 0
    assert hit.mean() > 0.9
    p = info[tri, 0:3] + u[:, None] * info[tri, 3:6] + v[:, None] * info[tri, 6:9]
    dist = np.where(hit, np.linalg.norm(p - o, axis=1), 1.0).astype(np.float32)
    rng = np.random.default_rng(1)
    for scale in (0.5, 0.999, 1.001, 2.0, 0.0, -1.0):
        tmax = (dist * scale).astype(np.float32)
        _, t_gpu, _, _ = g.trace(o, d, tmax)
        _, t_ref, _, _ = oracle.trace(tb, o, d, tmax)
        assert (t_gpu == t_ref).mean() > 0.999, scale
        if scale <= 0.999:
            assert (t_gpu[hit] != tri[hit]).all()            # nothing closer than the closest hit
        if scale >= 1.001:
            assert (t_gpu[hit] == tri[hit]).mean() > 0.999


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_render_fuzz_random_scenes(seed):
    """random emitter + random diffuse / rough-conductor triangle soups (intersecting, back-facing, slivers):
    finite images, GPU == oracle up to isolated samples, reverse mode == forward mode"""
    from helpers import random_tangents, dot_tables
    rng = np.random.default_rng(100 + seed)
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = 24
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 8, 0, 0, 0
    cam = psdr_cuda.PerspectiveCamera(45.0, 0.1, 1e3)
    cam.to_world = look_at([0, 0, 6], [0, 0, 0], [0, 1, 0])
    sc.add_sensor(cam)
    diffuse = psdr_cuda.Diffuse(rng.uniform(0.2, 0.9, 3)); diffuse.id = "d"
    metal = psdr_cuda.RoughConductor(float(rng.uniform(0.05, 0.5)), (0.2, 0.9, 1.1), (3.9, 2.4, 2.2)); metal.id = "m"
    black = psdr_cuda.Diffuse([0.0, 0.0, 0.0]); black.id = "k"
    for b in (diffuse, metal, black):
        sc.add_bsdf(b)

    def soup(n, spread, size):
        c = rng.uniform(-spread, spread, (n, 1, 3))
        t = c + size * rng.normal(size=(n, 3, 3))
        return t.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    for bsdf, n, emit in ((black, 6, [8.0, 6.0, 4.0]), (diffuse, int(rng.integers(20, 120)), None), (metal, int(rng.integers(20, 120)), None)):
        m = psdr_cuda.Mesh()
        m.use_face_normals = True
        m.enable_edges = False                         # a soup is not a manifold
        v, f = soup(n, 1.5, 0.6)
        m.set_geometry(v, f)
        sc.add_mesh(m, bsdf, emitter_radiance=emit)
    sc.finalize()
    sc.configure()
    tb = sc.tables(0)
    g = GpuScene(tb)
    adj = rng.random((24 * 24, 3)).astype(np.float32)
    for kw in (dict(bsdf_samples=1, light_samples=1), dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)):
        o = _abi.make_opts(spp=8, **kw)
        img, ref = g.render_c(o), oracle.render(tb, o)
        assert np.isfinite(img).all()
        bad = (np.abs(img - ref).max(1) > 2e-3 * (1 + np.abs(ref).max(1))).mean()
        assert bad < 0.03, (seed, kw, bad)
        for name in ("texels", "tri_info"):
            tan = random_tangents(tb, [name], seed=seed)
            _, dimg = g.render_d_fwd(o, [tan])
            assert np.isfinite(dimg[0]).all()
            _, grads = g.render_d_rev(o, adj, want=[name], with_image=False)
            lhs, rhs = float((adj.astype(np.float64) * dimg[0]).sum()), dot_tables(grads, tan)
            scale = float(np.abs(adj.astype(np.float64) * dimg[0]).sum())
            assert abs(lhs - rhs) <= 2e-2 * max(scale, 1e-6), (seed, kw, name, lhs, rhs, scale)


def test_capacity_edge_table_without_a_kept_edge_renders_zeros_not_nan():
    """A native configure() keeps the edge tables at the capacity of the candidate lists with the count on the device (csrc/psdr_tables.hip k_compact_*);
    when NO edge is kept the table is all zero rows with pmf 0 / cmf 1.  The primary-edge kernels treat pmf 0 as an invalid draw and a zero
    secondary-edge row fails its own validity test: the derivative image is the interior term alone, forward and reverse, and finite."""
    sc, P = load_scene("cbox_bunny", res=32, spp=2, sppe=4, sppse=4, translate=(1, (1.0, 0.5, 0.0)))
    from helpers import tangents_wrt
    tb = dict(sc.tables(0, capacity=True))
    tan = tangents_wrt(tb, P)
    o_all = _abi.make_opts(spp=2, sppe=4, sppse=4, bsdf_samples=1, light_samples=1)
    o_int = _abi.make_opts(spp=2, sppe=0, sppse=0, bsdf_samples=1, light_samples=1)
    for k in ("prim_edge", "prim_pmf", "sec_edge", "sec_pmf"):
        tb[k] = torch.zeros_like(tb[k].detach())
    tb["prim_cmf"], tb["sec_cmf"] = torch.ones_like(tb["prim_cmf"]), torch.ones_like(tb["sec_cmf"])
    g = GpuScene(tb)
    _, d_all = g.render_d_fwd(o_all, [tan])
    _, d_int = g.render_d_fwd(o_int, [tan])
    assert np.isfinite(d_all[0]).all() and np.abs(d_int[0]).max() > 0
    assert np.array_equal(d_all[0], d_int[0])
    adj = np.random.default_rng(0).random((32 * 32, 3)).astype(np.float32)
    _, ga = g.render_d_rev(o_all, adj, want=["tri_info", "sec_edge", "prim_edge"], with_image=False)
    _, gi = g.render_d_rev(o_int, adj, want=["tri_info"], with_image=False)
    assert all(np.isfinite(v).all() for v in ga.values())
    assert np.abs(ga["sec_edge"]).max() == 0 and np.abs(ga["prim_edge"]).max() == 0 and rel_l2(ga["tri_info"], gi["tri_info"]) < 1e-6


def test_capacity_ONE_edge_tables_whose_only_candidate_was_dropped():
    """with ONE candidate edge and none kept, DiscreteDistribution::sample_reuse takes its size == 1 shortcut and "draws" the zero row with
    pmf 1 -- the primary-edge kernels must not turn its zero length into an infinite pdf (primary_edge_sample: length > 0), the zero secondary-edge row
    fails its own validity test.  Forward and reverse: the interior term alone, finite."""
    sc, P = load_scene("cbox_bunny", res=32, spp=2, sppe=4, sppse=4, translate=(1, (1.0, 0.5, 0.0)))
    from helpers import tangents_wrt
    tb = dict(sc.tables(0, capacity=True))
    tan = dict(tangents_wrt(tb, P))
    tb["prim_edge"] = torch.zeros_like(tb["prim_edge"][:1].detach()); tb["prim_pmf"] = torch.zeros(1); tb["prim_cmf"] = torch.ones(1); tb["num_prim_edges"] = 1
    tb["sec_edge"] = torch.zeros_like(tb["sec_edge"][:1].detach()); tb["sec_pmf"] = torch.zeros(1); tb["sec_cmf"] = torch.ones(1); tb["num_sec_edges"] = 1
    if tb.get("prim_edge_z") is not None:
        tb["prim_edge_z"] = tb["prim_edge_z"][:1]
    if tb.get("sec_edge_faces") is not None:
        tb["sec_edge_faces"] = tb["sec_edge_faces"][:1]
    for k in ("prim_edge", "sec_edge"):
        if tan.get(k) is not None:
            tan[k] = tan[k][:1]
    o_all = _abi.make_opts(spp=2, sppe=4, sppse=4, bsdf_samples=1, light_samples=1)
    o_int = _abi.make_opts(spp=2, sppe=0, sppse=0, bsdf_samples=1, light_samples=1)
    g = GpuScene(tb)
    _, d_all = g.render_d_fwd(o_all, [tan])
    _, d_int = g.render_d_fwd(o_int, [tan])
    assert np.isfinite(d_all[0]).all() and np.abs(d_int[0]).max() > 0
    assert np.array_equal(d_all[0], d_int[0])
    adj = np.random.default_rng(0).random((32 * 32, 3)).astype(np.float32)
    _, ga = g.render_d_rev(o_all, adj, want=["tri_info", "sec_edge", "prim_edge"], with_image=False)
    assert all(np.isfinite(v).all() for v in ga.values())
    assert np.abs(ga["sec_edge"]).max() == 0 and np.abs(ga["prim_edge"]).max() == 0
