"""The tree built ON THE DEVICE (csrc/psdr_lbvh.h: Morton codes, rocPRIM sort, Karras radix tree, bottom-up fit) -- what the
reference's OptiX GAS build does at every configure() (include/psdr/scene/optix.h:277-340, scene.cpp:247-248).  The closest
hit does not depend on the tree: a device-built tree must return the host-built tree's hits, the oracle's images, and survive a
refit.  psdr_scene_set_option("bvh_build", 1) forces it (by default only tables of >= 2^18 triangles take it)."""
import os
import time

import numpy as np
import pytest
import torch

import oracle
from helpers import GpuScene, camera_rays, load_scene, rel_l2
from psdr_cuda import _abi

pytestmark = pytest.mark.gpu


class forced_build:
    """Handles created inside get psdr_scene_set_option("bvh_build", 1 = device / 0 = host): tests/helpers.py GpuScene reads PSDR_OPTIONS
    (the library itself reads no environment variable)."""
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("PSDR_OPTIONS")
        os.environ["PSDR_OPTIONS"] = "bvh_build=%d" % (1 if self.mode == "device" else 0)

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["PSDR_OPTIONS"]
        else:
            os.environ["PSDR_OPTIONS"] = self.old


def bvh_stats(g):
    import ctypes as C
    out = (C.c_int32 * 4)()
    _abi.check(g.lib, g.lib.psdr_bvh_stats(g.h, out))
    return dict(builds=out[0], refits=out[1], nodes=out[2], depth=out[3])


def bounce_rays(tb, tri, u, v, seed):
    info = tb["tri_info"].cpu().numpy()
    idx = np.nonzero(tri >= 0)[0][:100_000]
    p = info[tri[idx], 0:3] + u[idx, None] * info[tri[idx], 3:6] + v[idx, None] * info[tri[idx], 6:9]
    d = np.random.default_rng(seed).normal(size=p.shape).astype(np.float32)
    return p.astype(np.float32), d / np.linalg.norm(d, axis=1, keepdims=True)


@pytest.mark.parametrize("scene", ["cbox_bunny", "bunny_light"])
def test_device_tree_returns_the_host_trees_hits(scene):
    sc, _ = load_scene(scene, res=64)
    tb = sc.tables(0)
    with forced_build("host"):
        gh = GpuScene(tb)
    with forced_build("device"):
        gd = GpuScene(tb)
    sh, sd = bvh_stats(gh), bvh_stats(gd)
    T = tb["tri_info"].shape[0]
    assert sd["nodes"] == T - 1 and 0 < sd["depth"] <= 38, sd                      # the radix tree over T triangles (node array; subtrees of <= 4 are leaves)
    print("%s: host tree %s, device tree %s" % (scene, sh, sd))
    o, d = camera_rays(tb, 200_000, seed=3)
    for rays in ((o, d), None):
        if rays is None:
            rays = bounce_rays(tb, a[1], a[2], a[3], 4)
        a, b = gh.trace(*rays), gd.trace(*rays)
        same = a[1] == b[1]
        assert same.mean() > 0.9995, same.mean()                               # equal-distance hits on shared edges may resolve differently
        assert np.array_equal(a[0][same], b[0][same])
        hit = same & (a[1] >= 0)
        # (the host tree of these scenes is the two-level one: walls are tested as parallelograms, a few ulp apart)
        assert hit.sum() > 20_000 and np.abs(a[2][hit] - b[2][hit]).max() < 1e-5 and np.abs(a[3][hit] - b[3][hit]).max() < 1e-5
    # rendered through the device tree: the oracle's image / derivative image
    opt = _abi.make_opts(spp=8, bsdf_samples=1, light_samples=1)
    sc2, _ = load_scene(scene, res=48, spp=8)
    tb2 = sc2.tables(0)
    with forced_build("device"):
        g2 = GpuScene(tb2)
    img, ref = g2.render_c(opt), oracle.render(tb2, opt)
    assert rel_l2(img, ref) < 1e-3, rel_l2(img, ref)


def test_device_tree_refit_follows_the_vertices():
    sc, _ = load_scene("cbox_bunny", res=64)
    tb = sc.tables(0)
    with forced_build("device"):
        g = GpuScene(tb)
    assert bvh_stats(g)["builds"] == 1
    # move the bunny (Mesh[1]) by writing new rows into the table the handle points at, then psdr_bvh_build again: a refit
    f0, f1 = int(tb["face_offset"][1]), int(tb["face_offset"][2])
    rows = g.tb["tri_info"]
    rows[f0:f1, 0] += 7.5
    rows[f0:f1, 2] -= 3.0
    torch.cuda.synchronize()
    _abi.check(g.lib, g.lib.psdr_bvh_build(g.h, None))
    st = bvh_stats(g)
    assert st["builds"] == 1 and st["refits"] == 1, st
    tb_moved = dict(tb); tb_moved["tri_info"] = rows.detach().cpu()
    with forced_build("host"):
        fresh = GpuScene(tb_moved)
    o, d = camera_rays(tb_moved, 200_000, seed=5)
    a, b = fresh.trace(o, d), g.trace(o, d)
    same = a[1] == b[1]
    assert same.mean() > 0.9995 and (a[1] >= f0).sum() > 1000
    hit = same & (a[1] >= 0)
    assert np.abs(a[2][hit] - b[2][hit]).max() < 1e-5


def test_large_table_is_built_on_the_device_by_default(tmp_path):
    """A 2 x 363 x 363 = 263 538-triangle height field: above the 2^18 threshold the library builds on the device by itself."""
    import psdr_cuda
    from psdr_cuda.scene import look_at
    n = 364
    xs = np.linspace(-200.0, 200.0, n)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = 30.0 * np.sin(X * 0.05) * np.cos(Z * 0.04) + 5.0 * np.sin(X * 0.31 + Z * 0.17)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    faces = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)])
    path = tmp_path / "field.obj"
    with open(path, "w") as f:
        f.write("".join("v %.6f %.6f %.6f\n" % t for t in zip(X.ravel(), Y.ravel(), Z.ravel())))
        f.write("".join("f %d %d %d\n" % tuple(r + 1) for r in faces))
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = 64
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 4, 0, 0, 0
    cam = psdr_cuda.PerspectiveCamera(40.0, 0.1, 1e4)
    cam.to_world = look_at([0, 260, 420], [0, 0, 0], [0, 1, 0])
    sc.add_sensor(cam)
    white = psdr_cuda.Diffuse([0.8, 0.8, 0.8]); white.id = "white"; sc.add_bsdf(white)
    black = psdr_cuda.Diffuse([0.0, 0.0, 0.0]); black.id = "black"; sc.add_bsdf(black)
    from psdr_cuda.fixtures import DATA_DIR
    light = psdr_cuda.Mesh(); light.load(os.path.join(DATA_DIR, "objects", "cbox", "emitter.obj"))
    xf = np.eye(4); xf[:3, 3] = [0, 120, 0]
    light._to_world_raw = torch.as_tensor(xf, dtype=torch.float32, device=light._to_world_raw.device)
    sc.add_mesh(light, black, [30.0, 30.0, 30.0])
    field = psdr_cuda.Mesh(); field.load(str(path)); field.enable_edges = False
    sc.add_mesh(field, white, None)
    sc.finalize(); sc.configure()
    tb = sc.tables(0)
    T = tb["tri_info"].shape[0]
    assert T >= 1 << 18
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g = GpuScene(tb); torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    st = bvh_stats(g)
    assert st["nodes"] == T - 1, st                                              # the radix tree, not the SAH tree
    with forced_build("host"):
        t0 = time.perf_counter(); gh = GpuScene(tb); torch.cuda.synchronize(); t_host = time.perf_counter() - t0
    print("T = %d: handle + build on the device %.1f ms (depth %d), on the host %.1f ms (depth %d)" % (T, t_dev * 1e3, st["depth"], t_host * 1e3, bvh_stats(gh)["depth"]))
    o, d = camera_rays(tb, 200_000, seed=7)
    a, b = gh.trace(o, d), g.trace(o, d)
    same = a[1] == b[1]
    assert same.mean() > 0.999 and (a[1] >= 0).mean() > 0.3
    hit = same & (a[1] >= 0)
    assert np.abs(a[2][hit] - b[2][hit]).max() < 1e-5
    so, stri, su, sv = oracle.trace(tb, o[:20000], d[:20000])
    assert (stri == b[1][:20000]).mean() > 0.999
    opt = _abi.make_opts(spp=4, bsdf_samples=1, light_samples=1)
    assert rel_l2(g.render_c(opt), gh.render_c(opt)) < 1e-4


def scene_info(g):
    return _abi.scene_stats(g.h)


def test_scratch_growth_of_a_device_build_keeps_the_reverse_mode_buffers():
    """lbvh_build's scratch-growth branch used to hipFree the reverse-mode / edge-list buffers of the handle without
    forgetting them.  Reverse mode (all three terms: split secondary-edge list, primary-edge replicas, value-sweep record), then a
    device build over a LARGER table on the same handle, then reverse mode again -- the gradients must be those of a fresh handle."""
    from psdr_cuda.fixtures import make_interior_scene
    sc, _ = load_scene("cbox_bunny", res=48, spp=4, sppe=4, sppse=4)
    tb = sc.tables(0)
    opt = _abi.make_opts(spp=4, sppe=4, sppse=4)
    optp = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=4)
    adj = np.random.default_rng(1).random((48 * 48, 3)).astype(np.float32)
    with forced_build("device"):
        g = GpuScene(tb)
        g.render_d_rev(opt, adj, with_image=False)
        g.render_d_rev(optp, adj, want=["tri_info", "texels"], with_image=False)
        big = make_interior_scene(seed=1, n_objects=6, res=48, spp=4)          # ~30 k triangles: the LBVH scratch has to grow
        big.opts.sppe = big.opts.sppse = 4
        big.configure()
        tb2 = big.tables(0)
        assert tb2["tri_info"].shape[0] > 4 * tb["tri_info"].shape[0]
        g.tb = {k: (v.detach().cuda() if isinstance(v, torch.Tensor) else v) for k, v in tb2.items()}
        g.set_guide(None)
        _abi.check(g.lib, g.lib.psdr_bvh_build(g.h, None))
        assert bvh_stats(g)["builds"] == 2
        fresh = GpuScene(tb2)
        for o, want in ((opt, None), (optp, ["tri_info", "texels"])):
            kw = {} if want is None else {"want": want}
            _, a = g.render_d_rev(o, adj, with_image=False, **kw)
            _, b = fresh.render_d_rev(o, adj, with_image=False, **kw)
            for k in b:
                assert np.isfinite(a[k]).all() and rel_l2(a[k], b[k]) < 1e-4, (k, rel_l2(a[k], b[k]))      # same tree, same samples: atomics order only
        g.close(); fresh.close()                                                  # psdr_scene_destroy frees the four buffers (once)


def _room_with_field(tmp_path, cells):
    """A height field of 2 * cells^2 triangles listed BEFORE the Cornell-box walls: the walls' global triangle ids start behind it."""
    import psdr_cuda
    from psdr_cuda.fixtures import DATA_DIR
    n = cells + 1
    xs = np.linspace(-80.0, 80.0, n)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = 40.0 + 12.0 * np.sin(X * 0.08) * np.cos(Z * 0.07)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    faces = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)])
    path = tmp_path / ("field%d.obj" % cells)
    with open(path, "w") as f:
        f.write("".join("v %.6f %.6f %.6f\n" % t for t in zip(X.ravel(), Y.ravel(), Z.ravel())))
        f.write("".join("f %d %d %d\n" % tuple(r + 1) for r in faces))
    base = psdr_cuda.Scene()
    from psdr_cuda.fixtures import scene_path
    base.load_file(scene_path("cbox"), False)
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = 48
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 4, 0, 0, 0
    sc.add_sensor(base.m_sensors[0])
    white = psdr_cuda.Diffuse([0.8, 0.8, 0.8]); white.id = "white"; sc.add_bsdf(white)
    black = psdr_cuda.Diffuse([0.0, 0.0, 0.0]); black.id = "black"; sc.add_bsdf(black)
    field = psdr_cuda.Mesh(); field.load(str(path)); field.enable_edges = False
    sc.add_mesh(field, white, None)
    light = psdr_cuda.Mesh(); light.load(os.path.join(DATA_DIR, "objects", "cbox", "emitter.obj"))
    xf = np.eye(4); xf[:3, 3] = [50, 190, 0]
    light._to_world_raw = torch.as_tensor(xf, dtype=torch.float32, device=light._to_world_raw.device)
    light.use_face_normals = True
    sc.add_mesh(light, black, [20.0, 20.0, 8.0])
    for name in ("floor", "ceil", "wall_back", "wall_left", "wall_right"):
        m = psdr_cuda.Mesh(); m.load(os.path.join(DATA_DIR, "objects", "cbox", name + ".obj")); m.use_face_normals = True
        sc.add_mesh(m, white, None)
    sc.finalize(); sc.configure()
    return sc.tables(0)


def test_inline_primitives_behind_65535_triangles_fall_back_to_one_tree(tmp_path):
    """an inline primitive packs two GLOBAL triangle ids into 16 bits each.  A room whose large mesh is listed first (as in
    cbox_bunny.xml) with >= 65535 triangles ahead of the walls must not take the two-level tree; a smaller one does, and both return the
    oracle's hits and images."""
    for cells, two_level in ((100, True), (182, False)):
        tb = _room_with_field(tmp_path, cells)
        T = tb["tri_info"].shape[0]
        assert (T > 0xffff + 12) == (not two_level)
        g = GpuScene(tb)
        info = scene_info(g)
        assert (info["n_blas"] > 0) == two_level, info
        o, d = camera_rays(tb, 100_000, seed=11)
        shape, tri, u, v = g.trace(o, d)
        _, otri, ou, ov = oracle.trace(tb, o[:30000], d[:30000])
        same = otri == tri[:30000]
        assert same.mean() > 0.999, same.mean()
        walls = tri >= T - 12
        assert walls.sum() > 5000                                                  # the walls (ids behind the field) are hit and reported with THEIR ids
        tm = tb["tri_mesh"].cpu().numpy() & ~0x40000000
        assert np.array_equal(shape[tri >= 0], tm[tri[tri >= 0]])
        rays = bounce_rays(tb, tri, u, v, 12)
        _, tri2, _, _ = g.trace(*rays)
        _, otri2, _, _ = oracle.trace(tb, rays[0][:20000], rays[1][:20000])
        assert (otri2 == tri2[:20000]).mean() > 0.999
        opt = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=4)
        img, ref = g.render_c(opt), oracle.render(tb, opt)
        # a wavy height field under grazing light: a handful of the 9 216 samples resolve an epsilon-sized tie differently in the two fp32
        # evaluations (the hit comparisons above are the point of this test) -- all but a few pixels agree to 1e-5
        bad = (np.abs(img - ref).max(axis=1) > 1e-5 * (1.0 + np.abs(ref).max(axis=1)))
        assert bad.mean() < 5e-3 and rel_l2(img, ref) < 3e-3, (bad.mean(), rel_l2(img, ref))
        g.close()


def test_two_level_tree_stands_only_while_the_mesh_level_tables_fit_its_kernels():
    """The two-level kernel instances read the mesh-level tables from LDS and have no other path (psdr_device.h Tab<FL>::lds_small): a scene whose
    tables outgrow that block is built as ONE tree, and a handle whose tables grow under a standing two-level tree (psdr_scene_set_tables
    without a rebuild) changes its tree before the next launch.  Same image either way."""
    sc, _ = load_scene("cbox_bunny", res=48, spp=4)
    tb = dict(sc.tables(0))
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=4)
    g = GpuScene(tb)
    assert _abi.scene_stats(g.h)["n_blas"] > 0
    ref = g.render_c(o)
    # 40 BSDF records (the scene's own, then copies nobody points at): more than the LDS block of the two-level kernels takes
    big = dict(tb)
    big["bsdf_rec"] = torch.cat([tb["bsdf_rec"], tb["bsdf_rec"][:1].repeat(40 - tb["bsdf_rec"].shape[0], 1)]).contiguous()
    big["num_bsdfs"] = 40
    g2 = GpuScene(big)
    assert _abi.scene_stats(g2.h)["n_blas"] == 0
    assert rel_l2(g2.render_c(o), ref) < 1e-5
    # the same growth on the standing handle: set_tables only, the launch brings the tree in line
    g.tb = {k: (v.detach().cuda() if isinstance(v, torch.Tensor) else v) for k, v in big.items()}
    g.set_guide(None)
    img = g.render_c(o)
    assert _abi.scene_stats(g.h)["n_blas"] == 0 and rel_l2(img, ref) < 1e-5
    # and back to tables that fit (the refit of an unchanged triangle table keeps the tree it finds: one tree serves them as well)
    g.tb = {k: (v.detach().cuda() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
    g.set_guide(None)
    _abi.check(g.lib, g.lib.psdr_bvh_build(g.h, None))
    assert rel_l2(g.render_c(o), ref) < 1e-5


def test_refit_is_not_taken_when_the_emitter_layout_changed():
    """psdr_bvh_build refits while the triangle count stands -- but the host copy of emitter_i (LDS table sizes of the two-level kernels, hot
    gradient rows) is refreshed by the build paths only: a changed emitter layout under an unchanged triangle count must take the
    full build, and the handle must then render what a fresh handle on the same tables renders."""
    sc, _ = load_scene("cbox_bunny", res=48, spp=8)
    tb = sc.tables(0)
    g = GpuScene(tb)
    opt = _abi.make_opts(spp=8, bsdf_samples=1, light_samples=1)
    g.render_c(opt)
    b0 = bvh_stats(g)
    # unchanged tables: a refit
    _abi.check(g.lib, g.lib.psdr_bvh_build(g.h, None))
    b1 = bvh_stats(g)
    assert b1["builds"] == b0["builds"] and b1["refits"] == b0["refits"] + 1, (b0, b1)
    # the area light keeps only its first triangle: same triangle table, another emitter_i row (written into the table the handle points at)
    ei = next(t for t in g.keep if t.data_ptr() == int(g.desc.emitter_i))        # the int32 table the handle points at
    ei = ei.view(-1, _abi.EMITTER_I_STRIDE)
    assert int(ei[0, 2]) == 2
    ei[0, 2] = 1
    torch.cuda.synchronize()
    _abi.check(g.lib, g.lib.psdr_bvh_build(g.h, None))
    b2 = bvh_stats(g)
    assert b2["builds"] == b1["builds"] + 1 and b2["refits"] == b1["refits"], (b1, b2)
    tb_mod = dict(tb); tb_mod["emitter_i"] = ei.detach().cpu().clone()
    fresh = GpuScene(tb_mod)
    a, b = g.render_c(opt), fresh.render_c(opt)
    assert np.isfinite(a).all() and a.mean() > 0 and rel_l2(a, b) < 1e-6, rel_l2(a, b)
