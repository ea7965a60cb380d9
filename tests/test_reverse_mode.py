"""Reverse mode (psdr_render_d_rev: hand-written adjoints + gradient scatter-add) against forward mode
(dual numbers) by the dot-product identity  <a, J t> == <J^T a, t>  for random tangent tables t and a
random adjoint image a.  Forward mode itself is pinned against the oracle elsewhere.  Runs the
product code on the host (hostcheck) here and through the C ABI on the GPU (-m gpu)."""
import numpy as np
import pytest

from helpers import (GpuScene, dot_tables, host_render, host_render_rev, load_scene, random_tangents, rel_l2, same_rays)
from psdr_cuda import _abi

CASES = [
    ("cbox", dict(bsdf_samples=1, light_samples=1), ["texels", "emitter_rad", "tri_info", "cam_to_world"], 0, 0),
    ("cbox_rough", dict(bsdf_samples=1, light_samples=1), ["texels", "tri_info", "cam_to_world"], 0, 0),
    ("cbox_rough", dict(bsdf_samples=2, light_samples=0), ["texels", "tri_info"], 0, 0),
    ("cbox_rough", dict(bsdf_samples=0, light_samples=2), ["texels", "tri_info"], 0, 0),
    ("cbox_rough", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3), ["texels", "emitter_rad", "tri_info", "cam_to_world"], 0, 0),
    ("cbox_bunny", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=4), ["texels", "tri_info"], 0, 0),
    ("cbox_occluder", dict(bsdf_samples=1, light_samples=1), ["tri_info", "sec_edge", "prim_edge", "cam_to_world"], 4, 4),
    ("cbox", dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["position"]), ["tri_info", "cam_to_world"], 0, 0),
    ("cbox_bunny", dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["shNormal"]), ["tri_info"], 0, 0),
    # environment map: lat-long texels, scale / rotation record, directions through the geometry and the camera
    ("bunny_env", dict(bsdf_samples=1, light_samples=1), ["texels", "env_f", "tri_info", "cam_to_world"], 0, 0),
    ("bunny_env", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3), ["texels", "env_f", "tri_info"], 0, 0),
    ("cbox_env", dict(bsdf_samples=2, light_samples=2), ["texels", "env_f", "emitter_rad", "tri_info"], 0, 0),
    ("cbox_env", dict(bsdf_samples=1, light_samples=1), ["tri_info", "sec_edge", "prim_edge"], 4, 4),
]


def _tangents(tb, n):
    """random tangent table; rows of the env-map bounding mesh stay zero: its vertices are constants in the
    reference (scene.cpp:143-172) and the hand-written adjoints use identities (orthonormal frames) that
    hold for tangents produced by the table chain, not for arbitrary ones on those rows"""
    tan = random_tangents(tb, [n], seed=1)
    if n == "tri_info" and tb.get("env_emitter", -1) >= 0:
        mesh = (tb["tri_mesh"] & 0x3fffffff).long()
        tan[n][tb["mesh_bsdf"][mesh] < 0] = 0.0
    return tan


def _setup(scene, kw, sppe, sppse, res=16, spp=4):
    sc, _ = load_scene(scene, res=res, spp=spp, sppe=sppe, sppse=sppse)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=spp, sppe=sppe, sppse=sppse, rng_offset=(2, 3, 4), **kw)
    adj = np.random.default_rng(5).random((res * res, 3)).astype(np.float32)
    return tb, o, adj


@pytest.mark.parametrize("scene,kw,names,sppe,sppse", CASES)
def test_dot_product_identity_host(scene, kw, names, sppe, sppse):
    tb, o, adj = _setup(scene, kw, sppe, sppse)
    for n in names:
        tan = _tangents(tb, n)
        img_f, dimg = host_render(tb, o, mode=1, tangents=tan)
        img_r, grads = host_render_rev(tb, o, adj, want=[n])
        lhs, rhs = float((adj.astype(np.float64) * dimg).sum()), dot_tables(grads, tan)
        scale = float(np.abs(adj.astype(np.float64) * dimg).sum())       # the sum itself may cancel
        assert abs(lhs - rhs) <= 1e-4 * max(scale, 1e-6), (n, lhs, rhs, scale)
        assert rel_l2(img_r, img_f) < 1e-4          # the reverse pass also delivers the primal image


def test_reverse_gradient_equals_forward_columns():
    """d loss / d albedo(r,g,b): reverse texel gradient == three forward-mode derivative images."""
    import torch
    tb, o, adj = _setup("cbox", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3), 0, 0)
    _, grads = host_render_rev(tb, o, adj, want=["texels"])
    for c in range(3):
        t = torch.zeros_like(tb["texels"]); t[c] = 1.0
        _, dimg = host_render(tb, o, mode=1, tangents={"texels": t})
        ref = float((adj.astype(np.float64) * dimg).sum())
        assert abs(grads["texels"][c] - ref) < 1e-4 * abs(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,kw,names,sppe,sppse", CASES)
def test_dot_product_identity_gpu(scene, kw, names, sppe, sppse):
    tb, o, adj = _setup(scene, kw, sppe, sppse, res=32, spp=8)
    g = GpuScene(tb)
    img_r, grads = g.render_d_rev(o, adj, want=names)
    for n in names:
        tan = _tangents(tb, n)
        img_f, dimg = g.render_d_fwd(o, [tan])
        lhs, rhs = float((adj.astype(np.float64) * dimg[0]).sum()), dot_tables(grads, tan)
        scale = float(np.abs(adj.astype(np.float64) * dimg[0]).sum())
        # bunny: isolated edge-on triangles have fp32-ill-conditioned derivatives (DESIGN.md 'numerical fragility')
        tol = 5e-3 if "bunny" in scene else 1e-3
        assert abs(lhs - rhs) <= tol * max(scale, 1e-6), (n, lhs, rhs, scale)
    # the two kernels are different instruction streams: a 1-ulp difference can flip the branch of an isolated
    # sample (a shadow test at its epsilon, an environment-map cell border) -- a handful of pixels at most
    if "bunny" in scene:
        assert rel_l2(img_r, img_f) < 2e-2
    else:
        bad = np.abs(img_r - img_f).max(1) > 1e-4 * (1.0 + np.abs(img_f).max(1))
        assert bad.mean() < 0.01 and rel_l2(img_r[~bad], img_f[~bad]) < 1e-4 and rel_l2(img_r, img_f) < 1e-3, (int(bad.sum()), rel_l2(img_r, img_f))


@pytest.mark.gpu
def test_gpu_reverse_matches_host_reverse():
    tb, o, adj = _setup("cbox_occluder", dict(bsdf_samples=1, light_samples=1), 4, 4)
    _, gh = host_render_rev(tb, o, adj)
    _, gg = GpuScene(tb).render_d_rev(o, adj)
    for n in gh:
        assert rel_l2(gg[n], gh[n]) < 1e-3, n


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["direct11", "path3"])
def test_split_reverse_launch_equals_the_fused_kernel(kind):
    """Tree scenes run reverse mode as two kernels -- the value sweep at the occupancy of a forward kernel, leaving a record per
    path (primary triangle, vertex count, suffix radiances, the triangles the rays arrived at), then the adjoint sweep from the
    record, without traversal (csrc/psdr_kernels.h render_rev).  Same samples, same arithmetic: the gradients of the one-kernel
    launch up to the order of the float adds.  psdr_scene_set_option("rev_split", 1 / 0) forces either (default: by scene and launch size)."""
    import os
    import numpy as np
    from helpers import GpuScene, load_scene, rel_l2
    from psdr_cuda import _abi
    kw = dict(bsdf_samples=1, light_samples=1) if kind == "direct11" else dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    sc, _ = load_scene("cbox_bunny", res=96, spp=8)
    tb = sc.tables(0)
    g = GpuScene(tb)
    o = _abi.make_opts(spp=8, **kw)
    adj = np.random.default_rng(3).random((96 * 96, 3)).astype(np.float32)
    out = {}
    # "0": one kernel; "1": value kernel + adjoint kernel; "w": the value sweep as the traced wavefront (two-level scenes, PathTracer: the default split)
    for mode in ("0", "1", "w"):
        g.set_option("rev_split", 0 if mode == "0" else 1)
        g.set_option("wf_traced", 1 if mode == "w" else 0)
        img, grads = g.render_d_rev(o, adj, want=["tri_info", "texels", "emitter_rad", "cam_to_world"])
        out[mode] = (img, grads, g.counters()[0])
    assert same_rays(out["0"][2], out["1"][2])                                       # the same rays traced
    assert rel_l2(out["1"][0], out["0"][0]) < 1e-6
    for k in ("tri_info", "texels", "emitter_rad", "cam_to_world"):
        a, b = out["0"][1][k], out["1"][1][k]
        assert np.abs(a).max() > 0 and rel_l2(b, a) < 2e-5, (k, rel_l2(b, a))
    # the wavefront value sweep: separately compiled fp32 kernels -- the same estimator on the same random numbers up to isolated samples
    # (an ulp in a bounce direction resolves a tie at a triangle edge the other way: tests/test_gpu_parity.py says the same of renderC)
    assert abs(out["w"][2] - out["0"][2]) <= 1e-4 * out["0"][2], (out["w"][2], out["0"][2])
    print("%s: wavefront value sweep vs one kernel: rays %d / %d, image rel-L2 %.2e, gradients %s" % (kind, out["w"][2], out["0"][2], rel_l2(out["w"][0], out["0"][0]),
          {k: "%.1e" % rel_l2(out["w"][1][k], out["0"][1][k]) for k in out["0"][1]}))
    assert rel_l2(out["w"][0], out["0"][0]) < 3e-4                         # measured 4.7e-5 (three of 457 855 rays differ)
    for k in ("tri_info", "texels", "emitter_rad", "cam_to_world"):
        assert rel_l2(out["w"][1][k], out["0"][1][k]) < 1e-3, (k, rel_l2(out["w"][1][k], out["0"][1][k]))          # measured 1e-6 .. 1.1e-4


@pytest.mark.gpu
def test_secondary_edge_split_launch_equals_one_kernel():
    """The secondary-edge term runs as a filter kernel (the two rays every slot traces; survivors compacted) plus the full kernel over
    the survivors, forward and reverse (csrc/psdr_kernels.h k_secondary_edge_filter).  Same samples, same arithmetic: the derivative
    image and the gradients of the one-kernel launch up to the order of the float adds, and the same number of rays."""
    import os
    from helpers import tangents_wrt
    sc, P = load_scene("cbox_occluder", res=64, spp=0, sppe=0, sppse=32, translate=(1, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    g = GpuScene(tb)
    o = _abi.make_opts(spp=0, sppe=0, sppse=32, bsdf_samples=1, light_samples=1)
    tan = tangents_wrt(tb, P)
    adj = np.random.default_rng(4).random((64 * 64, 3)).astype(np.float32)
    out = {}
    for mode in ("0", "1"):
        g.set_option("sedge_split", int(mode))
        _, d = g.render_d_fwd(o, [tan]); rays_f = g.counters()[0]
        _, grads = g.render_d_rev(o, adj, want=["tri_info", "sec_edge", "cam_to_world"], with_image=False); rays_r = g.counters()[0]
        out[mode] = (d[0], grads, rays_f, rays_r)
    assert same_rays(out["0"][2], out["1"][2]) and same_rays(out["0"][3], out["1"][3])
    assert np.abs(out["0"][0]).max() > 0 and rel_l2(out["1"][0], out["0"][0]) < 1e-5
    for k in ("tri_info", "sec_edge", "cam_to_world"):
        a, b = out["0"][1][k], out["1"][1][k]
        assert np.abs(a).max() > 0 and rel_l2(b, a) < 2e-5, (k, rel_l2(b, a))


@pytest.mark.gpu
@pytest.mark.parametrize("scene,kw", [("cbox", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)), ("cbox", dict(bsdf_samples=1, light_samples=1)),
                                      ("cbox_rough", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)), ("cbox_bunny", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3))])
def test_partial_tail_wave_gradients_equal_the_host_run(scene, kw):
    """19 x 19 pixels x 3 samples = 1 083 slots: the last workgroup is partly filled and its last wave holds 59 slots.  The wave totals of the sinks
    (DPP row scans + v_readlane of lane 63, the segmented run sums, the register accumulators of the emitter rows) need every lane of a wave in the
    call; the kernels round their loops up to whole workgroups for that.  The scalar host run of the same code has no waves: every gradient table
    must agree (a launch with a partial tail wave for every instance that sums through DPP)."""
    res, spp = 19, 3
    sc, _ = load_scene(scene, res=res, spp=spp)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=spp, rng_offset=(2, 3, 4), **kw)
    adj = np.random.default_rng(7).random((res * res, 3)).astype(np.float32)
    names = ["texels", "emitter_rad", "tri_info", "cam_to_world"]
    img_h, gh = host_render_rev(tb, o, adj, want=names)
    img_g, gg = GpuScene(tb).render_d_rev(o, adj, want=names)
    assert rel_l2(img_g, img_h) < (2e-2 if "bunny" in scene else 1e-4)
    for n in names:
        assert np.abs(gh[n]).max() > 0, n
        # bunny: isolated edge-on triangles (see test_dot_product_identity_gpu)
        assert rel_l2(gg[n], gh[n]) < (2e-2 if "bunny" in scene else 1e-3), (n, rel_l2(gg[n], gh[n]))


@pytest.mark.gpu
def test_render_c_keeps_the_value_sweep_records_for_the_reverse_call():
    """renderD + enoki.backward renders the primal image first (the loss needs it) and differentiates it later; on a two-level scene the reverse launch of a
    PathTracer would then repeat that render as its value sweep (the traced wavefront with recording stages).  psdr_render_c(PSDR_FLAG_KEEP_RECORDS) runs
    the recording stages right away and the psdr_render_d_rev of the same samples on the same tables runs its adjoint kernel only: same image, same
    gradients, NO ray traced by the reverse call.  Anything that changes the samples or the tables in between falls back to the full launch."""
    from helpers import GpuScene, load_scene, rel_l2
    res, spp = 256, 16                                          # 2^20 slots: where the library splits the reverse launch by itself
    sc, _ = load_scene("cbox_bunny", res=res, spp=spp)
    tb = sc.tables(0)
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp, rng_offset=(7, 0, 0))
    adj = np.random.default_rng(5).random((res * res, 3)).astype(np.float32)
    names = ["tri_info", "texels", "emitter_rad", "cam_to_world"]
    g = GpuScene(tb)
    img_plain = g.render_c(_abi.make_opts(**kw)); rays_c = g.counters()[0]
    _, g_plain = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False); rays_plain = g.counters()[0]
    assert same_rays(rays_plain, rays_c) and rays_c > 0                              # the reverse call traced the whole value sweep
    img_keep = g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    assert same_rays(g.counters()[0], rays_c) and rel_l2(img_keep, img_plain) < 1e-6
    _, g_keep = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
    assert g.counters()[0] == 0, g.counters()                    # adjoint kernel only: the replayed hits are not traced
    for k in names:
        assert np.abs(g_plain[k]).max() > 0 and rel_l2(g_keep[k], g_plain[k]) < 1e-5, (k, rel_l2(g_keep[k], g_plain[k]))
    # the records are ONE-SHOT (round 6, ADVICE r5: tables rewritten in place under an unchanged descriptor must never meet stale records): the reverse call that
    # consumed them cleared them -- the same call again runs its own value sweep, with the same result
    _, g_again = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
    assert same_rays(g.counters()[0], rays_c) and rel_l2(g_again["tri_info"], g_plain["tri_info"]) < 1e-5
    # ... and they never serve other samples, a call that wants the image, other tables, or a call after an option changed
    g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    g.render_d_rev(_abi.make_opts(**dict(kw, rng_offset=(9, 0, 0))), adj, want=names, with_image=False)
    assert g.counters()[0] > 0
    g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    img_r, _ = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=True)
    assert same_rays(g.counters()[0], rays_c) and rel_l2(img_r, img_plain) < 3e-4
    g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    g.tb["texels"] = g.tb["texels"].clone(); g.set_guide(None)   # a new table pointer: psdr_scene_set_tables installs a different descriptor
    _, g_new = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
    assert same_rays(g.counters()[0], rays_c) and rel_l2(g_new["tri_info"], g_plain["tri_info"]) < 1e-5
    g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    g.set_option("keep_records", 1)
    g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
    assert same_rays(g.counters()[0], rays_c)
    g.set_option("keep_records", 0)
    g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
    assert same_rays(g.counters()[0], rays_c)                             # flag ignored


@pytest.mark.gpu
def test_render_c_keeps_records_on_a_scene_without_a_tree():
    """PSDR_FLAG_KEEP_RECORDS where the traced wavefront does not apply (the headline's 12-triangle cbox): the reverse launch of a PathTracer runs both sweeps in ONE
    kernel there, a third of it the value sweep.  With the flag psdr_render_c runs the VALUE KERNEL of a split launch (k_camera_rev STAGE 1: image + one record per
    path) and the psdr_render_d_rev of the same samples its adjoint kernel only: same image, same gradients, no ray counted by the reverse call."""
    from helpers import GpuScene, load_scene, rel_l2
    res, spp = 128, 8
    sc, _ = load_scene("cbox", res=res, spp=spp)
    tb = sc.tables(0)
    adj = np.random.default_rng(6).random((res * res, 3)).astype(np.float32)
    names = ["tri_info", "texels", "emitter_rad", "cam_to_world"]
    for depth in (1, 3):
        kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=spp, rng_offset=(3, 0, 0))
        g = GpuScene(tb)
        img_plain = g.render_c(_abi.make_opts(**kw)); rays_c = g.counters()[0]
        _, g_plain = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
        assert same_rays(g.counters()[0], rays_c) and rays_c > 0
        img_keep = g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
        assert same_rays(g.counters()[0], rays_c) and rel_l2(img_keep, img_plain) < 1e-5, (g.counters(), rays_c, rel_l2(img_keep, img_plain))
        _, g_keep = g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
        assert g.counters()[0] == 0, g.counters()                # the adjoint kernel re-intersects the recorded triangles: no ray
        for k in names:
            assert np.abs(g_plain[k]).max() > 0 and rel_l2(g_keep[k], g_plain[k]) < 1e-5, (depth, k, rel_l2(g_keep[k], g_plain[k]))
        # texel-only gradients (no geometry table wanted) do not use the records; other samples neither
        g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
        _, g_tex = g.render_d_rev(_abi.make_opts(**kw), adj, want=["texels"], with_image=False)
        assert same_rays(g.counters()[0], rays_c) and rel_l2(g_tex["texels"], g_plain["texels"]) < 1e-5
        g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
        g.render_d_rev(_abi.make_opts(**dict(kw, rng_offset=(4, 0, 0))), adj, want=names, with_image=False)
        assert g.counters()[0] > 0
    # a path deeper than the LDS record holds: the flag is ignored, the reverse call runs both sweeps
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=10, spp=2, rng_offset=(3, 0, 0))
    g = GpuScene(tb)
    img_plain = g.render_c(_abi.make_opts(**kw)); rays_c = g.counters()[0]
    img_keep = g.render_c(_abi.make_opts(flags=_abi.FLAG_KEEP_RECORDS, **kw))
    assert rel_l2(img_keep, img_plain) < 1e-6
    g.render_d_rev(_abi.make_opts(**kw), adj, want=names, with_image=False)
    assert same_rays(g.counters()[0], rays_c)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,mesh_key,res,spp", [("cbox_bunny", "Mesh[1]", 256, 16), ("cbox", "Mesh[0]", 128, 8)])
def test_surface_backward_reuses_the_primal_render_of_a_path_tracer(scene, mesh_key, res, spp):
    """The same through the drop-in surface (docs/inverse_diff_render.rst: renderD, a torch loss on the image, enoki.backward): the vertex gradient of the bunny
    with and without the kept records, and the reverse call's ray counter."""
    import enoki as ek
    import psdr_cuda
    from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
    from psdr_cuda.fixtures import scene_path
    out = {}
    for keep in (1, 0):
        sc = psdr_cuda.Scene()
        sc.load_file(scene_path(scene), False)
        sc.native_options = {"keep_records": keep}
        sc.opts.width = sc.opts.height = res
        sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
        integ = psdr_cuda.PathTracer(3)
        mesh = sc.param_map[mesh_key]
        v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
        sc.configure()
        img = integ.renderD(sc)
        loss = (img.t * img.t).sum().reshape(1)                  # the primal image is looked at here: rendered with PSDR_FLAG_KEEP_RECORDS
        ek.backward(FloatD._wrap(loss))
        out[keep] = (ek.gradient(v).numpy().copy(), integ.last_counters[0], img.numpy().copy())
    (ga, rays_a, ia), (gb, rays_b, ib) = out[1], out[0]
    assert rays_a == 0 and rays_b > 0, (rays_a, rays_b)
    assert np.abs(gb).max() > 0 and np.isfinite(ga).all()
    # (a scene without a tree renders its recording primal image with the value kernel of a split launch: the same estimator in another kernel)
    assert np.linalg.norm(ga - gb) < 1e-4 * np.linalg.norm(gb) and np.linalg.norm(ia - ib) < (1e-6 if scene == "cbox_bunny" else 1e-5) * np.linalg.norm(ib)
