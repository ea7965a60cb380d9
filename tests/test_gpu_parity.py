"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.  Tolerances:
  * hits (the oracle traces in double = the exact hit): identical triangle ids except rays within fp32
    round-off of a shared edge (<= 0.1 %); barycentrics <= 1e-4 abs for 99.99 % of the hits, <= 2e-3 for all
    (edge-on triangles).
  * images (fp32, same RNG streams): rel-L2 <= 1e-4 on scenes whose samples are well conditioned
    (cbox: 12 large triangles); on bunny scenes isolated ill-conditioned samples (fp32
    Moeller-Trumbore derivatives of edge-on triangles seen from ~1000 units, DESIGN.md
    "numerical fragility") flip between ANY two fp32 implementations, so the bound is on the
    fraction of differing pixels plus a looser rel-L2.
"""
import numpy as np
import pytest

import oracle
from helpers import same_rays, isolated_pixels_unbiased, GpuScene, camera_rays, load_scene, rel_l2, tangents_wrt
from psdr_cuda import _abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["cbox", "cbox_bunny", "bunny_light"])
def test_trace_matches_oracle(scene):
    sc, _ = load_scene(scene, res=64)
    tb = sc.tables(0)
    g = GpuScene(tb)
    n = 200_000
    o, d = camera_rays(tb, n, seed=1)
    # second batch: incoherent rays from first hits
    shape, tri, u, v = g.trace(o, d)
    so, stri, su, sv = oracle.trace(tb, o, d)
    same = tri == stri
    assert same.mean() > 0.999
    assert np.array_equal(shape[same], so[same])
    hit = same & (tri >= 0)
    assert hit.sum() > n * 0.2
    # the oracle traces in double (the exact hit): fp32 Moeller-Trumbore is within 1e-5 of it except on edge-on
    # triangles (|det| tiny), where a handful of rays reach 1e-3
    du = np.maximum(np.abs(u[hit] - su[hit]), np.abs(v[hit] - sv[hit]))
    print("%s: barycentric error vs the exact hit: max %.1e, 99.99th percentile %.1e" % (scene, du.max(), np.percentile(du, 99.99)))
    assert np.percentile(du, 99.99) < 1e-4 and du.max() < 2e-3
    assert np.all(u[tri < 0] == -1.0)
    # incoherent bounce rays
    info = tb["tri_info"].cpu().numpy()
    idx = np.nonzero(hit)[0][:100_000]
    p = info[tri[idx], 0:3] + u[idx, None] * info[tri[idx], 3:6] + v[idx, None] * info[tri[idx], 6:9]
    rng = np.random.default_rng(2)
    d2 = rng.normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    _, tri2, u2, v2 = g.trace(p, d2)
    _, stri2, su2, sv2 = oracle.trace(tb, p, d2)
    same2 = tri2 == stri2
    assert same2.mean() > 0.995
    h2 = same2 & (tri2 >= 0)
    assert np.abs(u2[h2] - su2[h2]).max() < 1e-3


OPTS = {
    "direct11": dict(bsdf_samples=1, light_samples=1),
    "direct20": dict(bsdf_samples=2, light_samples=0),
    "direct02": dict(bsdf_samples=0, light_samples=2),
    "path3": dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3),
    "field_depth": dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["depth"]),
}


@pytest.mark.parametrize("scene", ["cbox", "cbox_rough", "cbox_occluder"])
@pytest.mark.parametrize("kind", list(OPTS))
def test_render_c_matches_oracle(scene, kind):
    sc, _ = load_scene(scene, res=48, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=16, rng_offset=(7, 0, 0), **OPTS[kind])
    ref = oracle.render(tb, o)
    img = GpuScene(tb).render_c(o)
    assert np.isfinite(img).all()
    assert rel_l2(img, ref) < 1e-4, rel_l2(img, ref)


def test_render_c_bunny():
    sc, _ = load_scene("cbox_bunny", res=64, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=16, bsdf_samples=1, light_samples=1)
    ref = oracle.render(tb, o)
    img = GpuScene(tb).render_c(o)
    badm = np.abs(img - ref).max(axis=1) > 1e-3 * (1 + np.abs(ref).max(axis=1))
    bad = badm.mean()
    print("cbox_bunny renderC: rel-L2 %.2e, pixels off by > 1e-3: %.2e" % (rel_l2(img, ref), bad))
    assert bad < 2e-3 and rel_l2(img, ref) < 1e-3, (bad, rel_l2(img, ref))
    isolated_pixels_unbiased(img, ref, badm, "cbox_bunny renderC")              # the excluded pixels' signed errors cancel (helpers.py)


def test_shards_sum_to_full_render():
    sc, _ = load_scene("cbox", res=32, spp=8)
    tb = sc.tables(0)
    g = GpuScene(tb)
    full = g.render_c(_abi.make_opts(spp=8))
    parts = sum(g.render_c(_abi.make_opts(spp=8, spp_range=r)) for r in ((0, 3), (3, 4), (4, 8)))
    assert rel_l2(parts, full) < 1e-6


@pytest.mark.parametrize("kind", ["direct11", "direct20", "path3"])
@pytest.mark.parametrize("scene,mesh", [("cbox", 0), ("cbox_occluder", 1), ("cbox_rough", 0)])
def test_render_d_fwd_matches_oracle(scene, mesh, kind):
    sc, P = load_scene(scene, res=32, spp=8, sppe=8, sppse=8, translate=(mesh, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=8, sppe=8, sppse=8 if kind.startswith("direct") else 0, **OPTS[kind])
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = GpuScene(tb).render_d_fwd(o, [tan])
    print("%s %s renderD: image rel-L2 %.2e, derivative image rel-L2 %.2e" % (scene, kind, rel_l2(img, ref_img), rel_l2(dimg[0], ref_d)))
    assert rel_l2(img, ref_img) < 1e-4, rel_l2(img, ref_img)
    assert rel_l2(dimg[0], ref_d) < 1e-3, rel_l2(dimg[0], ref_d)


@pytest.mark.parametrize("scene,mesh,depth,res,spp", [("cbox_bunny", 1, 3, 128, 8), ("cbox_bunny", 1, 5, 96, 8), ("cbox_bunny", 0, 3, 96, 8)])
def test_path_tracer_geometry_duals_through_the_traced_wavefront(scene, mesh, depth, res, spp):
    """Round 5: renderD + enoki.forward w.r.t. a geometry parameter -- the only AD mode the reference's harness uses,
    examples/run_test.py:126-129 -- on a two-level scene runs the PathTracer as the TRACED WAVEFRONT with dual-number stages (csrc/psdr_kernels.h
    k_wfg_camera / k_wfg_bounce: the trace kernel stays float, every vertex is re-derived differentiably from its stream record as scene.cpp:346-368
    does from the OptiX hit).  >= 2^16 slots so that the library picks the wavefront by itself.  Against the oracle (the translation of the bunny / of a
    wall: image 1e-4, derivative image 1e-3 up to the isolated samples of a tree scene) and against the fused kernel on the same samples."""
    assert res * res * spp >= 1 << 16
    sc, P = load_scene(scene, res=res, spp=spp, sppe=0, sppse=0, translate=(mesh, (1.0, 0.5, 0.25)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=spp, rng_offset=(3, 0, 0))
    g = GpuScene(tb)
    img_w, d_w = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw), [tan]); rays_w = g.counters()[0]
    img_f, d_f = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw), [tan]); rays_f = g.counters()[0]
    img_0, d_0 = g.render_d_fwd(_abi.make_opts(**kw), [tan])                       # the library's own choice: the wavefront
    g.set_option("wf_geo", 0)
    img_x, d_x = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw), [tan])   # option off: the fused kernel whatever the flag says
    assert np.abs(d_w[0]).max() > 0 and np.isfinite(d_w[0]).all()
    assert rel_l2(img_0, img_w) < 2e-6 and rel_l2(d_0[0], d_w[0]) < 2e-5, (rel_l2(img_0, img_w), rel_l2(d_0[0], d_w[0]))
    assert rel_l2(img_x, img_f) < 2e-6 and rel_l2(d_x[0], d_f[0]) < 2e-5
    # separately compiled fp32 kernels: isolated samples resolve an epsilon-sized tie the other way (test_wavefront_and_fused_agree_on_tree_scenes...)
    bad = np.abs(d_w[0] - d_f[0]).max(1) > 1e-4 * (1.0 + np.abs(d_f[0]).max(1))
    print("%s mesh %d depth %d: wavefront vs fused: image %.2e, derivative %.2e, pixels apart %d of %d, rays %d / %d" % (
        scene, mesh, depth, rel_l2(img_w, img_f), rel_l2(d_w[0], d_f[0]), bad.sum(), bad.size, rays_w, rays_f))
    assert abs(rays_w - rays_f) <= 1e-4 * rays_f and bad.mean() < 2e-3
    assert rel_l2(d_w[0][~bad], d_f[0][~bad]) < 1e-3
    ref_img, ref_d = oracle.render(tb, _abi.make_opts(**kw), mode=1, tangents=tan)
    badr = np.abs(d_w[0] - ref_d).max(1) > 1e-3 * (1.0 + np.abs(ref_d).max(1))
    print("    vs oracle: image %.2e, derivative %.2e (outside %d isolated pixels: %.2e); fused vs oracle %.2e" % (
        rel_l2(img_w, ref_img), rel_l2(d_w[0], ref_d), badr.sum(), rel_l2(d_w[0][~badr], ref_d[~badr]), rel_l2(d_f[0], ref_d)))
    assert rel_l2(img_w, ref_img) < 1e-3 and badr.mean() < 2e-3 and rel_l2(d_w[0][~badr], ref_d[~badr]) < 1e-3
    isolated_pixels_unbiased(d_w[0], ref_d, badr, "geometry-dual wavefront vs oracle", bias_bound=2e-2)      # (96^2-128^2 x 8 spp: ONE flipped sample of a derivative image is ~1e-3 of its energy)
    isolated_pixels_unbiased(d_w[0], d_f[0], bad, "geometry-dual wavefront vs fused", bias_bound=2e-2)
    assert rel_l2(d_w[0], ref_d) < 2.0 * max(rel_l2(d_f[0], ref_d), 5e-4)           # no worse than the kernel it replaces


def test_path_tracer_geometry_duals_wavefront_k3_and_camera_pose():
    """K = 3 tangent sets in one pass (a translation of the bunny, a translation of a wall, a material tangent riding along) and the camera pose as the
    geometry parameter: each derivative image equals the K = 1 launch of its set, and the fused kernel's."""
    import torch
    from helpers import random_tangents
    res, spp = 96, 8
    sc, P = load_scene("cbox_bunny", res=res, spp=spp, translate=(1, (0.0, 1.0, 0.0)))
    tb = sc.tables(0)
    t_mesh = tangents_wrt(tb, P)
    t_cam = random_tangents(tb, ["cam_to_world"])
    t_mix = random_tangents(tb, ["tri_info", "texels", "emitter_rad"])
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp, rng_offset=(9, 0, 0))
    g = GpuScene(tb)
    img3, d3 = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw), [t_mesh, t_cam, t_mix])
    for i, t in enumerate((t_mesh, t_cam, t_mix)):
        _, d1 = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw), [t])
        _, df = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw), [t])
        bad = np.abs(d1[0] - df[0]).max(1) > 1e-4 * (1.0 + np.abs(df[0]).max(1))
        print("set %d: K=3 vs K=1 %.2e, wavefront vs fused %.2e (%d pixels apart)" % (i, rel_l2(d3[i], d1[0]), rel_l2(d1[0], df[0]), bad.sum()))
        assert np.abs(d1[0]).max() > 0
        assert rel_l2(d3[i], d1[0]) < 1e-4
        assert bad.mean() < 2e-3 and rel_l2(d1[0][~bad], df[0][~bad]) < 1e-3


def test_render_d_albedo_k3():
    """d image / d (r,g,b) of BSDF[0].reflectance in one K=3 pass == three K=1 oracle passes."""
    import torch
    sc, _ = load_scene("cbox", res=32, spp=8)
    tb = sc.tables(0)
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=8)
    sets = []
    for c in range(3):
        t = torch.zeros_like(tb["texels"]); t[c] = 1.0
        sets.append({"texels": t})
    img, dimg = GpuScene(tb).render_d_fwd(o, sets)
    for c in range(3):
        _, ref = oracle.render(tb, o, mode=1, tangents=sets[c])
        assert rel_l2(dimg[c], ref) < 1e-3, rel_l2(dimg[c], ref)
        assert abs(dimg[c].sum() - ref.sum()) < 1e-4 * abs(ref.sum())


def test_guiding_grid_matches_oracle():
    sc, _ = load_scene("cbox_occluder", res=32, spp=4, sppe=4, sppse=4)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=4, sppe=4, sppse=4)
    reso = (200, 4, 4, 2)
    g = GpuScene(tb)
    mass = g.guide_build(o, reso, 2)
    ref = oracle.guide_build(tb, o, reso, 2)
    assert rel_l2(mass, ref) < 1e-3


def test_guiding_grid_at_the_reference_size_on_a_tree_scene():
    """DirectIntegrator::preprocess_secondary_edges at the size the reference's own configuration uses (examples/config.py cbox_MIS: resolution
    (40000, 5, 5, 2) on cbox_bunny; direct.cpp:166-204, cube_distrb.cpp:8-62): 1 M cells x 2 sample streams, here with 2 of its 32 rounds so that
    the oracle finishes in seconds.  Mass per cell against oracle.guide_build, both launch forms (one kernel / probe + dense trace + survivors)."""
    sc, _ = load_scene("cbox_bunny", res=64, spp=4, sppe=4, sppse=4)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=4, sppe=4, sppse=4)
    reso = (40000, 5, 5, 2)
    ref = oracle.guide_build(tb, o, reso, 2)
    assert ref.shape == (1000000,) and (ref > 0).sum() > 1000
    out = {}
    for mode in (1, 0):
        g = GpuScene(tb, options={"probe": mode})
        out[mode] = g.guide_build(o, reso, 2)
        g.close()
        print("guiding (40000, 5, 5, 2) x 2 rounds, probe=%d: mass rel-L2 %.2e, non-zero cells %d / %d (oracle %d)" % (mode, rel_l2(out[mode], ref), (out[mode] > 0).sum(), ref.size, (ref > 0).sum()))
        assert rel_l2(out[mode], ref) < 1e-3, (mode, rel_l2(out[mode], ref))
    assert rel_l2(out[1], out[0]) < 1e-4


@pytest.mark.parametrize("scene", ["cbox", "cbox_rough", "bunny_light"])
def test_wavefront_path_tracer_equals_fused(scene):
    """PSDR_FLAG_WAVEFRONT (per-bounce kernels + stream compaction) and PSDR_FLAG_FUSED evaluate the same
    estimator with the same random numbers: identical up to the order of the atomic splats."""
    import torch
    sc, _ = load_scene(scene, res=64, spp=16)
    tb = sc.tables(0)
    g = GpuScene(tb)
    for depth in (1, 2, 5):
        kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=16, rng_offset=(3, 0, 0))
        a = g.render_c(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw))
        rays_f = g.counters()[0]
        b = g.render_c(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw))
        rays_w = g.counters()[0]
        assert rel_l2(b, a) < 1e-5, (depth, rel_l2(b, a))
        assert same_rays(rays_f, rays_w)
    # material-only renderD, K = 3
    sets = [{"texels": torch.eye(tb["texels"].numel())[c]} for c in range(3)]
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16)
    ia, da = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw), sets)
    ib, db = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw), sets)
    # (the fused launch is the log-derivative kernel -- the estimator on plain floats -- the wavefront's stages carry dual numbers: the primal images differ
    # in the last bits of a few samples, 1.05e-5 on the bunny)
    assert rel_l2(ib, ia) < 3e-5 and rel_l2(db, da) < 1e-4


@pytest.mark.parametrize("scene,depth", [("cbox_bunny", 3), ("cbox_bunny", 6), ("interior", 3), ("interior", 6)])
def test_wavefront_and_fused_agree_on_tree_scenes_up_to_isolated_samples(scene, depth):
    """The two strategies are separately compiled fp32 kernels: the same estimator on the same random numbers, but a product contracted
    into an FMA in one and not in the other moves a bounce ray by an ulp, and on a tree scene a handful of rays per million then resolve an
    epsilon-sized tie the other way (a different triangle at an edge, a GGX sample on the other side of its pdf cut-off).  What must hold:
    the same number of rays, all but a few pixels equal to 1e-5, and the default strategy is a pure function of the scene and the
    options -- the same call returns the same image whatever was rendered on the handle before."""
    if scene == "interior":
        from psdr_cuda.fixtures import make_interior_scene
        sc = make_interior_scene(seed=0, n_objects=10, res=96, spp=16); sc.configure()
    else:
        sc, _ = load_scene(scene, res=96, spp=16)
    tb = sc.tables(0)
    g = GpuScene(tb)
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=16, rng_offset=(5, 0, 0))
    a = g.render_c(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw)); rays_f = g.counters()[0]
    b = g.render_c(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw)); rays_w = g.counters()[0]
    # diffuse scene: equal to 1e-5 but for isolated samples.  GGX lobes of alpha = 0.05 amplify the ulp: there BOTH kernels sit ~1e-4 per
    # pixel from the fp32 oracle (tools/wf_diff_probe.py: 1 107 / 1 147 of 9 216 pixels off by > 1e-5 at depth 3, 153 between the two)
    tol = 1e-5 if scene == "cbox_bunny" else 1e-3
    bad = np.abs(a - b).max(1) > tol * (1.0 + np.abs(a).max(1))
    print("%s depth %d: rel-L2 %.2e, pixels apart by > %g: %d of %d, rays %d / %d" % (scene, depth, rel_l2(b, a), tol, bad.sum(), bad.size, rays_f, rays_w))
    assert abs(rays_f - rays_w) <= 1e-4 * rays_f
    if scene == "cbox_bunny":
        assert bad.mean() < 2e-3
    else:
        # GGX interior: how many pixels differ between two fp32 builds moves with every recompilation (0.2-0.8 % at depth 3-6); what must hold is that
        # NEITHER strategy is further from the fp32 oracle than the other -- the differing pixels are ties resolved either way, not errors of one kernel
        ref = oracle.render(tb, _abi.make_opts(**kw))
        off_a = (np.abs(a - ref).max(1) > tol * (1.0 + np.abs(ref).max(1))).mean()
        off_b = (np.abs(b - ref).max(1) > tol * (1.0 + np.abs(ref).max(1))).mean()
        print("    pixels off the fp32 oracle by > %g: fused %.2e, wavefront %.2e; apart from each other %.2e" % (tol, off_a, off_b, bad.mean()))
        assert bad.mean() < 2e-2 and off_b < 1.5 * off_a + 2e-3 and off_a < 1.5 * off_b + 2e-3 and max(off_a, off_b) < 3e-2
    assert rel_l2(b[~bad], a[~bad]) < 10 * tol
    isolated_pixels_unbiased(b, a, bad, "%s depth %d wavefront vs fused" % (scene, depth), bias_bound=2e-3)
    # default strategy: decided by the scene and the options alone -- first call on a fresh handle, and again after other calls
    first = GpuScene(tb).render_c(_abi.make_opts(**kw))
    g.render_c(_abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=2, spp=16)); g.counters()
    later = g.render_c(_abi.make_opts(**kw))
    assert rel_l2(later, first) < 2e-6, rel_l2(later, first)                   # same kernels, same samples: the order of the atomic splats only
    assert min(rel_l2(first, a), rel_l2(first, b)) < 2e-6


@pytest.mark.parametrize("scene", ["cbox", "cbox_rough"])
def test_kernel_variant_choice_does_not_change_the_image(scene):
    """psdr_scene_desc.material_mask picks the kernel variant (all-diffuse scenes run without the GGX code);
    mask 0 = unknown = the general variant.  Same estimator either way."""
    sc, _ = load_scene(scene, res=48, spp=8)
    tb = sc.tables(0)
    assert tb["material_mask"] == (1 if scene == "cbox" else 3)
    tb0 = dict(tb); tb0["material_mask"] = 0
    for kind in ("direct11", "path3"):
        o = _abi.make_opts(spp=8, **OPTS[kind])
        a, b = GpuScene(tb).render_c(o), GpuScene(tb0).render_c(o)
        assert rel_l2(a, b) < 1e-6, kind
    import torch
    t = {"texels": torch.rand(tb["texels"].shape, generator=torch.Generator().manual_seed(0))}
    o = _abi.make_opts(spp=8, **OPTS["path3"])
    da, db = GpuScene(tb).render_d_fwd(o, [t])[1][0], GpuScene(tb0).render_d_fwd(o, [t])[1][0]
    assert rel_l2(da, db) < 1e-5
    adj = np.random.default_rng(0).random((48 * 48, 3)).astype(np.float32)
    ga = GpuScene(tb).render_d_rev(o, adj, want=["texels"])[1]["texels"]
    gb = GpuScene(tb0).render_d_rev(o, adj, want=["texels"])[1]["texels"]
    assert rel_l2(ga, gb) < 1e-4


def test_anisotropic_rough_conductor_matches_oracle():
    from test_rough_conductor import metal_floor
    import torch
    tb = metal_floor(0.1, 0.4, res=32, spp=16).tables(0)
    g = GpuScene(tb)
    for kind in ("direct11", "direct20", "path3"):
        o = _abi.make_opts(spp=16, **OPTS[kind])
        img, ref = g.render_c(o), oracle.render(tb, o)
        assert rel_l2(img, ref) < 1e-4, kind
    t = {"texels": torch.rand(tb["texels"].shape, generator=torch.Generator().manual_seed(1))}
    o = _abi.make_opts(spp=16, **OPTS["path3"])
    _, d = g.render_d_fwd(o, [t])
    _, rd = oracle.render(tb, o, mode=1, tangents=t)
    assert rel_l2(d[0], rd) < 2e-3


def test_wavefront_zeroes_whole_samples_like_the_fused_kernel():
    """masked(value, ~isfinite(value)) = 0 acts on the WHOLE sample (integrator.cpp:87): a path whose second or third vertex lands on
    a surface with a NaN albedo loses its finite first-vertex contribution too.  The wavefront carries the radiance gathered so far in
    its stream records and splats a sample once, when its path ends -- the same rule, whatever strategy the library picks."""
    import torch
    sc, _ = load_scene("cbox", res=64, spp=16)
    tb = dict(sc.tables(0))
    rec = tb["bsdf_rec"].cpu().numpy()
    # the albedo of ONE wall that is also seen directly: find the diffuse BSDF whose mesh has the most triangles hit by camera rays
    tex = tb["texels"].clone()
    g0 = GpuScene(tb)
    o, d = camera_rays(tb, 20000, seed=1)
    _, tri, _, _ = g0.trace(o, d)
    mesh = (tb["tri_mesh"].cpu().numpy()[tri[tri >= 0]] & 0x3fffffff)
    bsdf = tb["mesh_bsdf"].cpu().numpy()[mesh]
    ids, cnt = np.unique(bsdf[bsdf >= 0], return_counts=True)
    victim = int(ids[np.argsort(cnt)[len(cnt) // 2]])                  # neither the most nor the least visible one
    off = int(rec[victim, 1])
    tex[off] = float("nan")
    tb["texels"] = tex
    g = GpuScene(tb)
    for depth in (2, 3):
        kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=16, rng_offset=(3, 0, 0))
        a = g.render_c(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw))
        b = g.render_c(_abi.make_opts(flags=_abi.FLAG_WAVEFRONT, **kw))
        clean = GpuScene(sc.tables(0)).render_c(_abi.make_opts(flags=_abi.FLAG_FUSED, **kw))
        assert np.isfinite(a).all() and np.isfinite(b).all()
        lost = (a[:, 0] < clean[:, 0] - 1e-4).mean()                     # samples zeroed in the red channel
        assert lost > 0.2, lost
        assert rel_l2(b, a) < 1e-5, (depth, rel_l2(b, a))


@pytest.mark.parametrize("scene,mesh,res,spp", [("cbox_occluder", 1, 48, 8), ("cbox_bunny", 1, 64, 16), ("bunny_light", 0, 64, 16)])
def test_literal_form_flag_on_the_gpu(scene, mesh, res, spp):
    """PSDR_FLAG_LITERAL_FORMS makes the fp32-robust forms an A/B on the device.  With the flag the HIP
    kernels match oracle.render(reference_form=True, precision=0) sample for sample; without it the robust-form oracle; and on the bunny
    scenes the two forms are measurably apart in fp32 while fp64 holds them together (the reason the product evaluates the robust ones)."""
    from helpers import tangents_wrt
    sc, P = load_scene(scene, res=res, spp=spp, sppe=spp, sppse=spp, translate=(mesh, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    g = GpuScene(tb)
    kw = dict(spp=spp, sppe=spp, sppse=spp, rng_offset=(0, 3, 7))
    img_l, d_l = g.render_d_fwd(_abi.make_opts(flags=_abi.FLAG_LITERAL_FORMS, **kw), [tan])
    img_r, d_r = g.render_d_fwd(_abi.make_opts(**kw), [tan])
    o_l = oracle.render(tb, _abi.make_opts(**kw), mode=1, tangents=tan, precision=0, reference_form=True)
    o_r = oracle.render(tb, _abi.make_opts(**kw), mode=1, tangents=tan, precision=0, reference_form=False)
    o64 = oracle.render(tb, _abi.make_opts(**kw), mode=1, tangents=tan, precision=1, reference_form=True)
    e_ll, e_rr, e_lr = rel_l2(d_l[0], o_l[1]), rel_l2(d_r[0], o_r[1]), rel_l2(d_l[0], d_r[0])
    print("%s: literal vs oracle literal %.2e, robust vs oracle robust %.2e, literal vs robust (GPU) %.2e; against fp64 literal: GPU literal %.2e, GPU robust %.2e"
          % (scene, e_ll, e_rr, e_lr, rel_l2(d_l[0], o64[1]), rel_l2(d_r[0], o64[1])))
    def flips(a, b):
        bad = np.abs(a - b).max(1) > 1e-3 * (1.0 + np.abs(b).max(1))
        return bad.mean(), rel_l2(a[~bad], b[~bad])
    for a, b in ((img_l, o_l[0]), (img_r, o_r[0])):
        f, r = flips(a, b)
        assert f < 0.01 and r < 1e-3, (f, r)                                     # the same estimator: all but isolated samples
    if scene == "cbox_occluder":                                               # well-conditioned scene (distances ~ 100 units): sample for sample in both forms
        assert rel_l2(img_l, o_l[0]) < 1e-4 and rel_l2(img_r, o_r[0]) < 1e-4 and e_ll < 1e-3 and e_rr < 1e-3
    # the product's default (robust) forms are at least as close to the exact (fp64) value of the reference's estimator as the literal ones
    # evaluated in fp32 (cbox_bunny at this size: 5.5e-3 against 4.5e-2 -- the literal tests flip high-weight boundary samples)
    assert rel_l2(d_r[0], o64[1]) <= rel_l2(d_l[0], o64[1]) * 1.05 + 1e-4
    lib = g.lib
    import ctypes as C
    adj = torch_adj = None
    rc = lib.psdr_render_d_rev(g.h, C.byref(_abi.make_opts(flags=_abi.FLAG_LITERAL_FORMS, **kw)), C.c_void_p(8), None, C.byref(_abi.Grads()), None)
    assert rc != 0 and b"LITERAL" in lib.psdr_last_error()                     # forward-mode diagnostic only


@pytest.mark.parametrize("scene", ["cbox_bunny", "interior"])
def test_trace_kernel_workgroup_layouts_find_the_same_hits(scene):
    """The dense trace kernel of the traced wavefront runs as one workgroup per CU with the whole LDS or as two with half of it each
    (psdr_hip.hip launch_wf_trace, option trace_wg2: shorter stack columns -- the rest in the overflow columns -- and fewer staged node
    rows).  Same walk, same leaf test: the same image whichever layout traces, down to stack columns of 2 entries (nearly every push of
    a walk then lands in an overflow column)."""
    if scene == "interior":
        from psdr_cuda.fixtures import make_interior_scene
        sc = make_interior_scene(seed=0, n_objects=10, res=96, spp=16); sc.configure()
    else:
        sc, _ = load_scene(scene, res=96, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16, flags=_abi.FLAG_WAVEFRONT)
    ref, rays = None, None
    for wg2 in (0, 8, 2, -1):
        g = GpuScene(tb, options={"trace_wg2": wg2})
        img = g.render_c(o)
        r = g.counters()[0]
        if ref is None:
            ref, rays = img, r
        else:
            assert rel_l2(img, ref) < 2e-6 and r == rays, (wg2, rel_l2(img, ref), r, rays)          # the order of the atomic splats only


def test_excluded_pixels_are_single_flipped_samples():
    """Attribution of the pixels the tree-scene tests exclude: the sample streams are stateless, so one launch over the samples [s, s + 1) of every
    pixel IS sample s of the full render -- GPU and oracle are rendered sample by sample (cbox_bunny 256 x 256, spp 8, PathTracer(3), several stream offsets until >= 100
    pixels differ by > 1e-3) and every differing pixel is taken apart: in how many of its 8 samples do the two fp32 evaluations disagree?  An epsilon tie resolved the other
    way is ONE sample whose path takes another turn (rarely two in one pixel); a kernel bug on a rare branch would show as pixels whose samples ALL carry a small error, or
    as one-signed errors (isolated_pixels_unbiased).  Asserted: >= 90 % of the differing pixels hold exactly one differing sample, none more than three, their other samples
    agree to 1e-4, and the per-sample images sum to the full render."""
    res, spp = 256, 8
    sc, _ = load_scene("cbox_bunny", res=res, spp=spp)
    tb = sc.tables(0)
    g = GpuScene(tb)
    counts, signed, total_bad = [], np.zeros(3), 0
    for off in range(12):
        kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp, rng_offset=(11 * off, 0, 0))
        a, ref = g.render_c(_abi.make_opts(**kw)), oracle.render(tb, _abi.make_opts(**kw))
        bad = np.abs(a - ref).max(1) > 1e-3 * (1 + np.abs(ref).max(1))
        if not bad.any():
            continue
        a_s = np.stack([g.render_c(_abi.make_opts(spp_range=(s, s + 1), **kw)) for s in range(spp)])
        r_s = np.stack([oracle.render(tb, _abi.make_opts(spp_range=(s, s + 1), **kw)) for s in range(spp)])
        assert rel_l2(a_s.sum(0), a) < 1e-5 and rel_l2(r_s.sum(0), ref) < 1e-5          # the shards ARE the samples of the full render
        diff = np.abs(a_s - r_s).max(2) > 1e-4 * (1.0 / spp + np.abs(r_s).max(2))           # [spp, pixels]: this sample differs
        n = diff[:, bad].sum(0)
        counts += list(n)
        # the samples that do NOT differ agree to round-off in the differing pixels too
        same = ~diff[:, bad]
        assert np.abs((a_s - r_s)[:, bad][same]).max() <= 1e-4 * (1.0 / spp + np.abs(r_s[:, bad][same]).max())
        signed += (a - ref)[bad].sum(0)
        total_bad += int(bad.sum())
        if total_bad >= 100:
            break
    counts = np.array(counts)
    print("excluded pixels taken apart: %d pixels, differing samples per pixel: %s, signed error sum %s" % (len(counts), np.bincount(counts, minlength=4)[:6], signed))
    if len(counts) >= 20:
        assert (counts == 1).mean() >= 0.9 and counts.max() <= 3 and counts.min() >= 1, np.bincount(counts)
