"""The PRODUCT's estimator code (csrc/psdr_device.h, the functions the HIP kernels execute) run on the
host by tests/hostcheck and compared with the independent oracle on the same seeded inputs.  This is
the CPU-side half of the parity gate; tests/test_gpu_parity.py repeats it through the C ABI on the GPU.
fp32 both sides, same RNG streams: agreement to round-off on well-conditioned scenes."""
import numpy as np
import pytest

import oracle
from helpers import host_render, load_scene, rel_l2, tangents_wrt
from psdr_cuda import _abi

OPTS = {
    "direct11": dict(bsdf_samples=1, light_samples=1),
    "direct20": dict(bsdf_samples=2, light_samples=0),
    "direct02": dict(bsdf_samples=0, light_samples=2),
    "direct22": dict(bsdf_samples=2, light_samples=2),
    "path1": dict(integrator=_abi.INTEGRATOR_PATH, max_depth=1),
    "path4": dict(integrator=_abi.INTEGRATOR_PATH, max_depth=4),
    "field_pos": dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["position"]),
    "field_shn": dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["shNormal"]),
}


@pytest.mark.parametrize("scene", ["cbox", "cbox_rough", "cbox_occluder"])
@pytest.mark.parametrize("kind", list(OPTS))
def test_render_c(scene, kind):
    sc, _ = load_scene(scene, res=24, spp=8)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=8, rng_offset=(3, 0, 0), **OPTS[kind])
    assert rel_l2(host_render(tb, o), oracle.render(tb, o)) < 2e-5


def test_render_c_hide_emitters_and_shards():
    sc, _ = load_scene("cbox", res=16, spp=6)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=6, hide_emitters=True)
    full = host_render(tb, o)
    assert rel_l2(full, oracle.render(tb, o)) < 2e-5
    assert full.max() < 19.0
    parts = sum(host_render(tb, _abi.make_opts(spp=6, hide_emitters=True, spp_range=r)) for r in ((0, 2), (2, 6)))
    assert rel_l2(parts, full) < 1e-6


def test_render_c_bunny_bvh():
    sc, _ = load_scene("cbox_bunny", res=32, spp=4)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=4)
    a, b = host_render(tb, o), oracle.render(tb, o)
    bad = (np.abs(a - b).max(1) > 1e-3 * (1 + np.abs(b).max(1))).mean()
    assert bad < 0.01 and rel_l2(a, b) < 2e-2


@pytest.mark.parametrize("kind", ["direct11", "direct20", "direct02"])
@pytest.mark.parametrize("scene,mesh", [("cbox", 0), ("cbox_occluder", 1), ("cbox_rough", 0)])
def test_render_d_forward_all_terms(scene, mesh, kind):
    sc, P = load_scene(scene, res=24, spp=8, sppe=8, sppse=8, translate=(mesh, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=8, sppe=8, sppse=8, rng_offset=(0, 5, 9), **OPTS[kind])
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = host_render(tb, o, mode=1, tangents=tan)
    assert rel_l2(img, ref_img) < 2e-5
    assert rel_l2(dimg, ref_d) < 1e-4 and np.abs(ref_d).max() > 0


def test_render_d_material_and_emitter_tangents():
    import torch
    sc, _ = load_scene("cbox_rough", res=24, spp=8)
    tb = sc.tables(0)
    g = torch.Generator().manual_seed(0)
    tan = {"texels": torch.rand(tb["texels"].shape, generator=g), "emitter_rad": torch.rand(tb["emitter_rad"].shape, generator=g)}
    for kind in ("direct11", "path4"):
        o = _abi.make_opts(spp=8, **OPTS[kind])
        ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
        img, dimg = host_render(tb, o, mode=1, tangents=tan)
        assert rel_l2(dimg, ref_d) < 1e-3, kind      # GGX tangents through 4 bounces, fp32 round-off


def test_render_d_camera_tangent():
    import torch
    sc, _ = load_scene("cbox", res=24, spp=8)
    tb = sc.tables(0)
    d = torch.zeros(4, 4); d[0, 3] = 1.0; d[1, 3] = 0.3      # translate the camera
    o = _abi.make_opts(spp=8)
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents={"cam_to_world": d})
    img, dimg = host_render(tb, o, mode=1, tangents={"cam_to_world": d})
    assert rel_l2(dimg, ref_d) < 1e-4 and np.abs(ref_d).max() > 0


def test_render_d_path_tracer_geometry():
    """PathTracer D mode: the primary vertex o + t*d (solid-angle form, scene.cpp:355-376) sits up to
    ~1e-4 off the surface in fp32, so a grazing continuation ray can re-hit its own wall in one
    implementation and not in the other (DESIGN.md 'numerical fragility'): bound the outliers."""
    sc, P = load_scene("cbox", res=24, spp=8, sppe=8, translate=(0, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=8, sppe=8, **OPTS["path4"])
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = host_render(tb, o, mode=1, tangents=tan)
    bad = (np.abs(img - ref_img).max(1) > 1e-3 * (1 + np.abs(ref_img).max(1))).mean()
    assert bad < 0.02 and rel_l2(img, ref_img) < 3e-2 and rel_l2(dimg, ref_d) < 3e-2


def test_guided_secondary_edges():
    import ctypes as C
    from helpers import hostcheck_lib
    from psdr_cuda.scene import make_desc
    sc, P = load_scene("cbox_occluder", res=16, spp=4, sppe=0, sppse=16, translate=(1, (1.0, 0.0, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=4, sppse=16, spp_range=(0, 0))
    reso = (64, 4, 4, 2)
    mass = oracle.guide_build(tb, o, reso, 2)
    desc, keep = make_desc({k: (v.detach().cpu() if hasattr(v, "detach") else v) for k, v in tb.items()}, None, device="cpu")
    m2 = np.zeros_like(mass)
    r = (C.c_int * 4)(*reso)
    assert hostcheck_lib().hostcheck_guide(C.byref(desc), r, 2, C.c_void_p(m2.ctypes.data), 4) == 0
    assert mass.sum() > 0 and rel_l2(m2, mass) < 1e-4
    import torch
    from psdr_cuda.core import DiscreteDistribution
    d = DiscreteDistribution(); d.init(torch.as_tensor(mass))
    guide = (reso[:3], d.m_cmf, d.m_pmf, d.m_sum)
    _, ref_d = oracle.render(tb, o, mode=1, tangents=tan, guide=guide)
    _, dimg = host_render(tb, o, mode=1, tangents=tan, guide=guide)
    assert np.abs(ref_d).max() > 0 and rel_l2(dimg, ref_d) < 1e-3


def test_minimal_and_degenerate_scenes():
    """1-triangle scene (the BVH root is a leaf) and a mesh with a zero-area face: product code vs oracle"""
    from test_edge_cases_gpu import TRI, tiny_scene
    for verts, faces in (TRI, ([[-1, -1, 0], [1, -1, 0], [0, 1, 0], [2, 2, 0], [2, 2, 0], [2, 2, 0]], [[0, 1, 2], [3, 4, 5]])):
        tb = tiny_scene(verts, faces).tables(0)
        for kw in (OPTS["direct11"], OPTS["path4"], OPTS["field_pos"]):
            o = _abi.make_opts(spp=4, **kw)
            a, b = host_render(tb, o), oracle.render(tb, o)
            assert np.isfinite(a).all() and rel_l2(a, b) < 2e-5


@pytest.mark.parametrize("scene,mesh", [("cbox_occluder", 1), ("cbox", 0)])
def test_literal_form_flag_matches_the_oracles_reference_form(scene, mesh):
    """PSDR_FLAG_LITERAL_FORMS: the product evaluates p = ray(t), the edge rays without the adjacent-face skip and the fp32
    |its1.p - p1| < ShadowEpsilon test as the reference writes them (scene.cpp:368, direct.cpp:246-262) -- sample for sample the oracle's
    reference_form in fp32; without the flag it matches the oracle's robust forms (product code on the host)."""
    sc, P = load_scene(scene, res=24, spp=8, sppe=8, sppse=8, translate=(mesh, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    kw = dict(spp=8, sppe=8, sppse=8, rng_offset=(0, 5, 9), **OPTS["direct11"])
    lit = host_render(tb, _abi.make_opts(flags=_abi.FLAG_LITERAL_FORMS, **kw), mode=1, tangents=tan)
    ref = oracle.render(tb, _abi.make_opts(**kw), mode=1, tangents=tan, precision=0, reference_form=True)
    assert rel_l2(lit[0], ref[0]) < 2e-5 and rel_l2(lit[1], ref[1]) < 1e-4, (rel_l2(lit[0], ref[0]), rel_l2(lit[1], ref[1]))
    rob = host_render(tb, _abi.make_opts(**kw), mode=1, tangents=tan)
    ref_rob = oracle.render(tb, _abi.make_opts(**kw), mode=1, tangents=tan, precision=0, reference_form=False)
    assert rel_l2(rob[1], ref_rob[1]) < 1e-4
