"""Textured parameters: bilinear Bitmap lookups (src/core/bitmap.cpp:41-89) in the
kernels, derivative w.r.t. texels (forward) and per-texel gradient scatter-add (reverse)."""
import numpy as np
import pytest
import torch

import oracle
import psdr_cuda
from enoki.cuda_autodiff import Vector3f as Vector3fD
from helpers import GpuScene, dot_tables, host_render, host_render_rev, random_tangents, rel_l2
from psdr_cuda import _abi
from psdr_cuda.fixtures import scene_path


def textured_scene(res=24, spp=8):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("cbox_uv"), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
    g = torch.Generator().manual_seed(7)
    w, h = 6, 5
    data = torch.rand(w * h, 3, generator=g) * 0.8 + 0.1
    sc.param_map["BSDF[id=floor_tex]"].reflectance = psdr_cuda.Bitmap3fD(w, h, Vector3fD(data))
    sc.configure()
    return sc


def test_tables_carry_uv_and_texture():
    sc = textured_scene()
    tb = sc.tables(0)
    assert tb["tri_uv"] is not None and tb["tri_uv"].shape == (12, 8)
    rec = tb["bsdf_rec"].cpu().numpy()
    tex = [r for r in rec if r[2] == 6 and r[3] == 5]
    assert len(tex) == 1 and tb["texels"].numel() >= 6 * 5 * 3


@pytest.mark.parametrize("kw", [dict(bsdf_samples=1, light_samples=1), dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)])
def test_textured_render_host_vs_oracle(kw):
    sc = textured_scene()
    tb = sc.tables(0)
    o = _abi.make_opts(spp=8, **kw)
    ref = oracle.render(tb, o)
    assert rel_l2(host_render(tb, o), ref) < 2e-5
    tan = random_tangents(tb, ["texels"], seed=3)
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = host_render(tb, o, mode=1, tangents=tan)
    assert rel_l2(dimg, ref_d) < 1e-3 and np.abs(ref_d).max() > 0
    adj = np.random.default_rng(1).random((24 * 24, 3)).astype(np.float32)
    # (the primary hit is evaluated on-surface when only material tables are differentiated and in the solid-angle
    # form when a geometry table is: forward and reverse follow the same rule, so compare like with like)
    _, grads = host_render_rev(tb, o, adj, want=["texels"])
    lhs, rhs = float((adj.astype(np.float64) * dimg).sum()), dot_tables(grads, tan)
    assert abs(lhs - rhs) < 1e-4 * np.abs(adj * dimg).sum()
    _, grads = host_render_rev(tb, o, adj, want=["texels", "tri_info"])
    # uv adjoint path: a geometry tangent moves the texture lookup of the primary hit
    tg = random_tangents(tb, ["tri_info"], seed=4)
    ref_img, ref_dg = oracle.render(tb, o, mode=1, tangents=tg)
    _, dimg_g = host_render(tb, o, mode=1, tangents=tg)
    bad = (np.abs(dimg_g - ref_dg).max(1) > 1e-3 * (1 + np.abs(ref_dg).max(1))).mean()
    assert bad < 0.02, bad        # isolated fp32-fragile samples only
    lhs, rhs = float((adj.astype(np.float64) * dimg_g).sum()), dot_tables(grads, tg)
    assert abs(lhs - rhs) < 1e-4 * np.abs(adj * dimg_g).sum()


@pytest.mark.gpu
def test_textured_render_gpu():
    sc = textured_scene(res=48, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16)
    g = GpuScene(tb)
    assert rel_l2(g.render_c(o), oracle.render(tb, o)) < 1e-4
    tan = random_tangents(tb, ["texels"], seed=3)
    _, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = g.render_d_fwd(o, [tan])
    assert rel_l2(dimg[0], ref_d) < 3e-2
    adj = np.random.default_rng(1).random((48 * 48, 3)).astype(np.float32)
    _, grads = g.render_d_rev(o, adj, want=["texels"], with_image=False)
    lhs, rhs = float((adj.astype(np.float64) * dimg[0]).sum()), dot_tables(grads, tan)
    assert abs(lhs - rhs) < 1e-3 * np.abs(adj * dimg[0]).sum()


def rough_textured_scene(res=24, spp=8):
    """the uv-mapped floor as a rough conductor whose alpha_u / alpha_v / eta / k / specular reflectance are ALL
    bitmaps of different resolutions (roughconductor.cpp:40-92 does five texture lookups per call)"""
    from enoki.cuda_autodiff import Float32 as FloatD
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("cbox_uv"), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
    g = torch.Generator().manual_seed(11)
    metal = psdr_cuda.RoughConductor(0.2, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))
    metal.alpha_u = psdr_cuda.Bitmap1fD(4, 3, FloatD(torch.rand(12, generator=g) * 0.3 + 0.08))
    metal.alpha_v = psdr_cuda.Bitmap1fD(2, 5, FloatD(torch.rand(10, generator=g) * 0.3 + 0.08))
    metal.eta = psdr_cuda.Bitmap3fD(3, 3, Vector3fD(torch.rand(9, 3, generator=g) * 1.0 + 0.2))
    metal.k = psdr_cuda.Bitmap3fD(2, 2, Vector3fD(torch.rand(4, 3, generator=g) * 3.0 + 1.0))
    metal.specular_reflectance = psdr_cuda.Bitmap3fD(5, 4, Vector3fD(torch.rand(20, 3, generator=g) * 0.5 + 0.5))
    metal.m_anisotropic = True
    metal.id = "rough_tex"
    sc.add_bsdf(metal)
    floor = [m for m in sc.m_meshes if m.m_has_uv][0]
    floor.bsdf = metal
    floor.use_face_normals = False
    sc.configure()
    return sc


@pytest.mark.parametrize("kw", [dict(bsdf_samples=1, light_samples=1), dict(integrator=_abi.INTEGRATOR_PATH, max_depth=2)])
def test_all_five_rough_conductor_textures_host(kw):
    sc = rough_textured_scene()
    tb = sc.tables(0)
    rec = tb["bsdf_rec"].cpu().numpy()
    r = [x for x in rec if x[0] == _abi.BSDF_ROUGHCONDUCTOR and x[2] == 5][0]
    assert [tuple(r[1 + 3 * s + 1:1 + 3 * s + 3]) for s in range(5)] == [(5, 4), (4, 3), (2, 5), (3, 3), (2, 2)]
    o = _abi.make_opts(spp=8, **kw)
    assert rel_l2(host_render(tb, o), oracle.render(tb, o)) < 2e-5
    tan = random_tangents(tb, ["texels"], seed=5)
    _, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    _, dimg = host_render(tb, o, mode=1, tangents=tan)
    assert rel_l2(dimg, ref_d) < 2e-3 and np.abs(ref_d).max() > 0
    adj = np.random.default_rng(2).random((24 * 24, 3)).astype(np.float32)
    _, grads = host_render_rev(tb, o, adj, want=["texels"])
    lhs, rhs = float((adj.astype(np.float64) * dimg).sum()), dot_tables(grads, tan)
    assert abs(lhs - rhs) < 2e-4 * np.abs(adj * dimg).sum()


@pytest.mark.gpu
def test_all_five_rough_conductor_textures_gpu():
    sc = rough_textured_scene(res=32, spp=16)
    tb = sc.tables(0)
    g = GpuScene(tb)
    o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    assert rel_l2(g.render_c(o), oracle.render(tb, o)) < 1e-4
    tan = random_tangents(tb, ["texels"], seed=5)
    _, d = g.render_d_fwd(o, [tan])
    _, rd = oracle.render(tb, o, mode=1, tangents=tan)
    assert rel_l2(d[0], rd) < 2e-3
    adj = np.random.default_rng(2).random((32 * 32, 3)).astype(np.float32)
    _, grads = g.render_d_rev(o, adj, want=["texels"], with_image=False)
    lhs, rhs = float((adj.astype(np.float64) * d[0]).sum()), dot_tables(grads, tan)
    assert abs(lhs - rhs) < 1e-3 * np.abs(adj * d[0]).sum()
