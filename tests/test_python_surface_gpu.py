"""End-to-end through the drop-in Python surface on the GPU: the call sequences of the reference's
harness (examples/run_test.py run_orig / run_ad / run_fd, utils/differential.py, docs/inverse_diff_render.rst)."""
import numpy as np
import pytest
import torch

import enoki as ek
import oracle
import psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
from helpers import rel_l2, tangents_wrt
from psdr_cuda import _abi
from psdr_cuda.fixtures import scene_path

pytestmark = pytest.mark.gpu


def _scene(name, res=32, spp=8, sppe=0, sppse=0):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    return sc


def test_run_orig_sequence():
    sc = _scene("cbox")
    sc.configure()
    integ = psdr_cuda.DirectIntegrator(bsdf_samples=2, light_samples=2)
    a = integ.renderC(sc, 0).numpy()
    b = integ.renderC(sc, 0).numpy()                 # streams continue: a new, independent pass
    assert a.shape == (32 * 32, 3) and a.dtype == np.float32
    assert not np.array_equal(a, b) and abs(a.mean() - b.mean()) < 0.1 * a.mean()
    ref = oracle.render(sc.tables(0), _abi.make_opts(bsdf_samples=2, light_samples=2, spp=8))
    assert rel_l2(a, ref) < 1e-4
    ref2 = oracle.render(sc.tables(0), _abi.make_opts(bsdf_samples=2, light_samples=2, spp=8, rng_offset=(12, 0, 0)))
    assert rel_l2(b, ref2) < 1e-4                    # second pass == streams advanced by 2 + 3*2 + 2*2 draws


def test_run_ad_sequence_mesh_transform_forward():
    sc = _scene("cbox_occluder", spp=8, sppe=8, sppse=8)
    integ = psdr_cuda.DirectIntegrator(1, 1)
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P))
    sc.configure()
    img = integ.renderD(sc, 0)
    ek.forward(P, free_graph=True)
    grad_img = ek.gradient(img).numpy()
    assert grad_img.shape == (32 * 32, 3) and np.isfinite(grad_img).all() and np.abs(grad_img).max() > 0
    tb = sc.tables(0)
    ref_img, ref_d = oracle.render(tb, _abi.make_opts(spp=8, sppe=8, sppse=8), mode=1, tangents=tangents_wrt(tb, P))
    assert rel_l2(img.numpy(), ref_img) < 1e-4 and rel_l2(grad_img, ref_d) < 1e-3


def test_material_roughness_forward():
    sc = _scene("cbox_rough", spp=8)
    bs = sc.param_map["BSDF[id=metal]"]
    base = (ek.detach(bs.alpha_u.data), ek.detach(bs.alpha_v.data))
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    bs.alpha_u.data = FloatD(base[0]) + P
    bs.alpha_v.data = FloatD(base[1]) + P
    sc.configure()
    img = psdr_cuda.DirectIntegrator(1, 1).renderD(sc)
    ek.forward(P)
    g = ek.gradient(img).numpy()
    tb = sc.tables(0)
    _, ref = oracle.render(tb, _abi.make_opts(spp=8), mode=1, tangents=tangents_wrt(tb, P))
    assert np.abs(ref).max() > 0 and rel_l2(g, ref) < 1e-3


def test_backward_albedo_and_vertices():
    """docs/inverse_diff_render.rst: loss.backward -> gradients of reflectance texels and vertex positions."""
    sc = _scene("cbox", spp=8, sppe=4, sppse=4)
    refl = sc.param_map["BSDF[0]"].reflectance
    ek.set_requires_gradient(refl.data)
    mesh = sc.param_map["Mesh[0]"]
    v = Vector3fD(ek.detach(mesh.vertex_positions))
    ek.set_requires_gradient(v)
    mesh.vertex_positions = v
    sc.configure()
    integ = psdr_cuda.DirectIntegrator(1, 1)
    img = integ.renderD(sc, 0)
    target = torch.full_like(img.t, 0.3)
    w = torch.linspace(0.5, 1.5, img.t.numel(), device=img.t.device).reshape(img.t.shape)
    loss = FloatD._wrap((w * (img.t - target) ** 2).sum().reshape(1))
    ek.backward(loss)
    g_refl = ek.gradient(refl.data).numpy()
    g_v = ek.gradient(v).numpy()
    assert g_refl.shape == (1, 3) and g_v.shape == (4, 3)
    assert np.isfinite(g_refl).all() and np.isfinite(g_v).all() and np.abs(g_refl).min() > 0 and np.abs(g_v).max() > 0
    # oracle: dloss/dtheta = <adj, d img/d theta> with adj = 2 w (img - target); same tables, same RNG offsets
    tb = sc.tables(0)
    o = _abi.make_opts(spp=8, sppe=4, sppse=4)
    adj = (2 * w * (img.t.detach() - target)).cpu().numpy()
    for c in range(3):
        t = torch.zeros_like(tb["texels"]); t[c] = 1.0
        _, d = oracle.render(tb, o, mode=1, tangents={"texels": t.cpu()})
        assert abs(g_refl[0, c] - float((adj * d).sum())) < 2e-3 * abs(g_refl[0, c])
    # one vertex coordinate via the oracle's forward mode through the torch table graph
    Pv = FloatD(0.)
    ek.set_requires_gradient(Pv)
    sc2 = _scene("cbox", spp=8, sppe=4, sppse=4)
    m2 = sc2.param_map["Mesh[0]"]
    d = torch.zeros(4, 3, device=Pv.t.device); d[2, 0] = 1.0
    m2.vertex_positions = Vector3fD._wrap(m2.vertex_positions.t.detach() + d * Pv.t)
    sc2.configure()
    tb2 = sc2.tables(0)
    _, dd = oracle.render(tb2, o, mode=1, tangents=tangents_wrt(tb2, Pv))
    ref = float((adj * dd).sum())
    assert abs(g_v[2, 0] - ref) < 5e-3 * max(abs(ref), 1e-3), (g_v[2, 0], ref)


def test_guided_secondary_edges_and_path_tracer():
    sc = _scene("cbox_occluder", spp=4, sppe=0, sppse=16)
    integ = psdr_cuda.DirectIntegrator(0, 2)
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.rotate(Vector3fD([0., 0., 1.]), P))
    sc.configure()
    w = integ.preprocess_secondary_edges(sc, 0, np.array([200, 4, 4, 2]), 4)
    assert w.m_distrb.m_sum > 0
    img = integ.renderD(sc, 0)
    ek.forward(P)
    g = ek.gradient(img).numpy()
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    pt = psdr_cuda.PathTracer(max_depth=3)
    sc.opts.sppse = 0
    sc.configure()
    a = pt.renderC(sc).numpy()
    assert np.isfinite(a).all() and a.mean() > 0.1
    f = psdr_cuda.FieldExtractionIntegrator("silhouette").renderC(sc).numpy()
    assert set(np.unique(f)).issubset({0.0, 1.0}) or f.max() <= 1.0
