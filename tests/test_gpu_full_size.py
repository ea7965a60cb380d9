"""BASELINE.json's full-size configuration C2 (cbox 512x512, spp 64, PathTracer) on the GPU:
parity with the oracle (the GPU box has enough host cores to run it in seconds) plus
size-independent properties (linearity over sample shards, forward/reverse gradient identity).
The headline accuracy target -- gradient within 1e-3 relative of the reference -- is asserted here."""
import numpy as np
import pytest
import torch

import oracle
from helpers import isolated_pixels_unbiased, GpuScene, load_scene, rel_l2
from psdr_cuda import _abi

pytestmark = pytest.mark.gpu

RES, SPP = 512, 64


@pytest.fixture(scope="module")
def c2():
    sc, _ = load_scene("cbox", res=RES, spp=SPP)
    tb = sc.tables(0)
    return tb, GpuScene(tb)


def test_c2_render_c_and_albedo_gradient_match_oracle(c2):
    tb, g = c2
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=SPP)
    img = g.render_c(o)
    ref = oracle.render(tb, o)
    assert rel_l2(img, ref) < 1e-4, rel_l2(img, ref)
    assert abs(img.mean() - ref.mean()) < 1e-4 * ref.mean()
    # d image / d albedo(r,g,b) of BSDF[0] (white walls) in ONE K=3 pass; gradient of loss = sum(image)
    sets = []
    for c in range(3):
        t = torch.zeros_like(tb["texels"]); t[c] = 1.0
        sets.append({"texels": t})
    img_d, dimg = g.render_d_fwd(o, sets)
    grad_fwd = np.array([dimg[c].astype(np.float64).sum() for c in range(3)])
    grad_ref = np.zeros(3)
    for c in range(3):
        _, d = oracle.render(tb, o, mode=1, tangents=sets[c])
        grad_ref[c] = d.astype(np.float64).sum()
    rel = np.linalg.norm(grad_fwd - grad_ref) / np.linalg.norm(grad_ref)
    assert rel < 1e-3, (grad_fwd, grad_ref, rel)
    # reverse mode: same gradient through the adjoint kernels and the scatter-add
    adj = np.ones((RES * RES, 3), dtype=np.float32)
    _, grads = g.render_d_rev(o, adj, want=["texels"], with_image=False)
    rel_rev = np.linalg.norm(grads["texels"][:3] - grad_ref) / np.linalg.norm(grad_ref)
    assert rel_rev < 1e-3, (grads["texels"][:3], grad_ref, rel_rev)


def test_c2_shards_are_linear(c2):
    tb, g = c2
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=SPP)
    full = g.render_c(_abi.make_opts(**kw))
    parts = sum(g.render_c(_abi.make_opts(spp_range=(8 * k, 8 * k + 8), **kw)) for k in range(8))   # the 8-GPU partition
    assert rel_l2(parts, full) < 1e-5


def test_c2_direct_and_field_full_size(c2):
    tb, g = c2
    o = _abi.make_opts(bsdf_samples=1, light_samples=1, spp=SPP)
    assert rel_l2(g.render_c(o), oracle.render(tb, o)) < 1e-3
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["depth"], spp=4)
    assert rel_l2(g.render_c(o), oracle.render(tb, o)) < 1e-5


def test_c3_bunny_vertex_gradient_dot_product():
    """C3-style: bunny in the box, DirectIntegrator, all three terms, reverse gradients w.r.t. every
    triangle row vs forward mode for a rigid translation (512x512, spp 16)."""
    from helpers import tangents_wrt, dot_tables
    sc, P = load_scene("cbox_bunny", res=512, spp=16, sppe=16, sppse=16, translate=(1, (1.0, 0.3, 0.0)))
    tb = sc.tables(0)
    g = GpuScene(tb)
    o = _abi.make_opts(spp=16, sppe=16, sppse=16)
    tan = tangents_wrt(tb, P)
    adj = np.random.default_rng(3).random((512 * 512, 3)).astype(np.float32)
    img, dimg = g.render_d_fwd(o, [tan])
    _, grads = g.render_d_rev(o, adj, with_image=False)
    tan = {k: v for k, v in tan.items() if v is not None}
    lhs, rhs = float((adj.astype(np.float64) * dimg[0]).sum()), dot_tables(grads, tan)
    scale = float(np.abs(adj.astype(np.float64) * dimg[0]).sum())
    assert abs(lhs - rhs) < 5e-3 * scale, (lhs, rhs, scale)


def test_c4_one_gpu_share_of_1024x1024_spp512():
    """BASELINE config C4: cbox_bunny 1024x1024, global spp 512 sharded over 8 GPUs = 64 spp per GPU
    (67 108 864 slots, < INT_MAX, integrator.cpp:74).  Size-independent properties of one GPU's share:
    the share is the sum of its sub-shards (the all-reduce identity), it is normalised by the GLOBAL spp,
    its stream indices are the global ones, and it agrees statistically with an independent low-res render."""
    sc, _ = load_scene("cbox_bunny", res=1024, spp=512)
    tb = sc.tables(0)
    g = GpuScene(tb)
    kw = dict(bsdf_samples=1, light_samples=1, spp=512)
    share = g.render_c(_abi.make_opts(spp_range=(64, 128), **kw))                    # rank 1 of 8
    assert g.counters()[1] == 1024 * 1024 * 64
    assert np.isfinite(share).all()
    halves = g.render_c(_abi.make_opts(spp_range=(64, 96), **kw)) + g.render_c(_abi.make_opts(spp_range=(96, 128), **kw))
    assert rel_l2(halves, share) < 1e-5
    # normalised by the global spp: 8 such shares make the image; one share carries 1/8 of the energy
    sc2, _ = load_scene("cbox_bunny", res=64, spp=64)
    low = oracle.render(sc2.tables(0), _abi.make_opts(bsdf_samples=1, light_samples=1, spp=64))
    assert abs(8.0 * share.mean() / low.mean() - 1.0) < 0.02
    # the same share at a 16x16 crop-sized problem equals the oracle sample for sample (global stream ids)
    sc3, _ = load_scene("cbox_bunny", res=32, spp=512)
    tb3 = sc3.tables(0)
    o3 = _abi.make_opts(spp_range=(64, 128), **kw)
    a, b = GpuScene(tb3).render_c(o3), oracle.render(tb3, o3)
    bad = np.abs(a - b).max(1) > 1e-3 * (1 + np.abs(b).max(1))                       # pixels holding a sample that resolved a tie the other way
    assert bad.mean() < 0.005 and rel_l2(a[~bad], b[~bad]) < 1e-3, (bad.mean(), rel_l2(a[~bad], b[~bad]))
    isolated_pixels_unbiased(a, b, bad, "full-size tree scene")                 # the excluded pixels' signed errors cancel (helpers.py)


def test_c4_share_path_tracer_geometry_duals_at_full_size():
    """BASELINE config 4's PathTracer with the reference harness' AD mode (renderD + enoki.forward w.r.t. a translation of the bunny,
    examples/run_test.py:126-129) on one GPU's share at its stated size: 1024 x 1024, 64 of 512 spp = 67 M slots through the traced wavefront with
    dual-number stages (round 5).  Size-independent properties: the share's derivative image is the sum of its sub-shards' (the all-reduce identity; the
    second sub-shard crosses nothing the first one touched), it equals the fused kernel's on the same samples up to isolated pixels, it is finite and
    non-zero exactly where the bunny or its shadow / reflections are, and the primal image riding along equals renderC's."""
    from helpers import tangents_wrt
    sc, P = load_scene("cbox_bunny", res=1024, spp=512, sppe=0, sppse=0, translate=(1, (1.0, 0.5, 0.25)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    g = GpuScene(tb)
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=512)
    img, d = g.render_d_fwd(_abi.make_opts(spp_range=(64, 128), **kw), [tan])
    assert g.counters()[1] == 1024 * 1024 * 64
    assert np.isfinite(d[0]).all() and np.abs(d[0]).max() > 0
    ref_img = g.render_c(_abi.make_opts(spp_range=(64, 128), **kw))
    assert rel_l2(img, ref_img) < 3e-4                               # separately compiled fp32 kernels (float stages / dual stages): isolated samples, measured 3.3e-5
    ia, da = g.render_d_fwd(_abi.make_opts(spp_range=(64, 96), **kw), [tan])
    ib, db = g.render_d_fwd(_abi.make_opts(spp_range=(96, 128), **kw), [tan])
    assert rel_l2(da[0] + db[0], d[0]) < 1e-5 and rel_l2(ia + ib, img) < 1e-5
    _, df = g.render_d_fwd(_abi.make_opts(spp_range=(64, 128), flags=_abi.FLAG_FUSED, **kw), [tan])
    bad = np.abs(d[0] - df[0]).max(1) > 1e-3 * (np.abs(df[0]).max(1) + 1e-3 * np.abs(df[0]).max())
    print("C4 share, PathTracer(3) geometry duals: wavefront vs fused derivative image rel-L2 %.2e, pixels apart %d of %d" % (rel_l2(d[0], df[0]), bad.sum(), bad.size))
    assert bad.mean() < 2e-3 and rel_l2(d[0][~bad], df[0][~bad]) < 1e-3
    isolated_pixels_unbiased(d[0], df[0], bad, "C4 share geometry duals", bias_bound=5e-3)
    assert abs(float(d[0].astype(np.float64).sum()) - float(df[0].astype(np.float64).sum())) < 1e-3 * float(np.abs(df[0].astype(np.float64)).sum())


@pytest.mark.parametrize("res,spp", [(128, 1), (128, 4)])
def test_c1_literal_size_direct_render_c(res, spp):
    """BASELINE config C1 at its literal size: cbox 128x128, spp = 1, DirectIntegrator.renderC (the reference's CPU-runnable plumbing case)."""
    sc, _ = load_scene("cbox", res=res, spp=spp)
    tb = sc.tables(0)
    o = _abi.make_opts(bsdf_samples=1, light_samples=1, spp=spp)
    img, ref = GpuScene(tb).render_c(o), oracle.render(tb, o)
    assert img.shape == (res * res, 3) and rel_l2(img, ref) < 1e-4, rel_l2(img, ref)
