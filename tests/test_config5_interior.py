"""BASELINE config 5 (multi-material interior, ~50 k triangles, GGX materials; no such scene ships with
the reference, so a seeded generator stands in): parity, roughness derivative and vertex gradients."""
import numpy as np
import pytest
import torch

import oracle
from helpers import isolated_pixels_unbiased, GpuScene, dot_tables, host_render, load_scene, random_tangents, rel_l2
from psdr_cuda import _abi
from psdr_cuda.fixtures import make_interior_scene


def test_generator_is_seeded_and_sized():
    sc = make_interior_scene(seed=1, n_objects=2, res=16, spp=1)
    sc.configure()
    tb = sc.tables(0)
    assert tb["num_tris"] == 12 + 2 * 4968 and tb["num_bsdfs"] == 8
    sc2 = make_interior_scene(seed=1, n_objects=2, res=16, spp=1)
    sc2.configure()
    assert torch.equal(sc2.tables(0)["tri_info"], tb["tri_info"])
    o = _abi.make_opts(spp=1, bsdf_samples=1, light_samples=1)
    a, b = host_render(tb, o), oracle.render(tb, o)
    bad = (np.abs(a - b).max(1) > 1e-3 * (1 + np.abs(b).max(1))).mean()
    assert bad < 0.02


@pytest.mark.gpu
def test_config5_full_scene_gpu():
    sc = make_interior_scene(seed=0, n_objects=10, res=128, spp=8)
    sc.configure()
    tb = sc.tables(0)
    assert tb["num_tris"] == 12 + 10 * 4968          # ~50 k triangles
    g = GpuScene(tb)
    o = _abi.make_opts(spp=8, bsdf_samples=1, light_samples=1)
    img, ref = g.render_c(o), oracle.render(tb, o)
    badm = np.abs(img - ref).max(1) > 1e-3 * (1 + np.abs(ref).max(1))
    bad = badm.mean()
    print("C5 renderC: rel-L2 %.2e, pixels off by > 1e-3: %.2e" % (rel_l2(img, ref), bad))
    assert bad < 2e-3 and rel_l2(img, ref) < 1e-3, (bad, rel_l2(img, ref))
    isolated_pixels_unbiased(img, ref, badm, "C5 renderC")                      # the excluded pixels' signed errors cancel (helpers.py)
    # roughness derivative: every alpha texel moves together (material_roughness, differential.py:28-31)
    rec = tb["bsdf_rec"].cpu().numpy()
    t = torch.zeros_like(tb["texels"])
    for r in rec[rec[:, 0] == _abi.BSDF_ROUGHCONDUCTOR]:
        t[int(r[1 + 3 * _abi.SLOT_ALPHA_U])] = 1.0; t[int(r[1 + 3 * _abi.SLOT_ALPHA_V])] = 1.0
    _, dref = oracle.render(tb, o, mode=1, tangents={"texels": t})
    _, dimg = g.render_d_fwd(o, [{"texels": t}])
    badm = np.abs(dimg[0] - dref).max(1) > 2e-3 * (1 + np.abs(dref).max(1))
    bad = badm.mean()
    print("C5 roughness derivative: rel-L2 %.2e, pixels off by > 2e-3: %.2e" % (rel_l2(dimg[0], dref), bad))
    assert np.abs(dref).max() > 0 and bad < 0.02, bad
    isolated_pixels_unbiased(dimg[0], dref, badm, "C5 roughness derivative", bias_bound=2e-2)
    # vertex (triangle-table) + roughness gradients in reverse mode == forward mode
    adj = np.random.default_rng(0).random((128 * 128, 3)).astype(np.float32)
    tan = random_tangents(tb, ["tri_info", "texels"], seed=2)
    _, dfw = g.render_d_fwd(o, [tan])
    _, grads = g.render_d_rev(o, adj, want=["tri_info", "texels"], with_image=False)
    lhs, rhs = float((adj.astype(np.float64) * dfw[0]).sum()), dot_tables(grads, tan)
    assert abs(lhs - rhs) < 5e-3 * np.abs(adj * dfw[0]).sum(), (lhs, rhs)


@pytest.mark.gpu
def test_config5_path_tracer_and_geometry_gradients_against_the_oracle():
    """BASELINE config 5 against the ORACLE in every mode the config names: (i) PathTracer(3) renderC and its roughness
    derivative (the traced wavefront: dense trace kernel + bounce stages), (ii) forward-mode geometry duals of a rigid translation of one
    object with all three terms (DirectIntegrator, sppe = sppse > 0) against oracle.render(mode=1), (iii) the reverse-mode gradient of the
    same samples projected on that tangent against the same number.  96 x 96, spp 8: the oracle finishes in seconds on the box's cores."""
    import enoki as ek
    from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
    from helpers import tangents_wrt
    res, spp = 96, 8
    sc = make_interior_scene(seed=0, n_objects=10, res=res, spp=spp)
    sc.opts.sppe = sc.opts.sppse = spp
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.m_meshes[8].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.4, -0.3]) * P))          # one of the ten objects (meshes 0-5: light + walls)
    sc.configure()
    tb = sc.tables(0)
    assert tb["num_tris"] == 12 + 10 * 4968
    g = GpuScene(tb)

    def flips(a, b, tol):
        return float((np.abs(a - b).max(1) > tol * (1 + np.abs(b).max(1))).mean())
    # ---- (i) PathTracer(3)
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp)
    img, ref = g.render_c(o), oracle.render(tb, o)
    print("C5 PathTracer(3) renderC: rel-L2 %.2e, pixels off by > 1e-3: %.2e" % (rel_l2(img, ref), flips(img, ref, 1e-3)))
    assert flips(img, ref, 1e-3) < 5e-3 and rel_l2(img, ref) < 2e-3
    rec = tb["bsdf_rec"].cpu().numpy()
    t = torch.zeros_like(tb["texels"])
    for r in rec[rec[:, 0] == _abi.BSDF_ROUGHCONDUCTOR]:
        t[int(r[1 + 3 * _abi.SLOT_ALPHA_U])] = 1.0; t[int(r[1 + 3 * _abi.SLOT_ALPHA_V])] = 1.0
    _, dref = oracle.render(tb, o, mode=1, tangents={"texels": t})
    _, dimg = g.render_d_fwd(o, [{"texels": t}])
    print("C5 PathTracer(3) roughness derivative: rel-L2 %.2e, pixels off by > 2e-3: %.2e" % (rel_l2(dimg[0], dref), flips(dimg[0], dref, 2e-3)))
    assert np.abs(dref).max() > 0 and flips(dimg[0], dref, 2e-3) < 0.01 and rel_l2(dimg[0], dref) < 5e-3          # measured 3.7e-3 / 1.2e-3
    isolated_pixels_unbiased(dimg[0], dref, np.abs(dimg[0] - dref).max(1) > 2e-3 * (1.0 + np.abs(dref).max(1)), "C5 PathTracer(3) roughness derivative", bias_bound=2e-2)
    # ---- (ii) geometry duals, three terms, forward mode
    od = _abi.make_opts(spp=spp, sppe=spp, sppse=spp, bsdf_samples=1, light_samples=1)
    tan = tangents_wrt(tb, P)
    assert tan["tri_info"] is not None and tan["sec_edge"] is not None and tan["prim_edge"] is not None
    rimg, rd = oracle.render(tb, od, mode=1, tangents=tan)
    fimg, fd = g.render_d_fwd(od, [tan])
    adj = (0.5 + np.random.default_rng(3).random((res * res, 3))).astype(np.float64)
    b = float((adj * rd).sum())
    scale = float(np.abs(adj * rd).sum())
    a_fwd = float((adj * fd[0]).sum())
    print("C5 geometry (translation of one object, 3 terms): <A, dI> oracle %+.6e  HIP forward %+.6e  (sum|A dI| %.3e); derivative image rel-L2 %.2e, pixels off by > 2e-3: %.2e"
          % (b, a_fwd, scale, rel_l2(fd[0], rd), flips(fd[0], rd, 2e-3)))
    assert rel_l2(fimg, rimg) < 1e-3
    assert abs(a_fwd - b) < 2e-4 * scale, (a_fwd, b, scale)          # measured 2.6e-6
    assert flips(fd[0], rd, 2e-3) < 5e-3 and rel_l2(fd[0], rd) < 1e-3          # measured 0 / 1.8e-5
    # ---- (ii b) the same tangent through the PathTracer: geometry duals of the traced wavefront (round 5: k_wfg_camera / k_wfg_bounce on the rough-conductor,
    # 4-wide-tree instances of flag set 6) against the oracle and against the fused kernel
    op = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp, rng_offset=(11, 0, 0))
    _, pd_ref = oracle.render(tb, op, mode=1, tangents=tan)
    _, pd_w = g.render_d_fwd(_abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp, rng_offset=(11, 0, 0), flags=_abi.FLAG_WAVEFRONT), [tan]); rays_w = g.counters()[0]
    _, pd_f = g.render_d_fwd(_abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=spp, rng_offset=(11, 0, 0), flags=_abi.FLAG_FUSED), [tan]); rays_f = g.counters()[0]
    bp = float((adj * pd_ref).sum()); sp = float(np.abs(adj * pd_ref).sum())
    print("C5 PathTracer(3) geometry duals: <A, dI> oracle %+.6e  wavefront %+.6e  fused %+.6e (sum|A dI| %.3e); pixels off by > 2e-3: wavefront %.2e fused %.2e; rays %d / %d"
          % (bp, float((adj * pd_w[0]).sum()), float((adj * pd_f[0]).sum()), sp, flips(pd_w[0], pd_ref, 2e-3), flips(pd_f[0], pd_ref, 2e-3), rays_w, rays_f))
    assert np.abs(pd_ref).max() > 0 and abs(rays_w - rays_f) <= 1e-4 * rays_f
    assert abs(float((adj * pd_w[0]).sum()) - bp) < 2e-3 * sp and flips(pd_w[0], pd_ref, 2e-3) < 1e-2
    assert flips(pd_w[0], pd_f[0], 2e-3) < 1e-2
    # ---- (iii) reverse mode of the same samples projected on the tangent
    _, grads = g.render_d_rev(od, adj.astype(np.float32), want=["tri_info", "sec_edge", "prim_edge"], with_image=False)
    a_rev = dot_tables(grads, {k: v for k, v in tan.items() if v is not None})
    print("C5 geometry: HIP reverse %+.6e" % a_rev)
    assert abs(a_rev - b) < 2e-4 * scale, (a_rev, b, scale)
