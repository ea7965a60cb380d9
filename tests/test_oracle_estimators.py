"""Pins the oracle by closed forms, estimator identities and the reference's own validation method
(AD image == central finite difference, examples/run_test.py:44-231).  `parity unpinned` caveat:
the reference ships no golden outputs."""
import numpy as np
import torch

import oracle
from helpers import load_scene, rel_l2, tangents_wrt
from psdr_cuda import _abi


def test_mis_bsdf_only_and_light_only_agree():
    """config.py:46-78 (cbox_MIS / cbox_bs / cbox_es): three estimators, one integral."""
    sc, _ = load_scene("cbox", res=16, spp=512)
    tb = sc.tables(0)
    imgs = [oracle.render(tb, _abi.make_opts(bsdf_samples=b, light_samples=l, spp=512)) for b, l in ((1, 1), (2, 0), (0, 2))]
    m = [i.mean() for i in imgs]
    assert abs(m[0] - m[1]) < 0.02 * m[0] and abs(m[0] - m[2]) < 0.02 * m[0]
    assert rel_l2(imgs[2], imgs[0]) < 0.1


def test_pathtracer_depth1_is_direct_11_sample_for_sample():
    sc, _ = load_scene("cbox_rough", res=24, spp=4)
    tb = sc.tables(0)
    a = oracle.render(tb, _abi.make_opts(bsdf_samples=1, light_samples=1, spp=4))
    b = oracle.render(tb, _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=1, spp=4))
    assert np.array_equal(a, b)


def test_direct_light_on_a_diffuse_floor_closed_form():
    """Irradiance at floor points under the 80x80 Lambertian quad light, by brute-force quadrature of
    E = Le * int cos(t) cos(t') / r^2 dA; radiance = rho/pi * E.  Light-sampling-only estimator."""
    sc, _ = load_scene("cbox", res=48, spp=256)
    tb = sc.tables(0)
    pos = oracle.render(tb, _abi.make_opts(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["position"], spp=32))
    nrm = oracle.render(tb, _abi.make_opts(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["geoNormal"], spp=32))
    img = oracle.render(tb, _abi.make_opts(bsdf_samples=0, light_samples=1, spp=256))
    # pixels whose whole footprint lies on the floor (mean geometric normal == +y)
    floor = np.nonzero((nrm[:, 1] > 0.9999) & (np.abs(pos[:, 1]) < 1e-3))[0]
    assert floor.size > 50
    g = (np.arange(200) + 0.5) / 200 * 80 - 40
    lx, lz = np.meshgrid(50 + g, g, indexing="ij")
    errs = []
    for i in floor[:: max(1, floor.size // 16)]:
        p = pos[i]                         # mean position of the footprint (radiance is smooth across it)
        dx, dy, dz = lx - p[0], 190.0 - p[1], lz - p[2]
        r2 = dx * dx + dy * dy + dz * dz
        E = (dy * dy / (r2 * r2)).sum() * (80.0 / 200) ** 2      # cos*cos'/r^2, both normals along y
        expect = 0.95 / np.pi * E * np.array([20.0, 20.0, 8.0])
        errs.append(np.abs(img[i] - expect) / expect)
    assert np.median(errs) < 0.02 and np.max(errs) < 0.08, (np.median(errs), np.max(errs))


def test_camera_sample_direct_inverts_sample_primary_ray():
    """world_to_sample(position seen through pixel p) falls inside pixel p."""
    sc, _ = load_scene("cbox", res=16, spp=1)
    tb = sc.tables(0)
    pos = oracle.render(tb, _abi.make_opts(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["position"], spp=1))
    w2s = tb["cam"][32:48].reshape(4, 4).numpy().astype(np.float64)
    hit = np.abs(pos).sum(1) > 0
    q = np.concatenate([pos, np.ones((pos.shape[0], 1))], 1) @ w2s.T
    q = q[:, :2] / q[:, 3:4]
    pix = np.floor(q[:, 1] * 16) * 16 + np.floor(q[:, 0] * 16)
    assert hit.sum() > 100 and np.array_equal(pix[hit], np.arange(256)[hit])


def test_albedo_derivative_ad_equals_fd_with_common_random_numbers():
    sc, _ = load_scene("cbox", res=16, spp=8)
    tb = sc.tables(0)
    for kw in (dict(bsdf_samples=1, light_samples=1), dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)):
        o = _abi.make_opts(spp=8, **kw)
        dt = torch.zeros_like(tb["texels"]); dt[0] = 1.0
        _, dimg = oracle.render(tb, o, mode=1, tangents={"texels": dt}, precision=1)
        eps = 1e-3
        imgs = []
        for s in (+1, -1):
            t2 = dict(tb); t2["texels"] = tb["texels"].clone(); t2["texels"][0] += s * eps
            imgs.append(oracle.render(t2, o, precision=1))
        fd = (imgs[0] - imgs[1]) / (2 * eps)
        assert rel_l2(dimg, fd) < 1e-3


def test_roughness_derivative_ad_equals_fd():
    """material_roughness perturbation of the reference harness (utils/differential.py:28-31)."""
    sc, _ = load_scene("cbox_rough", res=16, spp=16)
    tb = sc.tables(0)
    rec = tb["bsdf_rec"].numpy()
    rc = int(np.nonzero(rec[:, 0] == _abi.BSDF_ROUGHCONDUCTOR)[0][0])
    au, av = int(rec[rc, 1 + 3 * _abi.SLOT_ALPHA_U]), int(rec[rc, 1 + 3 * _abi.SLOT_ALPHA_V])
    o = _abi.make_opts(spp=16, bsdf_samples=1, light_samples=1)
    dt = torch.zeros_like(tb["texels"]); dt[au] = 1.0; dt[av] = 1.0
    _, dimg = oracle.render(tb, o, mode=1, tangents={"texels": dt}, precision=1)
    eps = 1e-4
    imgs = []
    for s in (+1, -1):
        t2 = dict(tb); t2["texels"] = tb["texels"].clone(); t2["texels"][au] += s * eps; t2["texels"][av] += s * eps
        imgs.append(oracle.render(t2, o, precision=1))
    fd = (imgs[0] - imgs[1]) / (2 * eps)
    # light-sampled terms are smooth in alpha; BSDF-sampled directions move with alpha (the hit
    # triangle can change), so compare the image means and the bulk of the pixels
    assert abs(dimg.sum() - fd.sum()) < 0.05 * abs(fd.sum())
    assert np.median(np.abs(dimg - fd).max(1) / (np.abs(fd).max(1) + 1e-3)) < 0.05


def test_geometry_derivative_interior_plus_edges_matches_fd():
    """Moving an occluder: interior + primary-edge + secondary-edge terms vs central FD (the
    reference's cbox_MIS style check).  Monte-Carlo, so the tolerance is statistical."""
    res = 24     # coarser pixels make interior and primary-edge terms cancel almost exactly (noise-dominated)
    sc, P = load_scene("cbox_occluder", res=res, spp=4096, sppe=4096, sppse=4096, translate=(1, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=4096, sppe=4096, sppse=4096)
    _, ad = oracle.render(tb, o, mode=1, tangents=tan)
    _, ad_no_sec = oracle.render(tb, _abi.make_opts(spp=4096, sppe=4096, sppse=0), mode=1, tangents=tan)
    eps, M = 1.0, 65536
    imgs = []
    for s in (+1, -1):
        import enoki as ek
        from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
        sc2, _ = load_scene("cbox_occluder", res=res, spp=M)
        sc2.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.5, 0.0]) * FloatD(s * eps)))
        sc2.configure()
        imgs.append(oracle.render(sc2.tables(0), _abi.make_opts(spp=M)))
    fd = (imgs[0] - imgs[1]) / (2 * eps)
    e_all, e_nosec = rel_l2(ad, fd), rel_l2(ad_no_sec, fd)
    assert e_all < 0.1, e_all
    assert e_nosec > e_all + 0.05     # dropping the secondary-edge term must hurt


def test_fp32_and_fp64_oracle_agree():
    sc, _ = load_scene("cbox", res=16, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=16)
    assert rel_l2(oracle.render(tb, o, precision=0), oracle.render(tb, o, precision=1)) < 1e-4
