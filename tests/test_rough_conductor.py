"""RoughConductor / GGX (reference src/bsdf/roughconductor.cpp, ggx.cpp) beyond the isotropic fixtures:
anisotropic roughness (alpha_u != alpha_v), estimator identities that pin sample / pdf / eval against each
other, an energy bound, and product-vs-oracle parity of the anisotropic code path."""
import numpy as np
import pytest
import torch

import oracle
import psdr_cuda
from helpers import FloatD, host_render, host_render_rev, dot_tables, random_tangents, rel_l2
from psdr_cuda import _abi
from psdr_cuda.fixtures import scene_path
from psdr_cuda.scene import look_at


def metal_floor(alpha_u, alpha_v, env_value=None, res=12, spp=64, k=(4.8, 3.1, 2.1), eta=(0.16, 0.12, 0.14)):
    """a smooth-shaded rough-conductor floor under the Cornell-box light (or a constant environment)"""
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
    cam = psdr_cuda.PerspectiveCamera(20.0, 0.1, 1e4)
    cam.to_world = look_at([0, 250, 420], [0, 0, 50], [0, 1, 0])
    sc.add_sensor(cam)
    b = psdr_cuda.RoughConductor(alpha_u, eta, k)
    b.alpha_v.data = FloatD(float(alpha_v))
    b.m_anisotropic = alpha_u != alpha_v
    b.id = "metal"
    sc.add_bsdf(b)
    black = psdr_cuda.Diffuse([0.0, 0.0, 0.0]); black.id = "black"
    sc.add_bsdf(black)
    m = psdr_cuda.Mesh()
    m.load(scene_path("cbox").replace("scenes/cbox.xml", "objects/cbox/floor.obj"))
    m.use_face_normals = False                     # anisotropic BSDFs need smooth shading frames (mesh.cpp:219)
    sc.add_mesh(m, b)
    if env_value is None:
        light = psdr_cuda.Mesh()
        light.use_face_normals = True
        # 80 x 80 quad at y = 150 facing down (normal -y)
        light.set_geometry(np.array([[-40, 150, 10], [40, 150, 10], [40, 150, 90], [-40, 150, 90]], np.float32),
                           np.array([[0, 1, 2], [0, 2, 3]], np.int32))
        sc.add_mesh(light, black, emitter_radiance=[20.0, 20.0, 20.0])
    else:
        env = psdr_cuda.EnvironmentMap()
        env.radiance = psdr_cuda.Bitmap3fD(2, 2, torch.tensor([env_value] * 4, dtype=torch.float32))
        sc.add_environment_map(env)
    sc.finalize()
    sc.configure()
    return sc


@pytest.mark.parametrize("au,av", [(0.2, 0.2), (0.1, 0.4), (0.35, 0.08)])
def test_bsdf_light_and_mis_estimators_agree(au, av):
    """E[f/pdf_bsdf over BSDF samples] == E[f G / pdf_light over light samples] == MIS: any mismatch between
    GGX sample(), pdf() and eval() (e.g. alpha_u / alpha_v swapped in one of them) breaks the equality"""
    tb = metal_floor(au, av, spp=1024).tables(0)
    means = [oracle.render(tb, _abi.make_opts(spp=1024, hide_emitters=True, **kw)).astype(np.float64).mean(0) for kw in
             (dict(bsdf_samples=1, light_samples=0), dict(bsdf_samples=0, light_samples=1), dict(bsdf_samples=1, light_samples=1))]
    assert means[0].min() > 1e-3
    for m in means[1:]:
        assert np.abs(m / means[0] - 1).max() < 0.06, (means, au, av)


def test_anisotropy_direction_matters_and_is_symmetric_under_swap_with_rotated_frame():
    """alpha_u != alpha_v changes the image; swapping them is NOT the same image (the frame is fixed)"""
    a = oracle.render(metal_floor(0.1, 0.4, spp=256).tables(0), _abi.make_opts(spp=256, hide_emitters=True))
    b = oracle.render(metal_floor(0.4, 0.1, spp=256).tables(0), _abi.make_opts(spp=256, hide_emitters=True))
    c = oracle.render(metal_floor(0.2, 0.2, spp=256).tables(0), _abi.make_opts(spp=256, hide_emitters=True))
    assert rel_l2(a, b) > 0.05 and rel_l2(a, c) > 0.05


@pytest.mark.parametrize("au,av", [(0.05, 0.05), (0.3, 0.3), (0.1, 0.4)])
def test_energy_bound_under_a_constant_environment(au, av):
    """a conductor never reflects more than it receives: under a constant environment L the reflected radiance
    is <= L (single scattering loses the multiply-scattered energy, more so when rough), and close to the
    Fresnel reflectance when smooth"""
    L = 2.0
    tb = metal_floor(au, av, env_value=[L, L, L], spp=256, eta=(0.2, 0.2, 0.2), k=(6.0, 6.0, 6.0)).tables(0)
    img = oracle.render(tb, _abi.make_opts(spp=256, bsdf_samples=1, light_samples=0)).astype(np.float64)
    floor = img[img[:, 0] < 0.999 * L]           # pixels that see the floor (the others see the environment directly)
    assert floor.shape[0] > 20
    assert floor.max() <= L * 1.0 + 1e-4 and floor.mean() > 0.5 * L
    if au == av == 0.05:
        assert floor.mean() > 0.85 * L            # F(eta = 0.2, k = 6) ~ 0.97, little masking at alpha = 0.05


@pytest.mark.parametrize("kind", ["direct11", "path3"])
def test_anisotropic_product_code_matches_oracle(kind):
    kw = dict(bsdf_samples=1, light_samples=1) if kind == "direct11" else dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    tb = metal_floor(0.1, 0.4, res=16, spp=8).tables(0)
    assert tb["material_mask"] == 3
    o = _abi.make_opts(spp=8, **kw)
    assert rel_l2(host_render(tb, o), oracle.render(tb, o)) < 2e-5
    # alpha_u / alpha_v / eta / k / reflectance texels: forward tangents vs oracle, reverse vs forward
    tan = random_tangents(tb, ["texels"], seed=2)
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = host_render(tb, o, mode=1, tangents=tan)
    assert rel_l2(dimg, ref_d) < 2e-3 and np.abs(ref_d).max() > 0
    adj = np.random.default_rng(3).random((16 * 16, 3)).astype(np.float32)
    _, grads = host_render_rev(tb, o, adj, want=["texels"])
    lhs, rhs = float((adj.astype(np.float64) * dimg).sum()), dot_tables(grads, tan)
    assert abs(lhs - rhs) < 2e-4 * np.abs(adj * dimg).sum()
