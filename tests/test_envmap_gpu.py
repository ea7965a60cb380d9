"""EnvironmentMap on the GPU (run with `-m gpu`): the HIP path through the C ABI and through the
Python surface against the CPU oracle on the same seeded inputs (CPU half: tests/test_envmap.py)."""
import numpy as np
import pytest
import torch

import enoki as ek
import oracle
import psdr_cuda
from helpers import FloatD, GpuScene, Matrix4fD, Vector3fD, load_scene, rel_l2, tangents_wrt
from psdr_cuda import _abi
from test_envmap import KINDS, constant_env_floor, env_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["bunny_env", "cbox_env"])
@pytest.mark.parametrize("kind", list(KINDS))
def test_render_c_matches_oracle(scene, kind):
    sc, _ = load_scene(scene, res=48, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=16, rng_offset=(7, 0, 0), **KINDS[kind])
    ref = oracle.render(tb, o)
    img = GpuScene(tb).render_c(o)
    assert np.isfinite(img).all()
    bad = (np.abs(img - ref).max(1) > 1e-3 * (1 + np.abs(ref).max(1))).mean()
    assert bad < 0.005 and rel_l2(img, ref) < 5e-3


def test_wavefront_path_tracer_equals_fused_under_environment_light():
    sc, _ = load_scene("bunny_env", res=48, spp=8)
    tb = sc.tables(0)
    g = GpuScene(tb)
    a = g.render_c(_abi.make_opts(spp=8, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=_abi.FLAG_FUSED))
    b = g.render_c(_abi.make_opts(spp=8, integrator=_abi.INTEGRATOR_PATH, max_depth=3, flags=_abi.FLAG_WAVEFRONT))
    assert rel_l2(a, b) < 1e-5


def test_constant_environment_closed_form_on_gpu():
    L, scale, albedo = [2.0, 1.0, 0.5], 1.5, [0.8, 0.5, 0.3]
    tb = constant_env_floor(L, scale, albedo, res=32, spp=64).tables(0)
    want = np.array(L) * scale * np.array(albedo)
    g = GpuScene(tb)
    img = g.render_c(_abi.make_opts(spp=64, bsdf_samples=1, light_samples=0))
    assert np.abs(img / want - 1).max() < 2e-5
    img = g.render_c(_abi.make_opts(spp=64, bsdf_samples=1, light_samples=1))
    assert np.abs(img.mean(0) / want - 1).max() < 0.01


def test_forward_env_parameters_match_oracle():
    sc, P = env_scene(0.0, True, res=48, spp=8)
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    t_scale = {"env_f": torch.zeros_like(tb["env_f"])}
    t_scale["env_f"][18] = 1.0
    gen = torch.Generator().manual_seed(3)
    t_tex = {"texels": torch.rand(tb["texels"].shape, generator=gen)}
    g = GpuScene(tb)
    for kind in ("direct11", "path3"):
        o = _abi.make_opts(spp=8, **KINDS[kind])
        img, dimgs = g.render_d_fwd(o, [tan, t_scale, t_tex])           # K = 3: one pass
        for k, t in enumerate((tan, t_scale, t_tex)):
            ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=t)
            bad = (np.abs(dimgs[k] - ref_d).max(1) > 1e-3 * (1 + np.abs(ref_d).max(1))).mean()
            assert bad < 0.02 and rel_l2(dimgs[k], ref_d) < 3e-2 and np.abs(ref_d).max() > 0, (kind, k)


def test_forward_geometry_with_edges_against_environment_light():
    sc, P = load_scene("cbox_env", res=48, spp=8, sppe=8, sppse=8, translate=(1, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=8, sppe=8, sppse=8, rng_offset=(0, 5, 9), bsdf_samples=1, light_samples=1)
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimgs = GpuScene(tb).render_d_fwd(o, [tan])
    bad = (np.abs(dimgs[0] - ref_d).max(1) > 1e-3 * (1 + np.abs(ref_d).max(1))).mean()
    assert rel_l2(img, ref_img) < 1e-3 and bad < 0.02 and np.abs(ref_d).max() > 0


def test_python_surface_envmap_rotate_forward():
    """examples/run_test.py:114-129 with AD type `envmap_rotate` (examples/config.py:128-145)"""
    sc = psdr_cuda.Scene()
    from psdr_cuda.fixtures import scene_path
    sc.load_file(scene_path("bunny_env"), False)
    sc.opts.width = sc.opts.height = 32
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 8, 0, 0, 0
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.param_map["Emitter[0]"].set_transform(Matrix4fD.rotate(Vector3fD([0., 1., 0.]), P))
    sc.configure()
    integ = psdr_cuda.DirectIntegrator(2, 2)
    img = integ.renderD(sc, 0)
    ek.forward(P, free_graph=True)
    g = ek.gradient(img).numpy()
    tb = sc.tables(0)
    ref_img, ref_d = oracle.render(tb, _abi.make_opts(spp=8, bsdf_samples=2, light_samples=2), mode=1, tangents=tangents_wrt(tb, P))
    assert np.isfinite(g).all() and np.abs(ref_d).max() > 0
    bad = (np.abs(g - ref_d).max(1) > 1e-3 * (1 + np.abs(ref_d).max(1))).mean()
    assert bad < 0.02 and rel_l2(img.numpy(), ref_img) < 5e-3


def test_python_surface_backward_env_radiance_scale_and_rotation():
    """docs/inverse_diff_render.rst pattern under environment lighting: loss.backward() delivers gradients
    of the map's texels, its scale and the rotation angle; each equals <adjoint, forward-mode derivative>."""
    from psdr_cuda.fixtures import scene_path
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("bunny_env"), False)
    sc.opts.width = sc.opts.height = 32
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 8, 0, 0, 0
    env = sc.param_map["Emitter[0]"]
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    ek.set_requires_gradient(env.radiance.data)
    ek.set_requires_gradient(env.scale)
    env.set_transform(Matrix4fD.rotate(Vector3fD([0., 1., 0.]), P))
    sc.configure()
    integ = psdr_cuda.DirectIntegrator(1, 1)
    tb = sc.tables(0)
    tan_P = tangents_wrt(tb, P)             # before backward() frees the table graph
    img = integ.renderD(sc, 0)
    w = torch.linspace(0.5, 1.5, 32 * 32 * 3, device="cuda").reshape(-1, 3)
    loss = (img.t * w).sum()
    loss.backward()
    g_tex, g_scale, g_P = ek.gradient(env.radiance.data).numpy(), ek.gradient(env.scale).numpy(), ek.gradient(P).numpy()
    assert g_tex.shape == (64 * 32, 3) and np.isfinite(g_tex).all() and np.abs(g_tex).max() > 0
    o = _abi.make_opts(spp=8, bsdf_samples=1, light_samples=1)
    g = GpuScene(tb)
    wn = w.cpu().numpy().astype(np.float64)
    t_scale = {"env_f": torch.zeros_like(tb["env_f"])}
    t_scale["env_f"][18] = 1.0
    _, d = g.render_d_fwd(o, [t_scale, tan_P, t_scale])
    assert abs(float(g_scale.reshape(-1)[0]) / (wn * d[0]).sum() - 1) < 1e-3
    assert abs(float(g_P.reshape(-1)[0]) / (wn * d[1]).sum() - 1) < 5e-3
    # image = scale * (linear in texels): <g_tex, texels> == <w, image>
    lhs = float((g_tex.astype(np.float64) * env.radiance.data.numpy()).sum())
    assert abs(lhs / float((wn * img.numpy()).sum()) - 1) < 1e-3


def test_reference_ballroom_map_full_size_cells():
    """the reference's 1024x512 PIZ panorama: 2046 x 1022 luminance cells (21-step binary search per light
    sample), rough conductor alpha = 0.05"""
    sc, _ = load_scene("bunny_env_ballroom", res=48, spp=8)
    tb = sc.tables(0)
    assert tb["env_cmf"].numel() == 2046 * 1022
    g = GpuScene(tb)
    for kind in ("direct11", "direct02", "path3"):
        o = _abi.make_opts(spp=8, rng_offset=(1, 0, 0), **KINDS[kind])
        ref = oracle.render(tb, o)
        img = g.render_c(o)
        bad = (np.abs(img - ref).max(1) > 1e-3 * (1 + np.abs(ref).max(1))).mean()
        assert np.isfinite(img).all() and bad < 0.01 and rel_l2(img, ref) < 2e-2, kind
