"""EnvironmentMap emitter (reference src/emitter/envmap.cpp, bounding mesh src/scene/scene.cpp:135-180,
loader scene_loader.cpp:294-311): host tables, oracle closed forms / estimator identities / AD-vs-FD,
and product-vs-oracle parity of the device code on the host (tests/hostcheck).  The GPU half is in
tests/test_envmap_gpu.py."""
import numpy as np
import pytest
import torch

import enoki as ek
import oracle
import psdr_cuda
from helpers import FloatD, Matrix4fD, Vector3fD, host_render, load_scene, rel_l2, tangents_wrt
from psdr_cuda import _abi
from psdr_cuda.fixtures import scene_path
from psdr_cuda.scene import look_at


def env_scene(angle=0.0, grad=False, res=24, spp=8, name="bunny_env"):
    """the reference's `envmap_rotate` perturbation (examples/utils/differential.py:33-35)"""
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
    P = FloatD(angle)
    if grad:
        ek.set_requires_gradient(P)
    sc.param_map["Emitter[0]"].set_transform(Matrix4fD.rotate(Vector3fD([0., 1., 0.]), P))
    sc.configure()
    return sc, P


def constant_env_floor(value, scale, albedo, res=8, spp=16):
    """a diffuse floor seen from above under a CONSTANT environment map"""
    sc = psdr_cuda.Scene()
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, 0, 0, 0
    cam = psdr_cuda.PerspectiveCamera(10.0, 0.1, 1e4)
    cam.to_world = look_at([0, 400, 50], [0, 0, 50], [0, 0, -1])
    sc.add_sensor(cam)
    b = psdr_cuda.Diffuse(albedo)
    b.id = "floor"
    sc.add_bsdf(b)
    m = psdr_cuda.Mesh()
    m.load(scene_path("cbox").replace("scenes/cbox.xml", "objects/cbox/floor.obj"))
    m.use_face_normals = True
    sc.add_mesh(m, b)
    env = psdr_cuda.EnvironmentMap()
    env.radiance = psdr_cuda.Bitmap3fD(3, 2, torch.tensor([value] * 6, dtype=torch.float32))
    env.scale = FloatD(scale)
    sc.add_environment_map(env)
    sc.finalize()
    sc.configure()
    return sc


# ------------------------------------------------------------------------------ host tables
def test_loader_builds_bounding_mesh_once_and_tables():
    sc, _ = env_scene()
    tb = sc.tables(0)
    assert sc.num_meshes == 2 and len(sc.m_meshes) == 2 and tb["num_tris"] == 4968 + 12
    assert tb["mesh_bsdf"].tolist() == [0, -1] and tb["mesh_emitter"].tolist() == [-1, 0]
    assert tb["env_emitter"] == 0 and tb["env_tex"][1:] == [64, 32] and tb["env_reso"] == [126, 62]
    assert tb["emitter_i"][0].tolist() == [1, 4968, 12, 0] and float(tb["emitter_f"][0, 3]) == 1.0
    # AABB = meshes + camera, grown by 5 % of its smallest extent (scene.cpp:135-141); the box is closed
    lo, hi = tb["env_f"][19:22], tb["env_f"][22:25]
    v = sc.m_meshes[0]._vertex_positions
    assert (lo < v.min(0)[0]).all() and (hi > v.max(0)[0]).all() and float(hi[2]) > 400.0
    rows = tb["tri_info"][4968:]
    assert torch.allclose(rows[:, 21].sum(), 2 * ((hi - lo)[0] * (hi - lo)[1] + (hi - lo)[1] * (hi - lo)[2] + (hi - lo)[0] * (hi - lo)[2]), rtol=1e-5)
    # from_world is the inverse rotation, scale from the XML
    R = tb["env_f"][0:9].reshape(3, 3) @ tb["env_f"][9:18].reshape(3, 3)
    assert torch.allclose(R, torch.eye(3), atol=1e-6) and abs(float(tb["env_f"][18]) - 0.9) < 1e-7
    # cell masses: luminance * sin(theta), normalised pmf sums to one
    assert tb["env_cmf"].shape[0] == 126 * 62 and abs(float(tb["env_cmf"][-1]) / tb["env_sum"] - 1) < 1e-5
    # re-configure keeps ONE bounding mesh with its original vertices (m_has_bound_mesh)
    before = sc.m_meshes[1]._vertex_positions.clone()
    sc.configure()
    assert len(sc.m_meshes) == 2 and torch.equal(before, sc.m_meshes[1]._vertex_positions)
    assert float(sc.tables(0)["emitter_f"][0, 3]) == 1.0


def test_loader_rejects_second_envmap_and_unknown_emitters():
    xml = open(scene_path("bunny_env")).read()
    two = xml.replace("</emitter>", "</emitter>\n<emitter type=\"envmap\"><string name=\"filename\" value=\"./data/envmaps/synthetic_sky_64x32.exr\"/></emitter>", 1)
    with pytest.raises(RuntimeError, match="only allowed to have one envmap"):
        psdr_cuda.Scene().load_string(two, False)
    with pytest.raises(RuntimeError, match="Unsupported emitter"):
        psdr_cuda.Scene().load_string(xml.replace('type="envmap"', 'type="point"'), False)


# ------------------------------------------------------------------------ oracle: closed forms
def test_constant_environment_closed_form():
    """Under a constant environment L the outgoing radiance of a diffuse plane is albedo * L * scale:
    exactly (zero variance) with cosine-weighted BSDF sampling, in expectation with light sampling / MIS."""
    L, scale, albedo = [2.0, 1.0, 0.5], 1.5, [0.8, 0.5, 0.3]
    sc = constant_env_floor(L, scale, albedo)
    tb = sc.tables(0)
    want = np.array(L) * scale * np.array(albedo)
    img = oracle.render(tb, _abi.make_opts(spp=16, bsdf_samples=1, light_samples=0))
    assert np.abs(img / want - 1).max() < 2e-5
    for kw in (dict(bsdf_samples=0, light_samples=1), dict(bsdf_samples=1, light_samples=1)):
        img = oracle.render(tb, _abi.make_opts(spp=16, **kw))
        assert np.abs(img.mean(0) / want - 1).max() < 0.03, kw


def test_background_pixels_show_the_map():
    """a primary ray that leaves the scene returns scale * radiance(direction) (EnvironmentMap::eval)"""
    sc, _ = env_scene(res=16, spp=4)
    tb = sc.tables(0)
    img = oracle.render(tb, _abi.make_opts(spp=4, bsdf_samples=1, light_samples=1)).reshape(16, 16, 3)
    corner = img[0, 0]              # top-left pixel sees only the sky
    # direction of that pixel -> lat-long lookup through the host Bitmap mirror
    cam = tb["cam"]
    s2c = cam[_abi.CAM_SAMPLE_TO_CAMERA:_abi.CAM_SAMPLE_TO_CAMERA + 16].reshape(4, 4)
    tw = cam[_abi.CAM_TO_WORLD:_abi.CAM_TO_WORLD + 16].reshape(4, 4)
    v = s2c @ torch.tensor([0.5 / 16, 0.5 / 16, 0.0, 1.0])
    d = tw[:3, :3] @ torch.nn.functional.normalize(v[:3] / v[3], dim=0)
    loc = tb["env_f"][0:9].reshape(3, 3) @ d
    uv = torch.stack([torch.atan2(loc[0], -loc[2]) / (2 * np.pi), torch.acos(loc[1].clamp(-1, 1)) / np.pi]) % 1.0
    want = sc.m_emitter_env.radiance.eval(uv.reshape(1, 2), False).numpy()[0] * 0.9
    assert np.abs(corner / want - 1).max() < 0.05          # 4 jittered samples inside one pixel


def test_bsdf_light_and_mis_estimators_agree_under_environment_light():
    sc, _ = env_scene(res=12, spp=256)
    tb = sc.tables(0)
    means = [oracle.render(tb, _abi.make_opts(spp=256, **kw)).mean(0) for kw in
             (dict(bsdf_samples=1, light_samples=0), dict(bsdf_samples=0, light_samples=1), dict(bsdf_samples=1, light_samples=1))]
    for m in means[1:]:
        assert np.abs(m / means[0] - 1).max() < 0.03


def test_environment_rotation_ad_matches_fd():
    """forward-mode derivative w.r.t. the env-map rotation angle vs central differences of the fp64 oracle
    on the same sample streams (BSDF sampling only: light sampling moves with the map, which the
    reference detaches, so AD and same-stream FD then agree in expectation only)."""
    sc, P = env_scene(0.0, True, res=24, spp=4)
    tb = sc.tables(0)
    tan = tangents_wrt(tb, P)
    assert tan["env_f"] is not None and all(tan[k] is None for k in tan if k != "env_f")
    o = _abi.make_opts(spp=4, bsdf_samples=1, light_samples=0)
    eps = 1e-4
    ip = oracle.render(env_scene(eps, res=24, spp=4)[0].tables(0), o, mode=1, precision=1)[0].astype(np.float64)
    im = oracle.render(env_scene(-eps, res=24, spp=4)[0].tables(0), o, mode=1, precision=1)[0].astype(np.float64)
    fd = (ip - im) / (2 * eps)
    ad = oracle.render(tb, o, mode=1, tangents=tan, precision=1)[1]
    assert np.abs(fd).mean() > 0.05
    err = np.abs(ad - fd).max(-1) / (np.abs(fd).max(-1) + 1e-2)
    assert (err < 0.02).mean() > 0.99 and rel_l2(ad, fd) < 5e-3


def test_scale_and_texel_tangents_are_linear():
    """image = scale * f(texels): d/dscale = image / scale for pure env lighting, and the texel tangent
    of a uniformly brightened map equals the image itself"""
    sc, _ = env_scene(res=16, spp=8)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=8, bsdf_samples=1, light_samples=1)
    t = torch.zeros_like(tb["env_f"]); t[18] = 1.0
    img, d_scale = oracle.render(tb, o, mode=1, tangents={"env_f": t})
    assert rel_l2(d_scale * 0.9, img) < 1e-5
    off, w, h = tb["env_tex"]
    tt = torch.zeros_like(tb["texels"]); tt[off:off + w * h * 3] = tb["texels"][off:off + w * h * 3]
    img, d_tex = oracle.render(tb, o, mode=1, tangents={"texels": tt})
    assert rel_l2(d_tex, img) < 1e-5


# ------------------------------------------------------ product device code vs oracle (host)
KINDS = {
    "direct11": dict(bsdf_samples=1, light_samples=1),
    "direct20": dict(bsdf_samples=2, light_samples=0),
    "direct02": dict(bsdf_samples=0, light_samples=2),
    "path3": dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3),
}


@pytest.mark.parametrize("scene", ["bunny_env", "cbox_env"])
@pytest.mark.parametrize("kind", list(KINDS))
def test_hostcheck_render_c(scene, kind):
    sc, _ = load_scene(scene, res=24, spp=8)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=8, rng_offset=(3, 0, 0), **KINDS[kind])
    a, b = host_render(tb, o), oracle.render(tb, o)
    assert np.isfinite(a).all() and rel_l2(a, b) < 2e-5


def test_hostcheck_forward_env_parameters():
    sc, P = env_scene(0.0, True, res=24, spp=8)
    tb = sc.tables(0)
    g = torch.Generator().manual_seed(1)
    tan = tangents_wrt(tb, P)
    tan["texels"] = torch.rand(tb["texels"].shape, generator=g)
    tan["env_f"] = tan["env_f"].clone(); tan["env_f"][18] = 0.7
    for kind in ("direct11", "path3"):
        o = _abi.make_opts(spp=8, **KINDS[kind])
        ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
        img, dimg = host_render(tb, o, mode=1, tangents=tan)
        # renderD traces the primary ray in the solid-angle form: isolated bunny samples flip (DESIGN.md
        # "numerical fragility"), so bound the differing-pixel fraction and a looser rel-L2
        bad = (np.abs(dimg - ref_d).max(1) > 1e-3 * (1 + np.abs(ref_d).max(1))).mean()
        assert bad < 0.02 and rel_l2(dimg, ref_d) < 2e-2 and np.abs(ref_d).max() > 0, kind


def test_hostcheck_forward_geometry_with_edges_against_environment_light():
    sc, P = load_scene("cbox_env", res=24, spp=8, sppe=8, sppse=8, translate=(1, (1.0, 0.5, 0.0)))
    tb = sc.tables(0)
    assert tb["num_sec_edges"] > 0 and tb["num_prim_edges"] > 0
    tan = tangents_wrt(tb, P)
    o = _abi.make_opts(spp=8, sppe=8, sppse=8, rng_offset=(0, 5, 9), bsdf_samples=1, light_samples=1)
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    img, dimg = host_render(tb, o, mode=1, tangents=tan)
    bad = (np.abs(dimg - ref_d).max(1) > 1e-3 * (1 + np.abs(ref_d).max(1))).mean()
    assert rel_l2(img, ref_img) < 1e-3 and bad < 0.02 and np.abs(ref_d).max() > 0
