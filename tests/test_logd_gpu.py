"""PathTracer forward mode with tangents on diffuse albedo texels only (the headline's renderD w.r.t. the diffuse albedo, examples/run_test.py:126-129) runs the
LOG-DERIVATIVE kernel: the estimator on plain floats, d/dP [beta c] = beta c * sum (d rho / rho) over the path's vertices (csrc/psdr_device.h li_path_logd,
psdr_kernels.h k_camera_logd).  It must agree with the dual-number kernel (option logd = 0) and the oracle, and hand the launch back to the dual-number
kernel wherever the quotient cannot be formed (a texel with a tangent and a zero albedo) or another table carries a tangent."""
import numpy as np
import pytest
import torch

import oracle
from helpers import GpuScene, load_scene, random_tangents, rel_l2
from psdr_cuda import _abi

pytestmark = pytest.mark.gpu


def fwd(tb, o, tans, logd):
    g = GpuScene(tb, options={"logd": logd})
    return g.render_d_fwd(o, tans)


@pytest.mark.parametrize("depth", [1, 3, 5])
def test_log_derivative_kernel_equals_the_dual_number_kernel_and_the_oracle(depth):
    sc, _ = load_scene("cbox", res=64, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=depth, spp=16)
    tan = random_tangents(tb, ["texels"], seed=11)
    img1, d1 = fwd(tb, o, [tan], 1)
    img0, d0 = fwd(tb, o, [tan], 0)
    assert np.abs(d0).max() > 0
    assert rel_l2(img1, img0) < 2e-6 and rel_l2(d1[0], d0[0]) < 2e-5, (rel_l2(img1, img0), rel_l2(d1[0], d0[0]))
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents=tan)
    assert rel_l2(img1, ref_img) < 1e-4 and rel_l2(d1[0], ref_d) < 1e-3
    # K = 3: d / d(r, g, b) of one BSDF's albedo in one launch (the round-1 headline form)
    t3 = []
    for c in range(3):
        t = torch.zeros_like(tb["texels"]); t[c] = 1.0
        t3.append({"texels": t})
    _, e1 = fwd(tb, o, t3, 1)
    _, e0 = fwd(tb, o, t3, 0)
    for c in range(3):
        assert np.abs(e0[c]).max() > 0 and rel_l2(e1[c], e0[c]) < 2e-5, (c, rel_l2(e1[c], e0[c]))
        assert np.abs(e1[c][:, [k for k in range(3) if k != c]]).max() == 0          # the red albedo moves the red channel only


def test_log_derivative_on_a_bitmap_texture():
    from test_textures import textured_scene
    sc = textured_scene(res=48, spp=16)
    tb = sc.tables(0)
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16)
    tan = random_tangents(tb, ["texels"], seed=3)
    _, d1 = fwd(tb, o, [tan], 1)
    _, d0 = fwd(tb, o, [tan], 0)
    assert np.abs(d0).max() > 0 and rel_l2(d1[0], d0[0]) < 2e-5, rel_l2(d1[0], d0[0])


def test_a_zero_albedo_or_another_tangent_table_hands_the_launch_to_the_dual_number_kernel():
    sc, _ = load_scene("cbox", res=48, spp=16)
    tb = dict(sc.tables(0))
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16)
    # the green channel of the first BSDF's albedo is EXACTLY zero and carries a tangent: d/dP of that channel is not zero, and the quotient
    # d rho / rho does not exist -- the gate kernel sees it and the dual-number kernel runs
    tex = tb["texels"].clone(); tex[1] = 0.0
    tb["texels"] = tex
    t = torch.zeros_like(tex); t[0:3] = 1.0
    img1, d1 = fwd(tb, o, [{"texels": t}], 1)
    img0, d0 = fwd(tb, o, [{"texels": t}], 0)
    ref_img, ref_d = oracle.render(tb, o, mode=1, tangents={"texels": t})
    assert np.abs(d0[0][:, 1]).max() > 0                                              # the zero channel's derivative is there
    assert rel_l2(d1[0], d0[0]) < 2e-6 and rel_l2(d1[0], ref_d) < 1e-3, (rel_l2(d1[0], d0[0]), rel_l2(d1[0], ref_d))
    # a tangent on the emitter's radiance beside the texels: not a texels-only launch
    tb = dict(sc.tables(0))
    tans = {"texels": random_tangents(tb, ["texels"], seed=2)["texels"], "emitter_rad": torch.ones_like(tb["emitter_rad"])}
    _, d1 = fwd(tb, o, [tans], 1)
    _, d0 = fwd(tb, o, [tans], 0)
    _, ref_d = oracle.render(tb, o, mode=1, tangents=tans)
    assert rel_l2(d1[0], d0[0]) < 2e-6 and rel_l2(d1[0], ref_d) < 1e-3
