"""The product's actual use (docs/inverse_diff_render.rst, examples/utils/adam.py): recover parameters by
gradient descent through renderD + enoki.backward on the GPU."""
import numpy as np
import pytest
import torch

import enoki as ek
import psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
from psdr_cuda.fixtures import scene_path

pytestmark = pytest.mark.gpu


def _scene(name, res=48, spp=16, sppe=0, sppse=0):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    return sc


def test_recover_wall_albedo_with_adam():
    integ = psdr_cuda.PathTracer(max_depth=2)
    ref_sc = _scene("cbox", spp=512)
    ref_sc.configure()
    target = integ.renderC(ref_sc).torch().clone()            # white walls = (0.95, 0.95, 0.95); low-noise target
    sc = _scene("cbox")
    refl = sc.param_map["BSDF[id=white]"].reflectance
    refl.data = Vector3fD([0.4, 0.6, 0.8])
    ek.set_requires_gradient(refl.data)
    opt = torch.optim.Adam([refl.data.t], lr=0.05)
    losses = []
    for it in range(60):
        opt.zero_grad()
        sc.configure()
        img = integ.renderD(sc)
        loss = ek.hmean(ek.hsum(ek.sqr(img - Vector3fD._wrap(target))))
        ek.backward(loss)
        opt.step()
        with torch.no_grad():
            refl.data.t.clamp_(0.01, 0.99)
        losses.append(float(loss.t.item()))
    got = refl.data.numpy().reshape(3)
    # the loss floor is the Monte-Carlo variance of a 16-spp render, so judge by the parameters
    assert np.abs(got - 0.95).max() < 0.06, (got, losses[0], losses[-1])


def test_recover_occluder_translation():
    """Geometry: all three terms (interior + primary + secondary edges) drive a translation back."""
    integ = psdr_cuda.DirectIntegrator(1, 1)
    ref_sc = _scene("cbox_occluder", spp=32)
    ref_sc.configure()
    target = integ.renderC(ref_sc).torch().clone()
    sc = _scene("cbox_occluder", spp=16, sppe=16, sppse=16)
    P = FloatD(12.0)                                           # start 12 units off along x
    ek.set_requires_gradient(P)
    opt = torch.optim.Adam([P.t], lr=1.0)
    hist = []
    for it in range(50):
        opt.zero_grad()
        sc.param_map["Mesh[id=occluder]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P))
        sc.configure()
        img = integ.renderD(sc)
        loss = ek.hmean(ek.hsum(ek.sqr(img - Vector3fD._wrap(target))))
        ek.backward(loss)
        opt.step()
        hist.append(float(P.t.item()))
    assert abs(hist[-1]) < 3.0, hist[::5]
