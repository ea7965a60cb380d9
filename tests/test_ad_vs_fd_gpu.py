"""The reference's own validation method (examples/run_test.py:44-231, examples/config.py): the AD derivative
image against central finite differences of renderC -- here on the PRODUCT alone (HIP kernels through the
Python surface, no oracle involved), for the perturbation types of the reference's scenarios: albedo,
roughness, environment rotation (smooth integrands: same sample streams on both sides, tight bounds) and an
object translation (visibility changes: interior + primary-edge + secondary-edge terms, statistical bound)."""
import numpy as np
import pytest
import torch

import enoki as ek
import psdr_cuda
from helpers import FloatD, Matrix4fD, Vector3fD, rel_l2
from psdr_cuda.fixtures import scene_path

pytestmark = pytest.mark.gpu


def scene(name, res, spp, sppe=0, sppse=0):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    return sc


def render_c_fixed_streams(integ, sc):
    """renderC on the sample streams a fresh scene starts with (common random numbers for the two FD sides)"""
    sc._rng_offset = [0, 0, 0]
    return integ.renderC(sc).numpy().astype(np.float64)


def ad_image(integ, sc, P):
    sc._rng_offset = [0, 0, 0]
    img = integ.renderD(sc)
    ek.forward(P, free_graph=True)
    return ek.gradient(img).numpy().astype(np.float64)


def test_albedo_forward_and_backward():
    integ = psdr_cuda.PathTracer(max_depth=3)
    eps = 1e-2

    def build(delta, grad=False):
        sc = scene("cbox", 48, 64)
        P = FloatD(delta)
        if grad:
            ek.set_requires_gradient(P)
        b = sc.param_map["BSDF[0]"]
        b.reflectance.data = Vector3fD(ek.detach(b.reflectance.data).t + torch.tensor([1.0, 0.5, 0.25], device="cuda") * P.t)
        sc.configure()
        return sc, P
    sc, P = build(0.0, True)
    ad = ad_image(integ, sc, P)
    fd = (render_c_fixed_streams(integ, build(eps)[0]) - render_c_fixed_streams(integ, build(-eps)[0])) / (2 * eps)
    assert np.abs(fd).mean() > 0.05 and rel_l2(ad, fd) < 2e-3
    # reverse mode: d sum(w * image) / dP through backward()
    sc, P = build(0.0, True)
    sc._rng_offset = [0, 0, 0]
    img = integ.renderD(sc)
    w = torch.linspace(0.5, 1.5, 48 * 48 * 3, device="cuda").reshape(-1, 3)
    (img.t * w).sum().backward()
    gP = float(ek.gradient(P).numpy().reshape(-1)[0])
    assert abs(gP / float((w.cpu().numpy() * fd).sum()) - 1) < 2e-3


def test_roughness():
    integ = psdr_cuda.DirectIntegrator(1, 1)
    eps = 2e-3

    def build(delta, grad=False):
        sc = scene("cbox_rough", 48, 64)
        bs = sc.param_map["BSDF[id=metal]"]
        P = FloatD(delta)
        if grad:
            ek.set_requires_gradient(P)
        base = (ek.detach(bs.alpha_u.data), ek.detach(bs.alpha_v.data))
        bs.alpha_u.data = FloatD(base[0]) + P
        bs.alpha_v.data = FloatD(base[1]) + P
        sc.configure()
        return sc, P
    sc, P = build(0.0, True)
    ad = ad_image(integ, sc, P)
    fd = (render_c_fixed_streams(integ, build(eps)[0]) - render_c_fixed_streams(integ, build(-eps)[0])) / (2 * eps)
    # BSDF-sampled directions move with alpha (the reference keeps that dependency: roughconductor.cpp:79-92), so
    # same-stream FD sees rays crossing geometry edges as jumps: compare the bulk and the totals
    err = np.abs(ad - fd).max(1) / (np.abs(fd).max(1) + 0.05)
    assert np.abs(fd).mean() > 0.01 and np.median(err) < 0.02 and abs(ad.sum() / fd.sum() - 1) < 0.03


def test_environment_rotation():
    integ = psdr_cuda.DirectIntegrator(1, 0)          # BSDF sampling only: the light samples move with the map (detached)
    eps = 1e-4

    def build(angle, grad=False):
        sc = scene("bunny_env", 48, 16)
        P = FloatD(angle)
        if grad:
            ek.set_requires_gradient(P)
        sc.param_map["Emitter[0]"].set_transform(Matrix4fD.rotate(Vector3fD([0., 1., 0.]), P))
        sc.configure()
        return sc, P
    sc, P = build(0.0, True)
    ad = ad_image(integ, sc, P)
    fd = (render_c_fixed_streams(integ, build(eps)[0]) - render_c_fixed_streams(integ, build(-eps)[0])) / (2 * eps)
    err = np.abs(ad - fd).max(1) / (np.abs(fd).max(1) + 1e-2)
    assert np.abs(fd).mean() > 0.05 and (err < 0.03).mean() > 0.97 and abs(ad.sum() / fd.sum() - 1) < 0.02


def test_object_translation_needs_all_three_terms():
    """cbox_MIS-style scenario (config.py:46-78): an occluder moves; AD = interior + primary edges + secondary
    edges.  FD of a Monte-Carlo image with visibility changes is noisy: many samples, coarse pixels, and the
    comparison also shows that each boundary term is needed."""
    res, spp = 24, 8192
    direction = [1.0, 0.5, 0.0]

    def build(delta, grad=False, sppe=0, sppse=0, n=spp):
        sc = scene("cbox_occluder", res, n, sppe, sppse)
        P = FloatD(delta)
        if grad:
            ek.set_requires_gradient(P)
        sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD(direction) * P))
        sc.configure()
        return sc, P
    integ = psdr_cuda.DirectIntegrator(1, 1)
    sc, P = build(0.0, True, spp, spp)
    ad = ad_image(integ, sc, P)
    sc, P = build(0.0, True, spp, 0)
    ad_no_sec = ad_image(integ, sc, P)
    sc, P = build(0.0, True, 0, spp)
    ad_no_prim = ad_image(integ, sc, P)
    eps, M = 1.0, 131072
    fd = (integ.renderC(build(eps, n=M)[0]).numpy().astype(np.float64) - integ.renderC(build(-eps, n=M)[0]).numpy()) / (2 * eps)
    e_all, e_no_sec, e_no_prim = rel_l2(ad, fd), rel_l2(ad_no_sec, fd), rel_l2(ad_no_prim, fd)
    assert e_all < 0.08, (e_all, e_no_sec, e_no_prim)
    assert e_no_sec > e_all + 0.05 and e_no_prim > e_all + 0.05, (e_all, e_no_sec, e_no_prim)
