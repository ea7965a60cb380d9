#!/usr/bin/env python
"""Inverse rendering through renderD + backward, after the reference's docs/inverse_diff_render.rst:

    python examples/inverse_rendering.py albedo       # recover a wall colour (material parameter)
    python examples/inverse_rendering.py translation  # move an occluder back (geometry: all three terms)
    python examples/inverse_rendering.py envmap       # recover the environment map's pixels under a metal bunny

Needs an MI355X (the render path has no CPU fallback)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "psdr-cuda_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import enoki as ek  # noqa: E402
import psdr_cuda  # noqa: E402
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD  # noqa: E402
from psdr_cuda.fixtures import scene_path  # noqa: E402


def load(name, res=64, spp=16, sppe=0, sppse=0):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    return sc


def loop(sc, integ, target, params, lr, steps, after_step=None, report=None):
    opt = torch.optim.Adam([p.t for p in params], lr=lr)
    t0 = time.perf_counter()
    for it in range(steps):
        opt.zero_grad()
        sc.configure()
        img = integ.renderD(sc)
        loss = ek.hmean(ek.hsum(ek.sqr(img - Vector3fD._wrap(target))))
        ek.backward(loss)
        opt.step()
        if after_step:
            after_step()
        if it % 10 == 0 or it == steps - 1:
            print("  step %3d  loss %.5f  %s" % (it, float(loss.t.item()), report() if report else ""))
    torch.cuda.synchronize()
    print("  %.1f ms per step" % ((time.perf_counter() - t0) / steps * 1e3))


def albedo():
    integ = psdr_cuda.PathTracer(max_depth=2)
    ref = load("cbox", spp=512); ref.configure()
    target = integ.renderC(ref).torch().clone()
    sc = load("cbox")
    refl = sc.param_map["BSDF[id=white]"].reflectance
    refl.data = Vector3fD([0.4, 0.6, 0.8])
    ek.set_requires_gradient(refl.data)
    loop(sc, integ, target, [refl.data], 0.05, 60, lambda: refl.data.t.data.clamp_(0.01, 0.99),
         lambda: "albedo %s (target 0.95 0.95 0.95)" % np.round(refl.data.numpy().reshape(3), 3))


def translation():
    integ = psdr_cuda.DirectIntegrator(1, 1)
    ref = load("cbox_occluder", spp=64); ref.configure()
    target = integ.renderC(ref).torch().clone()
    sc = load("cbox_occluder", spp=16, sppe=16, sppse=16)
    P = FloatD(12.0)
    ek.set_requires_gradient(P)
    mesh = sc.param_map["Mesh[1]"]

    def apply():
        mesh.set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P))
    apply()
    loop(sc, integ, target, [P], 1.0, 50, apply, lambda: "offset %.3f (target 0)" % float(P.t.item()))


def envmap():
    integ = psdr_cuda.DirectIntegrator(1, 1)
    ref = load("bunny_env", spp=256); ref.configure()
    target = integ.renderC(ref).torch().clone()
    truth = ref.param_map["Emitter[0]"].radiance.data.numpy().copy()
    sc = load("bunny_env", spp=32)
    env = sc.param_map["Emitter[0]"]
    env.radiance.data = Vector3fD(torch.full((truth.shape[0], 3), 0.5, device="cuda"))
    ek.set_requires_gradient(env.radiance.data)
    loop(sc, integ, target, [env.radiance.data], 0.05, 80, lambda: env.radiance.data.t.data.clamp_(0.0, 100.0),
         lambda: "mean |map - truth| %.4f" % float(np.abs(env.radiance.data.numpy() - truth).mean()))


if __name__ == "__main__":
    {"albedo": albedo, "translation": translation, "envmap": envmap}[sys.argv[1] if len(sys.argv) > 1 else "albedo"]()
