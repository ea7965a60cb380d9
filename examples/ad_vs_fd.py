#!/usr/bin/env python
"""AD-vs-finite-difference comparison images, in the spirit of the reference's examples/run_test.py (its own
validation harness): for each scenario the derivative image from renderD + forward(P) and the central finite
difference of renderC are written as EXR files (<out>/<name>_ad.exr, _fd.exr, _orig.exr).

    python examples/ad_vs_fd.py [--out results] [--res 128] [scenario ...]

Scenarios (fixtures of psdr-cuda_amd/data): albedo, roughness, envmap_rotate, translate.
Needs an MI355X (the render path has no CPU fallback)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "psdr-cuda_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import enoki as ek  # noqa: E402
import psdr_cuda  # noqa: E402
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD  # noqa: E402
from psdr_cuda.exr import save_exr_rgb  # noqa: E402
from psdr_cuda.fixtures import scene_path  # noqa: E402


def load(name, res, spp, sppe=0, sppse=0):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    return sc


def albedo(sc, P):
    b = sc.param_map["BSDF[0]"]
    b.reflectance.data = Vector3fD(ek.detach(b.reflectance.data).t + torch.tensor([1.0, 0.5, 0.25], device="cuda") * P.t)


def roughness(sc, P):
    bs = sc.param_map["BSDF[id=metal]"]
    base = (ek.detach(bs.alpha_u.data), ek.detach(bs.alpha_v.data))
    bs.alpha_u.data = FloatD(base[0]) + P
    bs.alpha_v.data = FloatD(base[1]) + P


def envmap_rotate(sc, P):
    sc.param_map["Emitter[0]"].set_transform(Matrix4fD.rotate(Vector3fD([0., 1., 0.]), P))


def translate(sc, P):
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.5, 0.0]) * P))


SCENARIOS = {
    # name: (scene, integrator, perturbation, eps, (spp, sppe, sppse) for AD, spp for FD, npass)
    "albedo": ("cbox", lambda: psdr_cuda.PathTracer(max_depth=3), albedo, 1e-2, (256, 0, 0), 256, 1),
    "roughness": ("cbox_rough", lambda: psdr_cuda.DirectIntegrator(1, 1), roughness, 2e-3, (256, 0, 0), 256, 1),
    "envmap_rotate": ("bunny_env", lambda: psdr_cuda.DirectIntegrator(2, 2), envmap_rotate, 1e-2, (256, 0, 0), 1024, 8),
    "translate": ("cbox_occluder", lambda: psdr_cuda.DirectIntegrator(1, 1), translate, 1.0, (1024, 1024, 1024), 8192, 8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="results")
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("scenarios", nargs="*", default=list(SCENARIOS))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for name in args.scenarios:
        scene, make_integ, perturb, eps, (spp, sppe, sppse), fd_spp, npass = SCENARIOS[name]
        integ = make_integ()
        n = args.res * args.res
        ad = np.zeros((n, 3))
        orig = None
        for _ in range(npass):
            sc = load(scene, args.res, spp, sppe, sppse)
            P = FloatD(0.)
            ek.set_requires_gradient(P)
            perturb(sc, P)
            sc.configure()
            img = integ.renderD(sc)
            ek.forward(P, free_graph=True)
            ad += ek.gradient(img).numpy() / npass
            orig = img.numpy()
        fd = np.zeros((n, 3))
        for _ in range(npass):
            sides = []
            for sgn in (+1, -1):
                sc = load(scene, args.res, fd_spp)
                perturb(sc, FloatD(sgn * eps))
                sc.configure()
                sides.append(integ.renderC(sc).numpy().astype(np.float64))
            fd += (sides[0] - sides[1]) / (2 * eps) / npass
        for tag, a in (("orig", orig), ("ad", ad), ("fd", fd)):
            save_exr_rgb(os.path.join(args.out, "%s_%s.exr" % (name, tag)), a.reshape(args.res, args.res, 3).astype(np.float32))
        rel = np.linalg.norm(ad - fd) / max(np.linalg.norm(fd), 1e-30)
        print("%-14s rel-L2(AD, FD) = %.4f   sum AD %.4f   sum FD %.4f   -> %s/%s_{orig,ad,fd}.exr" % (name, rel, ad.sum(), fd.sum(), args.out, name))


if __name__ == "__main__":
    main()
