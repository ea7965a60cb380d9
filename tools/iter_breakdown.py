#!/usr/bin/env python
"""Where one inverse-rendering iteration spends its time (developer tool): configure() (torch table chain +
host BVH build), renderD (primal), backward (reverse kernels + torch chain), per scene."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import enoki as ek
import psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
from psdr_cuda.fixtures import scene_path


def sync():
    torch.cuda.synchronize()


for scene, mesh_key in (("cbox", "Mesh[0]"), ("cbox_bunny", "Mesh[1]")):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(scene), False)
    sc.opts.width = sc.opts.height = 256
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 16, 8, 8, 0
    mesh = sc.param_map[mesh_key]
    v0 = mesh.vertex_positions.t.detach().clone()
    integ = psdr_cuda.DirectIntegrator(1, 1)
    t = {"configure": 0.0, "renderD": 0.0, "backward": 0.0, "(reverse kernels)": 0.0}
    orig_rev = integ._render_rev

    def timed_rev(*a, **k):
        sync(); t0 = time.perf_counter()
        r = orig_rev(*a, **k)
        sync(); t["(reverse kernels)"] += time.perf_counter() - t0
        return r
    integ._render_rev = timed_rev
    n = 6
    for it in range(n + 1):
        P = Vector3fD(v0.clone())
        ek.set_requires_gradient(P)
        mesh.vertex_positions = P
        sync(); a = time.perf_counter()
        sc.configure()
        sync(); b = time.perf_counter()
        img = integ.renderD(sc, 0)
        sync(); c = time.perf_counter()
        loss = (img.t * img.t).sum()
        loss.backward()
        sync(); d = time.perf_counter()
        if it:                      # first iteration warms up
            t["configure"] += b - a; t["renderD"] += c - b; t["backward"] += d - c
        else:
            t["(reverse kernels)"] = 0.0
    print("%-11s T=%6d  " % (scene, sc.tables(0)["num_tris"]) + "  ".join("%s %.2f ms" % (k, v / n * 1e3) for k, v in t.items()))
