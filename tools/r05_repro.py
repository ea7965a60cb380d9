#!/usr/bin/env python
"""Round 5: the primal image of renderD (PathTracer, vertex positions with a gradient -> the recording wavefront, PSDR_FLAG_KEEP_RECORDS) of two FRESH
identical scenes in one process differed in one pixel, reproducibly.  Which call is the odd one?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import enoki as ek
import psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
from psdr_cuda.fixtures import scene_path

def scene(grad):
    sc = psdr_cuda.Scene(); sc.load_file(scene_path("cbox_bunny"), False)
    sc.opts.width = sc.opts.height = 256
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 32, 0, 0, 0
    if grad:
        mesh = sc.param_map["Mesh[1]"]
        v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
    sc.configure()
    return sc

def cmp(tag, a, b):
    d = np.abs(a - b); i = np.unravel_index(np.argmax(d), d.shape)
    print("%-58s pixels off by > 1e-4: %4d  max |diff| %.3e at %s" % (tag, int((d > 1e-4).sum()), d[i], i), flush=True)

imgs = {}
for name, grad, kind in (("grad#1", True, "D"), ("grad#2", True, "D"), ("plain#1", False, "C"), ("grad#3", True, "D"), ("plain#2", False, "C")):
    sc = scene(grad)
    pt = psdr_cuda.PathTracer(3)
    img = pt.renderD(sc, 0) if kind == "D" else pt.renderC(sc, 0)
    imgs[name] = img.numpy().copy()
    tb = sc._tables if hasattr(sc, "_tables") else None
    if tb is not None and "tri_info" in tb:
        imgs[name + "_rows"] = tb["tri_info"].detach().cpu().numpy().copy()
for a, b in (("grad#1", "grad#2"), ("grad#2", "grad#3"), ("plain#1", "plain#2"), ("grad#1", "plain#1"), ("grad#2", "plain#1"), ("grad#3", "plain#1")):
    cmp("image %s vs %s" % (a, b), imgs[a], imgs[b])
    if a + "_rows" in imgs and b + "_rows" in imgs:
        ra, rb = imgs[a + "_rows"], imgs[b + "_rows"]
        if ra.shape == rb.shape:
            print("    triangle rows: %d words differ, max |diff| %.3e" % (int((ra != rb).sum()), float(np.abs(ra - rb).max())), flush=True)
