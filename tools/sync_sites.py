#!/usr/bin/env python
"""Where one geometry iteration synchronises with the device (developer tool): torch's sync debug mode, the Python line of every
synchronising call inside configure() / renderD / backward on cbox_bunny."""
import os, sys, traceback, warnings, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, enoki as ek, psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
from psdr_cuda.fixtures import scene_path
sc = psdr_cuda.Scene(); sc.load_file(scene_path("cbox_bunny"), False)
sc.opts.width = sc.opts.height = 256; sc.opts.spp = 8; sc.opts.sppe = 4; sc.opts.sppse = 4; sc.opts.log_level = 0
mesh = sc.param_map["Mesh[1]"]
integ = psdr_cuda.DirectIntegrator(1, 1)
def step():
    v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
    sc.configure()
    img = integ.renderD(sc, 0)
    ek.backward(FloatD._wrap(((img.t - 0.3) ** 2).sum().reshape(1)))
    return ek.gradient(v)
for _ in range(3): step()
sites = collections.Counter()
def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "psdr-cuda_amd" in f.filename or "tools/" in f.filename]
    sites[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-4:]))] += 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
for k, n in sites.items(): print(n, k)
