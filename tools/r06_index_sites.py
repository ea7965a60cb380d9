"""round 6: which `x[index tensor]` gathers on tensors that require grad run during one forward-mode step through the surface (their backward is torch's
sort-based index_put(accumulate): 0.25 ms a call on the GPU).  Prints the call sites."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "psdr-cuda_amd"))
import numpy as np, torch
import enoki as ek, psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
from psdr_cuda.fixtures import scene_path
sc = psdr_cuda.Scene(); sc.load_file(scene_path("bunny_light"), False)
sc.opts.width = sc.opts.height = 64
sc.opts.spp = sc.opts.sppe = sc.opts.sppse = 4
sc.opts.log_level = 0
m = sc.param_map["Mesh[0]"]; sc.configure()
integ = psdr_cuda.DirectIntegrator(1, 1)
sites = collections.Counter()
orig = torch.Tensor.__getitem__
def has_t(i):
    return isinstance(i, torch.Tensor) or (isinstance(i, tuple) and any(isinstance(j, torch.Tensor) for j in i))
def gi(self, idx):
    r = orig(self, idx)
    if has_t(idx) and self.requires_grad:
        fr = traceback.extract_stack(limit=2)[0]
        sites["%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, (fr.line or "")[:150])] += 1
    return r
torch.Tensor.__getitem__ = gi
P = FloatD(0.); ek.set_requires_gradient(P)
m.set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P)); sc.configure()
img = integ.renderD(sc)
ek.forward(P, free_graph=True)
for k, v in sites.most_common(): print(v, k)
