#!/usr/bin/env python
"""Kernel launches and host time per section of one geometry iteration (configure with vertex gradients + backward): developer probe."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, enoki as ek, psdr_cuda
from psdr_cuda import scene as S
from torch.profiler import profile, ProfilerActivity, record_function
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD
from psdr_cuda.fixtures import scene_path

def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function("SEC:" + label):
            return f(*a, **k)
    setattr(obj, name, g)
wrap(S.Scene, "_configure_meshes", "configure_meshes"); wrap(S.Scene, "_secondary_edges", "secondary_edges"); wrap(S.Scene, "_material_tables", "material_tables")
wrap(S.PerspectiveCamera, "configure", "camera+primary_edges"); wrap(S, "process_mesh", "process_mesh")
sc = psdr_cuda.Scene(); sc.load_file(scene_path("cbox_bunny"), False)
sc.opts.width = sc.opts.height = 256; sc.opts.spp = 8; sc.opts.sppe = 4; sc.opts.sppse = 4; sc.opts.log_level = 0
mesh = sc.param_map["Mesh[1]"]
integ = psdr_cuda.DirectIntegrator(1, 1)
def step():
    v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
    with record_function("SEC:configure_total"):
        sc.configure()
    with record_function("SEC:renderD"):
        img = integ.renderD(sc, 0)
    with record_function("SEC:backward_total"):
        ek.backward(FloatD._wrap(((img.t - 0.3) ** 2).sum().reshape(1)))
    return ek.gradient(v)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = list(prof.events())
secs = [e for e in ev if e.name.startswith("SEC:")]
launch = [e for e in ev if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel")]
for s in secs:
    n = sum(1 for l in launch if s.time_range.start <= l.time_range.start <= s.time_range.end)
    print("%-28s host %7.0f us  launches %4d" % (s.name[4:], s.cpu_time_total, n))
print("total launches", len(launch))
