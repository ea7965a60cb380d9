#!/usr/bin/env python
"""torch.profiler view of one Scene.configure() with vertex gradients on the GPU (developer tool): op counts, host time, syncs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, enoki as ek, psdr_cuda
from torch.profiler import profile, ProfilerActivity, record_function
from enoki.cuda_autodiff import Vector3f as Vector3fD
from psdr_cuda.fixtures import scene_path
sc = psdr_cuda.Scene(); sc.load_file(scene_path("cbox_bunny"), False)
sc.opts.width = sc.opts.height = 256; sc.opts.spp = 4; sc.opts.sppe = 4; sc.opts.sppse = 4; sc.opts.log_level = 0
mesh = sc.param_map["Mesh[1]"]
def step():
    v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
    sc.configure()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = prof.key_averages()
print("total aten calls", sum(e.count for e in ev if e.key.startswith("aten::")))
for e in sorted(ev, key=lambda e: -e.self_cpu_time_total)[:25]:
    print("%-46s n=%4d self cpu %7.0f us" % (e.key[:46], e.count, e.self_cpu_time_total))
