#!/usr/bin/env python
"""Blocking host calls inside one geometry-optimisation iteration (configure with vertex gradients, renderD, a torch loss, enoki.backward)
on cbox_bunny (developer tool): torch.profiler events that copy / synchronise, with their host time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, enoki as ek, psdr_cuda
from torch.profiler import profile, ProfilerActivity
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
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("iteration %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
tot = 0
for e in prof.events():
    if e.name in ("hipMemcpyWithStream", "hipStreamSynchronize", "hipDeviceSynchronize", "hipMemcpy", "hipEventSynchronize") and e.cpu_time_total > 40:
        print("%-26s host %6.0f us" % (e.name, e.cpu_time_total)); tot += e.cpu_time_total
ka = prof.key_averages()
print("blocking total %.2f ms; kernel launches %d; GPU kernel time %.2f ms" % (tot / 1e3, sum(e.count for e in ka if e.key in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel")),
      sum(e.device_time_total for e in ka if e.device_time_total) / 1e3))
if "--configure" in sys.argv:          # what configure() alone launches (the mesh still carries a gradient)
    v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        sc.configure(); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print("configure(): %d device events" % len(ev))
    for e in ev:
        print("  %7.1f us  %s" % (e.device_time_total, e.name[:150]))
    cpu = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is None]
    print("top-level aten ops: %d" % len(cpu))
if "--iteration" in sys.argv:          # every device event of one whole iteration, in order
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    ev = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
    print("iteration: %d device events, %.2f ms of device time" % (len(ev), sum(e.device_time_total for e in ev) / 1e3))
    for e in ev:
        print("  %7.1f us  %s" % (e.device_time_total, e.name[:130]))
