#!/usr/bin/env python
"""Host-to-device copies and host syncs inside one bench surface step (developer tool): which Python lines create device tensors from host
data (a synchronous copy queued behind the running render kernel stalls the host until that kernel ends)."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import bench
_argv = sys.argv; sys.argv = sys.argv[:1]; args = bench.parse(); sys.argv = _argv
w = bench.Workload(args, 1)
for _ in range(3):
    w.surface_step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
which = sys.argv[1] if len(sys.argv) > 1 else "forward"
step = w.surface_reverse_step if which == "reverse" else w.surface_step
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("hipMemcpyWithStream", "hipMemcpyAsync", "hipStreamSynchronize", "hipDeviceSynchronize", "aten::item", "aten::_local_scalar_dense", "aten::_to_copy", "aten::nonzero"):
        st = [s for s in (e.stack or []) if "psdr" in s or "enoki" in s or "bench.py" in s][:3]
        print("%-26s cpu %6.0f us  %s" % (e.name, e.cpu_time_total, " <- ".join(x.split("/")[-1] for x in st)))
