"""round 6: where the time of the harness' forward mode goes THROUGH THE SURFACE (bench row c3_bunny_fwd3_translation): per call, HIP-event times of
set_transform + configure / renderD (primal launch) / enoki.forward (table-chain JVP + forward launch), plus torch's own kernel table."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "psdr-cuda_amd"))
import numpy as np, torch
import enoki as ek, psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
from psdr_cuda.fixtures import scene_path
sc = psdr_cuda.Scene(); sc.load_file(scene_path("bunny_light"), False)
sc.opts.width = sc.opts.height = 512
sc.opts.spp = sc.opts.sppe = sc.opts.sppse = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sc.opts.log_level = 0
m = sc.param_map["Mesh[0]"]; sc.configure()
integ = psdr_cuda.DirectIntegrator(1, 1)
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
def step(prof=False):
    t = [ev()]
    P = FloatD(0.); ek.set_requires_gradient(P)
    m.set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.0]) * P)); sc.configure(); t.append(ev())
    img = integ.renderD(sc); t.append(ev())
    ek.forward(P, free_graph=True); g = ek.gradient(img); t.append(ev())
    m.set_transform(np.eye(4, dtype=np.float32))
    torch.cuda.synchronize()
    return [t[i].elapsed_time(t[i + 1]) for i in range(3)]
step(); step()
for _ in range(3):
    w = time.perf_counter(); r = step(); w = (time.perf_counter() - w) * 1e3
    print("configure %.2f  renderD %.2f  forward %.2f ms | wall %.2f" % (r[0], r[1], r[2], w))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    step()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
