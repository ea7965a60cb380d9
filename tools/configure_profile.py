#!/usr/bin/env python
"""Where Scene.configure() spends its host time when a mesh's vertices carry a gradient (developer tool): cProfile by function."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, enoki as ek, psdr_cuda
from enoki.cuda_autodiff import Vector3f as Vector3fD
from psdr_cuda.fixtures import scene_path
for name, key in (("cbox", "Mesh[0]"), ("cbox_bunny", "Mesh[1]")):
    sc = psdr_cuda.Scene(); sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = 256; sc.opts.spp = 4; sc.opts.sppe = 4; sc.opts.sppse = 4; sc.opts.log_level = 0
    mesh = sc.param_map[key]

    def step():
        v = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(v); mesh.vertex_positions = v
        sc.configure()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize(); print("%s: configure %.2f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        step()
    torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative")
    import io
    buf = io.StringIO(); st.stream = buf; st.print_stats(28)
    for line in buf.getvalue().splitlines():
        if "scene.py" in line or "_array.py" in line or "core.py" in line or "ncalls" in line:
            print(line[:150])
