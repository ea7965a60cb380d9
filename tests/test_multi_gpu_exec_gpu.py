"""The multi-GPU code path EXECUTED on hardware: two processes share the one GPU of the test box (backend gloo,
CUDA tensors), each renders its half of the sample slots through the PUBLIC surface -- renderC, renderD +
enoki.forward, renderD + enoki.backward, preprocess_secondary_edges -- and the collectives inside
psdr_cuda/integrator.py (one all-reduce of [image || derivative images] or of the flat gradient buffer per render call,
one of the guiding mass) must reproduce the single-process result up to the order of the fp32 sums.

(8-GPU runs over RCCL are the driver's; on one GPU `nccl` cannot host two ranks, gloo moves the same buffers.)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_sequence():
    """The call sequence under test; returns a dict of numpy arrays.  Runs in every rank and in the single process."""
    import torch
    import enoki as ek
    import psdr_cuda
    from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
    from psdr_cuda.fixtures import scene_path
    out = {}

    def scene(name, spp, sppe, sppse, res=48):
        sc = psdr_cuda.Scene()
        sc.load_file(scene_path(name), False)
        sc.opts.width = sc.opts.height = res
        sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
        return sc
    # renderC, PathTracer (odd spp: the shards are 4 + 3 samples)
    sc = scene("cbox", 7, 0, 0)
    sc.configure()
    out["c_path"] = psdr_cuda.PathTracer(3).renderC(sc).numpy()
    out["c_direct_second_pass"] = psdr_cuda.DirectIntegrator(2, 1).renderC(sc).numpy()       # RNG offsets advanced on every rank alike
    # renderD + forward, all three terms
    sc = scene("cbox_occluder", 6, 5, 7)
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.5, 0.0]) * P))
    sc.configure()
    integ = psdr_cuda.DirectIntegrator(1, 1)
    img = integ.renderD(sc, 0)
    ek.forward(P, free_graph=True)
    out["d_fwd_img"], out["d_fwd_grad"] = img.numpy(), ek.gradient(img).numpy()
    # guiding grid, then a guided renderD
    w = integ.preprocess_secondary_edges(sc, 0, np.array([64, 4, 4, 2]), 2)
    out["guide_cmf"] = w.m_distrb.m_cmf.cpu().numpy()
    # renderD + backward: albedo texels and vertex positions
    sc = scene("cbox", 6, 4, 4)
    refl = sc.param_map["BSDF[0]"].reflectance
    ek.set_requires_gradient(refl.data)
    mesh = sc.param_map["Mesh[0]"]
    v = Vector3fD(ek.detach(mesh.vertex_positions))
    ek.set_requires_gradient(v)
    mesh.vertex_positions = v
    sc.configure()
    img = psdr_cuda.DirectIntegrator(1, 1).renderD(sc, 0)
    wgt = torch.linspace(0.5, 1.5, img.t.numel(), device=img.t.device).reshape(img.t.shape)
    ek.backward(FloatD._wrap((wgt * (img.t - 0.3) ** 2).sum().reshape(1)))
    out["d_rev_img"] = img.numpy()
    out["g_refl"], out["g_vert"] = ek.gradient(refl.data).numpy(), ek.gradient(v).numpy()
    # round 5, PathTracer on a two-level scene at a size where the library picks its wavefront launches by itself (2^20 slots per rank at two ranks):
    # geometry tangents through the traced wavefront with dual-number stages, then renderD + backward whose reverse call reuses the primal render's records
    sc = scene("cbox_bunny", 32, 0, 0, res=256)
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([0.5, 1.0, 0.0]) * P))
    sc.configure()
    pt = psdr_cuda.PathTracer(3)
    img = pt.renderD(sc, 0)
    ek.forward(P, free_graph=True)
    out["pt_fwd_img"], out["pt_fwd_grad"] = img.numpy(), ek.gradient(img).numpy()
    sc = scene("cbox_bunny", 32, 0, 0, res=256)
    mesh = sc.param_map["Mesh[1]"]
    v = Vector3fD(ek.detach(mesh.vertex_positions))
    ek.set_requires_gradient(v)
    mesh.vertex_positions = v
    sc.configure()
    img = pt.renderD(sc, 0)
    ek.backward(FloatD._wrap((img.t * img.t).sum().reshape(1)))
    out["pt_rev_img"], out["pt_g_vert"] = img.numpy(), ek.gradient(v).numpy()
    out["pt_rev_rays"] = np.array([pt.last_counters[0]])               # 0: the reverse call ran its adjoint kernel only (PSDR_FLAG_KEEP_RECORDS)
    return out


def main():
    """Entry point of one rank (python tests/test_multi_gpu_exec_gpu.py <out.npz> under torch.distributed.run)."""
    for p in ("psdr-cuda_amd", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                       # both ranks on the one GPU
    dist.init_process_group("gloo")
    out = run_sequence()
    out["world"] = np.array([dist.get_world_size()])
    if dist.get_rank() == 0:
        np.savez(sys.argv[1], **out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_single_process_results(tmp_path):
    from helpers import rel_l2
    path = str(tmp_path / "ranks.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 1000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.abspath(__file__), path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(path)
    assert int(got["world"][0]) == 2
    ref = run_sequence()
    for k, v in ref.items():
        assert got[k].shape == v.shape, k
        if k == "pt_rev_rays":
            continue
        tol = 1e-4 if k.startswith("g_") or k in ("pt_fwd_grad", "pt_g_vert") else 2e-5          # same samples, different grouping of the fp32 sums / atomics (3e-7 between two runs of one process since the table chain is reproducible: tests/test_tables_native_gpu.py)
        assert rel_l2(got[k], v) < tol, (k, rel_l2(got[k], v))
    assert np.abs(ref["d_fwd_grad"]).max() > 0 and np.abs(ref["g_vert"]).max() > 0 and np.abs(ref["g_refl"]).min() > 0
    assert np.abs(ref["pt_fwd_grad"]).max() > 0 and np.abs(ref["pt_g_vert"]).max() > 0
    assert int(ref["pt_rev_rays"][0]) == 0 and int(got["pt_rev_rays"][0]) == 0          # one process and each of two ranks: the reverse call reused the primal render's records


if __name__ == "__main__":
    main()


def test_bench_multi_rank_path_executes_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with the developer switch
    PSDR_BENCH_ONE_GPU=1: both ranks on cuda:0 over gloo.  Not a measurement -- it EXECUTES the script's multi-rank code: sharded spp, the
    (deferred) image all-reduce of renderC, the [image || derivative image] all-reduce, the max-over-ranks clock, rank 0's JSON line."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PSDR_BENCH_ONE_GPU="1")
    port = 29700 + os.getpid() % 1000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-pmc",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["global_spp"] == 128
    assert d["value"] > 0 and d["config"]["allreduce_bytes_per_step"] == 3 * 512 * 512 * 3 * 4
