"""RCCL EXECUTED on the one GPU of the test box: a process group with backend `nccl` (= RCCL on ROCm) at world size 1, and
PSDR_FORCE_COLLECTIVES=1 so that psdr_cuda/integrator.py issues every collective of a render call although there is one rank -- the
asynchronous image all-reduce of renderC joined through an NCCL work handle (_ReducingImage), the [image || derivative images]
all-reduce of renderD + enoki.forward, the flat gradient-buffer all-reduce of renderD + enoki.backward, the guiding-mass all-reduce.
The results must equal the same calls without a process group (a one-rank sum is the identity); the RCCL version that ran is written to
gpurun_out/rccl_single_rank.json (builder-kept copy: profiles/).  Multi-rank correctness is tests/test_multi_gpu_gloo.py (CPU) and
tests/test_multi_gpu_exec_gpu.py (two ranks on one GPU over gloo); `bench.py --gpus 2` without a launcher is covered here too."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    for p in ("psdr-cuda_amd", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import torch
    import torch.distributed as dist
    from test_multi_gpu_exec_gpu import run_sequence
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.arange(8, device="cuda", dtype=torch.float32)
    dist.all_reduce(t)                               # a first collective: creates the RCCL communicator
    torch.cuda.synchronize()
    assert t.cpu().tolist() == list(range(8))
    import psdr_cuda.integrator as integ
    assert integ._dist() is None                      # one rank: no collective unless asked for
    integ.force_collectives(os.environ.get("PSDR_FORCE_COLLECTIVES") == "1")          # the test driver's switch; the package itself reads no environment variable
    assert integ._dist() is dist                      # the render calls will issue their collectives
    calls = {"all_reduce": 0, "async": 0}
    real = dist.all_reduce

    def counting(tensor, *a, **kw):
        calls["all_reduce"] += 1
        calls["async"] += 1 if kw.get("async_op") else 0
        assert tensor.is_cuda
        return real(tensor, *a, **kw)
    dist.all_reduce = counting
    out = run_sequence()
    dist.all_reduce = real
    v = torch.cuda.nccl.version()
    out["meta"] = np.array(json.dumps({"rccl_version": ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v), "backend": dist.get_backend(),
                                       "world_size": dist.get_world_size(), "all_reduce_calls": calls["all_reduce"], "async_all_reduce_calls": calls["async"],
                                       "device": torch.cuda.get_device_name(0)}))
    np.savez(sys.argv[1], **out)
    dist.barrier()
    dist.destroy_process_group()


def test_render_calls_over_rccl_at_world_size_one(tmp_path):
    from helpers import rel_l2
    from test_multi_gpu_exec_gpu import run_sequence
    path = str(tmp_path / "rccl.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PSDR_FORCE_COLLECTIVES="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), path, str(29800 + os.getpid() % 1000)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(path)
    meta = json.loads(str(got["meta"]))
    assert meta["backend"] == "nccl" and meta["world_size"] == 1 and meta["rccl_version"] not in ("", "None")
    # renderC x2 (async), renderD+forward, guiding mass, renderD+backward (primal image + gradient buffer): at least six collectives, two of them asynchronous
    assert meta["all_reduce_calls"] >= 6 and meta["async_all_reduce_calls"] >= 2, meta
    ref = run_sequence()
    for k, v in ref.items():
        assert got[k].shape == v.shape, k
        # same samples in two processes: the fp32 atomics land in a different order
        tol = 1e-4 if k.startswith("g_") or k in ("pt_fwd_grad", "pt_g_vert") else 2e-5
        assert rel_l2(got[k], v) < tol, (k, rel_l2(got[k], v))
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "rccl_single_rank.json"), "w") as f:
        json.dump(meta, f)
    print("RCCL", meta)


def test_bench_c4_over_rccl_at_world_size_one():
    """bench.py --config c4 --gpus 1 with PSDR_FORCE_COLLECTIVES=1: the strong-scaling workload of BASELINE configs[3] with its three all-reduces per step
    going through RCCL (one rank), reduced size."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PSDR_FORCE_COLLECTIVES="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--gpus", "1", "--steps", "2", "--warmup", "1", "--res", "256", "--spp", "16",
                        "--no-pmc", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["rccl_version"] and d["config"]["collectives_forced"] is True
    assert d["value"] > 0 and d["grad_check"]["finite"]


def test_bench_relaunches_itself_for_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher on the command line: the script re-executes itself under torch.distributed.run (bench.py
    relaunch_under_torchrun); PSDR_BENCH_ONE_GPU=1 lets the two ranks share cuda:0 over gloo."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PSDR_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["value"] > 0
    # the N > 1 line answers the north star -- rank 0 ran the counter passes and the single-GPU side blocks while rank 1 waited,
    # and every rank took part in the strong-scaling block of BASELINE configs[3]
    assert d["roofline"]["frac"] is not None and 0.05 < d["roofline"]["frac"] < 1.0, d["roofline"]
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0 and d["grad_rel_l2"]["rel_l2"] < 1e-3
    ts = d["tree_scenes"]
    assert "error" not in ts and "pmc_error" not in ts, ts
    for k in ("c4_shard_path3_renderC", "c4_shard_path3_rev", "c4_shard_path3_fwd_geo", "c5_path3_renderC", "c5_path3_fwd_geo", "c3_direct_fwd3"):
        assert ts[k]["ms"] > 0 and ts[k]["dominant_kernel"]["valu_issue_frac"] > 0, (k, ts.get(k))
    assert 0.0 < ts["c4_shard_path3_renderC"]["hbm"]["hbm_measured_frac"] < 1.0
    c4 = d["c4_strong"]
    assert "error" not in c4, c4
    assert c4["world_size"] == 2 and c4["scaling"] == "strong" and c4["ms_per_step"] > 0 and c4["allreduce_bytes_per_step"] > 2 * 1024 * 1024 * 12 and c4["grad_finite"]
    assert 0.0 < c4["wavefront_hbm"]["kernels"]["all_wavefront_kernels"]["hbm_measured_frac"] < 1.0


if __name__ == "__main__":
    main()
