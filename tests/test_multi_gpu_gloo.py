"""N>1 path on CPU (world_size 2, gloo): the spp sharding + single all-reduce protocol of SURVEY 8(e).
Each rank evaluates ITS shard of the sample slots (the oracle stands in for the GPU renderer here --
the shard bookkeeping, option block and collective are the code under test) and the all-reduced
image must equal the single-process render."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("psdr-cuda_amd", "oracle", "tests"):
        sys.path.insert(0, os.path.join(root, p))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import psdr_cuda
    from helpers import load_scene, tangents_wrt
    sc, P = load_scene("cbox_occluder", res=16, spp=6, sppe=4, sppse=5, translate=(1, (1.0, 0.0, 0.0)))
    tb = sc.tables(0)
    integ = psdr_cuda.DirectIntegrator(1, 1)
    opts = integ._opts(sc, with_edges=True)          # picks this rank's shard from torch.distributed
    assert (opts.spp_end - opts.spp_begin) == 3 and opts.spp == 6
    img, dimg = oracle.render(tb, opts, mode=1, tangents=tangents_wrt(tb, P), nthreads=2)
    buf = torch.from_numpy(np.concatenate([img.reshape(-1), dimg.reshape(-1)]))   # [image || derivative image]
    dist.all_reduce(buf)                             # the ONE collective of a render call
    if rank == 0:
        np.save(out_path, buf.numpy())
    dist.destroy_process_group()


def test_two_rank_spp_sharding_equals_single_process(tmp_path):
    import oracle
    from helpers import load_scene, rel_l2, tangents_wrt
    from psdr_cuda import _abi
    out = str(tmp_path / "sharded.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sc, P = load_scene("cbox_occluder", res=16, spp=6, sppe=4, sppse=5, translate=(1, (1.0, 0.0, 0.0)))
    tb = sc.tables(0)
    img, dimg = oracle.render(tb, _abi.make_opts(spp=6, sppe=4, sppse=5), mode=1, tangents=tangents_wrt(tb, P))
    n = img.size
    assert rel_l2(got[:n], img.reshape(-1)) < 1e-6
    assert rel_l2(got[n:], dimg.reshape(-1)) < 1e-5 and np.abs(dimg).max() > 0
