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


def _solo_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("psdr-cuda_amd", "oracle", "tests"):
        sys.path.insert(0, os.path.join(root, p))
    from datetime import timedelta
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(backend="gloo", timeout=timedelta(minutes=5))        # bench.py's host-side barrier group
    import psdr_cuda
    from psdr_cuda import integrator as I
    from helpers import load_scene
    sc, _ = load_scene("cbox", res=8, spp=6)
    integ = psdr_cuda.PathTracer(3)
    sharded = integ._opts(sc, with_edges=False)
    res = {"sharded": (sharded.spp_begin, sharded.spp_end), "dist_outside": I._dist() is not None}
    if rank == 0:
        # what bench.py does with rank 0's side blocks at N > 1: the whole sample range, no collective -- the other rank is NOT in here
        with I.solo():
            o = integ._opts(sc, with_edges=False)
            res["solo"] = (o.spp_begin, o.spp_end)
            res["dist_inside"] = I._dist() is not None
            with I.solo():                                                       # nests
                pass
            res["dist_nested_exit"] = I._dist() is not None
            node_solo = I._RenderNode(integ, sc, 0, None, o, None)           # what renderD creates inside the block ...
        res["dist_after"] = I._dist() is not None
        # ... and evaluates later, OUTSIDE it (the lazy image, its backward): the decision travels with the node
        node_job = I._RenderNode(integ, sc, 0, None, sharded, None)
        with I._decided(node_solo.collective):
            res["deferred_solo_node"] = I._dist() is not None
        with I.solo():
            with I._decided(node_job.collective):
                res["deferred_job_node_inside_solo"] = I._dist() is not None
        I.force_collectives(True); res["forced_in_solo"] = [I._dist() is not None]
        with I.solo():
            res["forced_in_solo"].append(I._dist() is not None)
        I.force_collectives(False)
    dist.barrier(group=side)                                                     # the others wait here meanwhile
    if rank == 0:
        import json
        json.dump(res, open(out_path, "w"))
    dist.destroy_process_group()


def test_solo_block_runs_one_rank_without_sharding_or_collectives(tmp_path):
    """bench.py at N > 1: rank 0 runs its counter passes and single-GPU side blocks inside `psdr_cuda.integrator.solo()` while the
    other ranks wait at a gloo barrier -- inside the block a render call covers the whole sample range and issues no collective."""
    import json
    out = str(tmp_path / "solo.json")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_solo_worker, args=(2, port, out), nprocs=2, join=True)
    r = json.load(open(out))
    assert r["sharded"] == [0, 3] and r["dist_outside"] and r["dist_after"]
    assert r["solo"] == [0, 6] and not r["dist_inside"] and not r["dist_nested_exit"]
    assert r["deferred_solo_node"] is False and r["deferred_job_node_inside_solo"] is True and r["forced_in_solo"] == [True, False]


def test_shard_ranges_at_eight_ranks():
    """The spp shards of the node the north star names (8 ranks): BASELINE configs[3]'s 512 spp are 64 per rank; an odd count and a count below the rank count still
    partition [0, spp) into contiguous ranges in rank order (some EMPTY: such a rank launches nothing and only enters the collectives)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "psdr-cuda_amd"))
    from psdr_cuda.integrator import shard_range
    assert [shard_range(512, r, 8) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    for spp in (13, 7, 3, 1, 0, 8, 9):
        rs = [shard_range(spp, r, 8) for r in range(8)]
        assert rs[0][0] == 0 and rs[-1][1] == spp and all(a[1] == b[0] for a, b in zip(rs, rs[1:])), (spp, rs)
        sizes = [b - a for a, b in rs]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True), (spp, sizes)
