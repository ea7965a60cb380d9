"""psdr_scene_set_option: the developer switches of a handle (the library reads no environment variable)."""
import os

import numpy as np
import pytest

from helpers import GpuScene, load_scene, rel_l2
from psdr_cuda import _abi

pytestmark = pytest.mark.gpu


def test_unknown_option_fails_and_known_ones_switch_strategies():
    sc, _ = load_scene("cbox_bunny", res=64, spp=16)
    tb = sc.tables(0)
    g = GpuScene(tb)
    assert g.lib.psdr_scene_set_option(g.h, b"no_such_option", 1.0) != 0
    assert b"unknown option" in g.lib.psdr_last_error()
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16, flags=_abi.FLAG_WAVEFRONT)
    stats = _abi.scene_stats(g.h)
    assert stats.get("n_blas", 0) == 1                                  # the two-level tree of a room with one object
    a = g.render_c(o)
    for opts in ({"wf_traced": 0}, {"wf_traced": 0, "wf_binned": 0}, {"two_level": 0}):
        g2 = GpuScene(tb, options=opts)
        if "two_level" in opts:
            assert _abi.scene_stats(g2.h).get("n_blas", 0) == 0
        b = g2.render_c(o)
        bad = np.abs(a - b).max(1) > 1e-5 * (1.0 + np.abs(a).max(1))
        assert bad.mean() < 2e-3 and rel_l2(b[~bad], a[~bad]) < 1e-4, (opts, bad.mean())


def test_every_option_the_header_names_is_accepted():
    """include/psdr_hip.h lists the developer options of a handle; psdr_scene_set_option must know each of them (two were missing from the list)."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "psdr_hip.h")).read()
    block = hdr[hdr.index("Developer options of a handle"):hdr.index("int psdr_scene_set_option")]
    names = set(re.findall(r"\b([a-z][a-z0-9]*(?:_[a-z0-9]+)+|wide|probe|logd)\b", block)) - {"psdr_scene_set_option", "psdr_bvh_build", "per_cu", "psdr_hip"}
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "psdr-cuda_amd", "csrc", "psdr_hip.hip")).read()
    known = set(re.findall(r'n == "([a-z0-9_]+)"', src[src.index("int psdr_scene_set_option"):src.index("int psdr_scene_destroy")]))
    assert known <= names | {"emitter_layout"}, sorted(known - names)           # every option the library accepts is documented
    sc, _ = load_scene("cbox", res=16, spp=2)
    g = GpuScene(sc.tables(0))
    for n in sorted(known):
        assert g.lib.psdr_scene_set_option(g.h, n.encode(), 1.0) == 0, n
    for n in sorted(names & known):
        assert g.lib.psdr_scene_set_option(g.h, n.encode(), 0.0) == 0, n


def test_library_reads_no_environment_variable():
    """No PSDR_* name survives in the binary (17 getenv knobs lived in the hot host path)."""
    import subprocess
    out = subprocess.run(["strings", "-a", _abi.HIP_LIB_PATH], capture_output=True, text=True).stdout
    names = [l for l in out.splitlines() if l.startswith("PSDR_") and l.replace("_", "").isalnum() and l.isupper()]
    assert names == [], names


@pytest.mark.parametrize("kind", ["path3", "direct11"])
def test_chunked_and_sharded_launches_on_a_two_level_scene(kind):
    """The launches that keep per-slot state between kernels -- traced wavefront streams, the per-path records of a split reverse launch, the hit rows of
    probe / trace / final launches -- run chunk by chunk on large launches; `chunk_log2` makes them do so on a small scene: the same image, derivative
    image and gradients as one chunk (same samples; the order of the float adds only).  And the sample shards of a multi-GPU job add up to the full render."""
    from helpers import random_tangents, dot_tables, load_scene as ls
    sc, P = ls("cbox_bunny", res=96, spp=16, sppe=8, sppse=8, translate=(1, (1.0, 0.3, 0.0)))
    tb = sc.tables(0)
    kw = dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3) if kind == "path3" else dict(bsdf_samples=1, light_samples=1, sppe=8, sppse=8)
    o = _abi.make_opts(spp=16, **kw)
    adj = np.random.default_rng(5).random((96 * 96, 3)).astype(np.float32)
    from helpers import tangents_wrt
    tan = tangents_wrt(tb, P) if kind == "direct11" else {"texels": random_tangents(tb, ["texels"], seed=1)["texels"]}
    tan_geo = tangents_wrt(tb, P)                   # path3: the PathTracer's geometry duals -- the traced wavefront with dual-number stages (round 5) -- chunk by chunk too
    want = ["tri_info", "texels"] + (["sec_edge", "prim_edge"] if kind == "direct11" else [])
    res = {}
    for name, opts in (("one", {}), ("chunks", {"chunk_log2": 14})):            # 147 456 slots: nine chunks of 2^14
        g = GpuScene(tb, options=opts)
        res[name] = (g.render_c(o), g.render_d_fwd(o, [tan])[1][0], g.render_d_rev(o, adj, want=want, with_image=False)[1], g.counters()[0],
                     g.render_d_fwd(_abi.make_opts(spp=16, flags=_abi.FLAG_WAVEFRONT, **kw), [tan_geo])[1][0] if kind == "path3" else None)
    a, b = res["one"], res["chunks"]
    assert rel_l2(b[0], a[0]) < 1e-5 and rel_l2(b[1], a[1]) < 1e-4
    if kind == "path3":
        assert np.abs(a[4]).max() > 0 and rel_l2(b[4], a[4]) < 1e-4, rel_l2(b[4], a[4])
    for k in want:
        assert rel_l2(b[2][k], a[2][k]) < 2e-4, (k, rel_l2(b[2][k], a[2][k]))
    # shards: samples [0, 5), [5, 6), [6, 16) of every pixel (and of the edge samplers) add up to the full launch
    g = GpuScene(tb)
    full = g.render_c(o)
    parts = 0
    for r in ((0, 5), (5, 6), (6, 16)):
        er = (r[0] // 2, r[1] // 2)
        parts = parts + g.render_c(_abi.make_opts(spp=16, spp_range=r, **({**kw, "sppe_range": er, "sppse_range": er} if kind == "direct11" else kw)))
    assert rel_l2(parts, full) < 1e-5


def test_c4_shard_as_one_chunk_equals_two_chunks():
    """The traced wavefront and the split reverse launch serve 2^26 slots per chunk: one GPU's share of C4 (cbox_bunny 1024^2, 64 of 512 spp) is ONE
    chunk.  The same launches as two chunks of 2^25 (the previous default) give the same image, ray count and gradients -- same samples, the order of the
    float adds only."""
    sc, _ = load_scene("cbox_bunny", res=1024, spp=512, sppe=0, sppse=0)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=512, spp_range=(0, 64), integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    adj = np.random.default_rng(7).random((1024 * 1024, 3)).astype(np.float32)
    res = {}
    for name, opts in (("one", {}), ("two", {"chunk_log2": 25}), ("grid16", {"blocks_per_cu": 16})):
        g = GpuScene(tb, options=opts)
        img = g.render_c(o)
        rays = g.counters()[0]
        res[name] = (img, rays, g.render_d_rev(o, adj, want=["tri_info", "texels"], with_image=False)[1])
        del g
    a = res["one"]
    for other in ("two", "grid16"):          # grid16: the stage kernels' grid of 16 workgroups per CU (the default of a launch this size is 40)
        b = res[other]
        assert a[1] == b[1] and rel_l2(b[0], a[0]) < 1e-5, other
        for k in ("tri_info", "texels"):
            assert rel_l2(b[2][k], a[2][k]) < 2e-4, (other, k, rel_l2(b[2][k], a[2][k]))


@pytest.mark.parametrize("scene", ["cbox", "cbox_occluder", "cbox_rough", "cbox_uv"])
def test_occluder_rows_do_not_change_a_single_sample(scene):
    """Round 6: on a scene without a tree a light ray tests only the rows of the primitive table that can lie between its vertex and the emitter sample
    (psdr_device.h closest_hit MASKED, psdr_bvh_build.h tiny_occluder_rows; tests/test_occluder_rows.py checks the table against brute force).  An extra
    test never changes a closest hit, so with the table (default) and with all-ones rows (`occ_rows` 0) the SAME kernel returns the same image bit for bit --
    renderC, forward mode and reverse mode, DirectIntegrator and PathTracer -- and traces the same number of rays."""
    import torch
    sc, _ = load_scene(scene, res=48, spp=16)
    tb = sc.tables(0)
    ga, gb = GpuScene(tb), GpuScene(tb, options={"occ_rows": 0})
    for kw in (dict(integrator=_abi.INTEGRATOR_PATH, max_depth=4), dict(bsdf_samples=2, light_samples=2), dict(bsdf_samples=0, light_samples=1)):
        o = _abi.make_opts(spp=16, rng_offset=(9, 0, 0), **kw)
        a = ga.render_c(o); ra = ga.counters()[0]
        b = gb.render_c(o); rb = gb.counters()[0]
        assert ra == rb and np.array_equal(a, b), (scene, kw, rel_l2(a, b))
    o = _abi.make_opts(spp=16, integrator=_abi.INTEGRATOR_PATH, max_depth=3)
    t = {"texels": torch.rand(tb["texels"].shape, generator=torch.Generator().manual_seed(2))}
    (ia, da), (ib, db) = ga.render_d_fwd(o, [t]), gb.render_d_fwd(o, [t])
    assert np.array_equal(ia, ib) and np.array_equal(da[0], db[0])
    adj = np.random.default_rng(5).random((48 * 48, 3)).astype(np.float32)
    xa, xb = ga.render_d_rev(o, adj, want=["texels", "tri_info"])[1], gb.render_d_rev(o, adj, want=["texels", "tri_info"])[1]
    for k in xa:
        assert rel_l2(xb[k], xa[k]) < 1e-5, k                      # (float atomics: the order of the adds)
