"""The reference's scenario table replayed on the reference's OWN files.

examples/config.py:45-167 lists eight validation scenarios; examples/psdr_test.py + run_test.py run each as  orig (renderC passes) / AD (per pass: P = FloatD(0),
the parameter applied, configure, guiding grid at pass 0, renderD, enoki.forward) / FD (central differences of renderC).  tests/ref_harness.py is this
repository's counterpart of that harness; tests/golden/refdata/ holds the reference's scene files and the objects its snapshot ships (tree0.obj 24 130 faces,
plane / emitter quads, the Cornell-box quads) as input fixtures -- `bunny_low.obj` and `ballroom_1k.exr` were fixtures already; `bunny.obj` is absent from the
snapshot and `bunny_low.obj` stands in where a file names it.

Every scenario runs THROUGH THE DROP-IN SURFACE on the GPU at the file's own resolution and the table's own sample counts (spp / sppe / sppse, bsdf / light
samples, guiding resolution 40000 x 5 x 5 x 2 and rounds, meshes without edges); the derivative image of every pass is compared with the CPU oracle on the same
sample streams, the same tables and the same guiding grid.  npass is capped (the table's 20-100 passes only reduce variance).  Two scenarios are also
validated the reference's own way, AD against central finite differences at the table's eps.
"""
import numpy as np
import pytest

import oracle
import ref_harness as H
from helpers import isolated_pixels_unbiased, rel_l2, tangents_wrt

pytestmark = pytest.mark.gpu

NPASS = 2          # passes per scenario (<= 4: the oracle replays every pass on the host cores)


def _opts_of_pass(integ, sc, tb, with_edges):
    o = integ._opts(sc, with_edges=with_edges)
    if with_edges and not (tb["num_prim_edges"] > 0):          # Integrator.renderD: no primary edge in view -> no primary-edge launch
        o.sppe = o.sppe_begin = o.sppe_end = 0
    return o


@pytest.mark.parametrize("name", list(H.SCENARIOS))
def test_scenario_of_the_reference_table_matches_the_oracle(name, tmp_path):
    args = H.SCENARIOS[name]
    integ = H.make_integrator(args)
    sc = H.load(args, tmp_path)
    W, Hh = sc.opts.width, sc.opts.height
    passes = []
    if "AD" not in args:
        # orig only (cbox_mutie: two area emitters): renderC passes against the oracle
        sc.configure()
        img = H.run_orig(integ, sc, NPASS, on_pass=lambda i, s, it: passes.append((s.tables(0), _opts_of_pass(it, s, s.tables(0), False))))
        ref = np.mean([oracle.render(tb, o).astype(np.float64) for tb, o in passes], axis=0)
        bad = np.abs(img - ref).max(1) > 1e-3 * (1.0 + np.abs(ref).max(1))
        print("%s orig %dx%d, %d passes: rel-L2 %.2e, isolated pixels %d" % (name, W, Hh, NPASS, rel_l2(img, ref), bad.sum()))
        assert np.isfinite(img).all() and ref.mean() > 0.01
        assert bad.mean() < 2e-3 and rel_l2(img[~bad], ref[~bad]) < 1e-4
        isolated_pixels_unbiased(img, ref, bad, name)
        return
    ad = args["AD"]
    state = {}

    def on_pass(i, s, it, P):
        tb = s.tables(0)
        passes.append((tb, _opts_of_pass(it, s, tb, True), tangents_wrt(tb, P), it._guide.get(0) if hasattr(it, "_guide") else None))
        state["edges"] = (tb["num_prim_edges"], tb["num_sec_edges"])
    d = H.run_ad(integ, sc, ad, NPASS, on_pass=on_pass)
    assert (sc.opts.spp, sc.opts.sppe, sc.opts.sppse) == (ad["spp"], ad["sppe"], ad["sppse"])            # the table's sample counts, not the file's
    if "guide" in ad:
        assert passes[0][3] is not None and list(passes[0][3][0]) == ad["guide"]["reso"][:3]               # 40000 x 5 x 5 cells, built with the table's rounds
    for m in ad.get("no_edge", []):
        assert sc.m_meshes[m].enable_edges is False
    ref = np.mean([oracle.render(tb, o, mode=1, tangents=tan, guide=g)[1].astype(np.float64) for tb, o, tan, g in passes], axis=0)
    assert np.isfinite(d).all() and np.abs(ref).max() > 0
    # boundary terms on meshes of thousands of small faces: isolated samples resolve an epsilon test the other way in two fp32 evaluations
    scale = np.abs(ref).max(1) + 1e-2 * np.abs(ref).max()
    bad = np.abs(d - ref).max(1) > 1e-3 * (scale + 1e-30) + 1e-7
    print("%s AD %dx%d spp/sppe/sppse = %d/%d/%d, %d passes, edges (primary, secondary) = %s: derivative image rel-L2 %.2e, outside %d isolated pixels %.2e" % (
        name, W, Hh, ad["spp"], ad["sppe"], ad["sppse"], NPASS, state["edges"], rel_l2(d, ref), bad.sum(), rel_l2(d[~bad], ref[~bad])))
    # tree: 23 677 boundary edges of centimetre-sized leaves, 64 boundary samples per pixel and pass -- a few per cent of the pixels hold ONE sample whose
    # epsilon test (the edge ray grazing a neighbouring leaf) the two fp32 evaluations resolve differently (measured 3.4 %; outside them 3e-5); what keeps that
    # honest is the net below: their signed errors must cancel
    assert bad.mean() < (6e-2 if name == "tree" else 1e-2), bad.mean()
    assert rel_l2(d[~bad], ref[~bad]) < 2e-3
    isolated_pixels_unbiased(d, ref, bad, name, bias_bound=5e-3)


@pytest.mark.parametrize("name", ["bunny_silhouette", "bunny_env_1"])
def test_scenario_ad_against_finite_differences_at_the_table_eps(name, tmp_path):
    """The reference's own validation (run_test.py run_ad / run_fd): the AD derivative image against central differences of renderC at the table's eps, both
    averaged over passes and compared on 16 x 16 pixel blocks (a coverage image moves by whole samples)."""
    args = H.SCENARIOS[name]
    ad, fdc = args["AD"], args["FD"]
    integ = H.make_integrator(args)
    sc = H.load(args, tmp_path)
    W, Hh = sc.opts.width, sc.opts.height
    npass = min(fdc["npass"], 24)
    d = H.run_ad(integ, sc, ad, npass).reshape(Hh, W, 3)
    fd = H.run_fd(H.make_integrator(args), args, tmp_path, npass).reshape(Hh, W, 3)
    bh, bw = Hh // 16, W // 16
    blk = lambda a: a[:bh * 16, :bw * 16].reshape(bh, 16, bw, 16, 3).sum(axis=(1, 3))
    err = np.linalg.norm(blk(d) - blk(fd)) / np.linalg.norm(blk(fd))
    print("%s: AD sum %.4g, FD sum %.4g (eps %g, %d passes), 16x16-block rel-L2 %.3f" % (name, d.sum(), fd.sum(), fdc["eps"], npass, err))
    assert np.isfinite(d).all() and np.abs(fd).max() > 0
    assert err < 0.1, err                                            # measured 0.027 (bunny_silhouette), 0.031 (bunny_env_1)
