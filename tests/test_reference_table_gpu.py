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


@pytest.mark.parametrize("name", ["bunny_silhouette", "bunny_env_1", "cbox_MIS"])
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
    assert err < 0.1, err                                            # measured 0.027 (bunny_silhouette), 0.031 (bunny_env_1), 0.006 (cbox_MIS)


def _ad_fd_blocks(args, tmp_path, face_normals, mesh, res, npass_ad, npass_fd, nblk):
    """AD and central-difference derivative images of a scenario row as nblk x nblk block sums, the moving mesh with face normals or the file's smooth ones"""
    from enoki.cuda_autodiff import Float32 as FloatD

    def load():
        sc = H.load(args, tmp_path, res)
        sc.param_map["Mesh[%d]" % mesh].use_face_normals = face_normals
        return sc
    sc = load()
    W, Hh = sc.opts.width, sc.opts.height
    d = H.run_ad(H.make_integrator(args), sc, args["AD"], npass_ad).reshape(Hh, W, 3)
    integ, scs, eps = H.make_integrator(args), [], args["FD"]["eps"]
    for sgn in (-1.0, 1.0):
        s = load()
        s.opts.sppe, s.opts.sppse = 0, 0
        H.apply_parameter(s, args["AD"], FloatD(sgn * eps), {})
        s.configure()
        scs.append(s)
    fd = 0
    for _ in range(npass_fd):
        fd = fd + (integ.renderC(scs[1]).numpy().astype(np.float64) - integ.renderC(scs[0]).numpy().astype(np.float64))
    fd = (fd / (2 * eps * npass_fd)).reshape(Hh, W, 3)
    bh, bw = Hh // nblk, W // nblk
    blk = lambda a: a[:bh * nblk, :bw * nblk].reshape(nblk, bh, nblk, bw, 3).sum(axis=(1, 3, 4))
    return blk(d), blk(fd)


@pytest.mark.parametrize("name", ["cbox_bunny_translate", "bunny_env_2"])
def test_moving_mesh_ad_meets_finite_differences_with_face_normals(name, tmp_path):
    """Why `tree` and `bunny_env_2` are NOT in the AD-vs-FD list above, and what can be said instead (round 6, tools/r06_fd_normals_probe.py, tools/r06_fd_terms_probe.py).
    `tree`: the table's AD row has spp = sppe = 0 -- the secondary-edge term alone -- while central differences see the whole derivative: not comparable.
    `bunny_env_2` (the bunny rotated under an environment map) and any moving SMOOTH-shaded coarse mesh: AD and FD agree on the floor (shadow boundaries: the
    secondary-edge term) and DISAGREE on the mesh's own pixels, stably in the pass count and in eps (measured: block rel-L2 0.25 for the translated bunny of
    cbox_bunny.xml, 0.40 rotated, 0.8 under the environment map).  With interpolated normals the lighting cut-off of a surface point follows its FACE's horizon,
    so the shading jumps across every mesh edge, and those jumps move across the film with the mesh; the reference puts non-silhouette edges into the primary-edge
    table only for face-normal meshes (perspective.cpp:57-66), so for smooth meshes its estimator -- restated here -- has no boundary term for them.  That is a
    property of the reference's estimator, not of this implementation: the SAME kernels with face normals on the moving mesh (every visible edge in the table)
    meet central differences on the mesh's own pixels.  Asserted here; the smooth-normal gap is printed beside it."""
    if name == "bunny_env_2":
        args, mesh, res, nblk = dict(H.SCENARIOS[name]), 0, None, 5
        args["AD"] = dict(args["AD"], spp=8, sppe=8)
    else:
        args = dict(test_type="direct", scene_file="cbox_bunny.xml", bsdf_samples=1, light_samples=1,
                    AD=dict(type="mesh_transform", Mesh_ID=[1], Mesh_dir=[[1., 0., 0.]], spp=16, sppe=16, sppse=64), FD=dict(npass=64, eps=0.2))
        mesh, res, nblk = 1, 128, 8
    out = {}
    for face in (True, False):
        bd, bf = _ad_fd_blocks(args, tmp_path, face, mesh, res, 48, 96, nblk)
        out[face] = float(np.linalg.norm(bd - bf) / np.linalg.norm(bf))
    print("%s: AD vs FD over %dx%d blocks -- face normals on the moving mesh %.3f, the file's smooth normals %.3f" % (name, nblk, nblk, out[True], out[False]))
    assert out[True] < (0.16 if name == "bunny_env_2" else 0.12), out          # measured 0.051 (cbox_bunny), 0.105 (bunny_env_2: 48 AD / 96 FD passes under a high-dynamic-range map)
    assert out[False] > out[True], out          # (the estimator's missing term: if this ever closes, the sentence above is out of date)
