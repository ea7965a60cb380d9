"""The two oracles against each other, against the product's table builder, and the GPU against both.

oracle/torch_oracle.py is an independent restatement (torch, fp64, its OWN table chain from the raw scene inputs,
brute-force hits, forward-mode AD); oracle/psdr_oracle.cpp is the scalar C++ one that reads the tables the product
built.  Neither can be pinned on reference output (the reference holds no vectors and cannot be built here), so they
are pinned on each other:
  (i)   the product's tables (psdr_cuda/scene.py, fp32 torch ops) == the independent tables (process_mesh, edge
        topology, coplanar filter, camera matrices, primary-edge list, emitter tables, distributions), and the JVP
        tangent tables of a mesh translation likewise;
  (ii)  C++ oracle (fp64, the reference's literal forms) == torch oracle on identical tables: image and derivative
        image, term by term;
  (iii) committed fixtures of the torch oracle (tests/golden/torch_*.npz) are reproduced by the C++ oracle and the GPU.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_torch_golden as mtg  # noqa: E402
import oracle  # noqa: E402
import torch_oracle as to  # noqa: E402
from helpers import load_scene, rel_l2, tangents_wrt  # noqa: E402
from psdr_cuda import _abi  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sorted_rows(a):
    a = a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    return a[np.lexsort(np.round(a, 4).T[::-1])]


def _np(a):
    return a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)


def _close(a, b, tol):
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("scene", ["cbox", "cbox_occluder", "cbox_bunny", "bunny_light"])
def test_product_tables_equal_the_independent_tables(scene):
    sc, _ = load_scene(scene, res=24, spp=1, sppe=1, sppse=1)
    tb = sc.tables(0)
    tt = to.build_tables(to.scene_inputs(sc))
    _close(tb["tri_info"][:, :22], tt["tri_info"][:, :22], 2e-6)            # process_mesh: fp32 against fp64
    assert torch.equal(tb["tri_mesh"].cpu(), tt["tri_mesh"]) and tb["mesh_bsdf"].tolist() == tt["mesh_bsdf"].tolist()
    assert tb["mesh_emitter"].tolist() == tt["mesh_emitter"].tolist() and torch.equal(tb["bsdf_rec"].cpu(), tt["bsdf_rec"])
    _close(tb["texels"], tt["texels"], 1e-7)
    _close(tb["cam"][:55], tt["cam"][:55], 1e-6)
    _close(tb["emitter_f"], tt["emitter_f"], 1e-6)
    assert tb["emitter_i"].tolist() == tt["emitter_i"].tolist()
    _close(tb["face_cmf"], tt["face_cmf"], 1e-6); _close(tb["face_pmf"], tt["face_pmf"], 1e-6)
    # edge tables: same SET of records (the row order follows each builder's edge enumeration)
    assert tb["num_sec_edges"] == tt["num_sec_edges"] and tb["num_prim_edges"] == tt["num_prim_edges"] > 0
    _close(_sorted_rows(tb["sec_edge"]), _sorted_rows(tt["sec_edge"]), 2e-6)
    # (the 2-D edge normal is the normalised difference of two projected points: short bunny edges lose 3 digits in fp32)
    _close(_sorted_rows(tb["prim_edge"][:, :7]), _sorted_rows(tt["prim_edge"][:, :7]), 3e-4)
    assert abs(tb["sec_sum"] / tt["sec_sum"] - 1) < 1e-5 and abs(tb["prim_sum"] / tt["prim_sum"] - 1) < 1e-5
    # the adjacent-face table agrees with the records it annotates: n0 / n1 are those faces' normals
    fa, se = tb["sec_edge_faces"].long().cpu(), tb["sec_edge"].cpu()
    _close(tb["tri_info"].cpu()[fa[:, 0], 18:21], se[:, 6:9], 1e-6)
    inner = fa[:, 1] >= 0
    _close(tb["tri_info"].cpu()[fa[inner, 1], 18:21], se[inner, 9:12], 1e-6)
    assert torch.equal(~inner, se[:, 15] != 0)


@pytest.mark.parametrize("scene,mesh", [("cbox", 0), ("cbox_occluder", 1)])
def test_product_tangent_tables_equal_the_independent_chain(scene, mesh):
    sc, P = load_scene(scene, res=16, spp=1, sppe=1, sppse=1, translate=(mesh, (1.0, 0.5, 0.0)))
    tan = tangents_wrt(sc.tables(0), P)
    sc0, _ = load_scene(scene, res=16, spp=1, sppe=1, sppse=1)
    _, _, prim, tang = to.render_d(to.scene_inputs(sc0), mesh, (1.0, 0.5, 0.0), spp=0, sppe=0, sppse=0)
    _close(tan["tri_info"][:, :22], tang["tri_info"][:, :22], 1e-5)
    assert float(tang["tri_info"].abs().max()) > 0.5
    # rows of (value | tangent) as a set: both builders enumerate the same edges, possibly in another order
    tb = sc.tables(0)
    _close(_sorted_rows(torch.cat([tb["sec_edge"].detach(), tan["sec_edge"]], 1)), _sorted_rows(torch.cat([prim["sec_edge"], tang["sec_edge"]], 1)), 1e-5)
    _close(_sorted_rows(torch.cat([tb["prim_edge"].detach()[:, :7], tan["prim_edge"][:, :7]], 1)),
           _sorted_rows(torch.cat([prim["prim_edge"][:, :7], tang["prim_edge"][:, :7]], 1)), 1e-4)


@pytest.mark.parametrize("scene,mesh", [("cbox", 0), ("cbox_occluder", 1)])
def test_cpp_oracle_equals_torch_oracle_on_the_same_tables(scene, mesh):
    """fp64 against fp64 on identical (fp32-rounded) tables and tangent tables; what is left is the fp32 rounding of
    the C++ oracle's output buffers."""
    res, spp = 12, 2
    sc, _ = load_scene(scene, res=res, spp=spp, sppe=2, sppse=2)
    _, _, prim, tang = to.render_d(to.scene_inputs(sc), mesh, (1.0, 0.5, 0.0), spp=0, sppe=0, sppse=0)
    pf = to.to_float_tables(prim)
    tan = {k: tang[k].float() for k in ("tri_info", "sec_edge", "prim_edge")}
    img = to.render(to.build_tables(to.scene_inputs(sc)), spp=spp).reshape(-1, 3).numpy()      # renderC on the fp64 tables
    assert rel_l2(img, oracle.render(pf, _abi.make_opts(spp=spp), precision=1, reference_form=True)) < 1e-6
    for kw in (dict(spp=spp), dict(spp=0, sppe=2), dict(spp=0, sppse=2), dict(spp=spp, sppe=2, sppse=2)):
        o = _abi.make_opts(spp=kw.get("spp", 0), sppe=kw.get("sppe", 0), sppse=kw.get("sppse", 0))
        a_img, a_d = to.render_d_from_tables(pf, tan, **kw)
        b_img, b_d = oracle.render(pf, o, mode=1, tangents=tan, precision=1, reference_form=True)
        if len(kw) == 3:
            assert np.abs(b_d).max() > 1e-3
        assert rel_l2(a_d.numpy(), b_d) < 1e-6 or np.abs(b_d).max() == 0, (kw, rel_l2(a_d.numpy(), b_d))
        if kw.get("spp", 0):
            assert rel_l2(a_img.numpy(), b_img) < 1e-6
        # the C++ oracle's robust forms (what the product evaluates) are the same function in fp64
        c_d = oracle.render(pf, o, mode=1, tangents=tan, precision=1, reference_form=False)[1]
        assert rel_l2(c_d, b_d) < 1e-6 or np.abs(b_d).max() == 0, kw


@pytest.mark.parametrize("name", list(mtg.CASES))
def test_cpp_oracle_reproduces_the_torch_fixtures(name):
    """The committed fixtures were produced by the torch oracle through ITS table chain; the C++ oracle gets the
    product's fp32 tables and JVP tangents.  fp32 tables against fp64 tables: 1e-5."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    scene, res, spp, sppe, sppse, mesh, direction = mtg.CASES[name]
    sc, P = load_scene(scene, res=res, spp=spp, sppe=sppe, sppse=sppse, translate=(mesh, direction))
    tb = sc.tables(0)
    _close(tb["tri_info"][:, :22], g["tri_info"][:, :22], 2e-6)
    img, dimg = oracle.render(tb, _abi.make_opts(spp=spp, sppe=sppe, sppse=sppse), mode=1, tangents=tangents_wrt(tb, P), precision=1, reference_form=True)
    assert rel_l2(img, g["img"]) < 1e-5 and rel_l2(dimg, g["dimg"]) < 1e-4, (rel_l2(img, g["img"]), rel_l2(dimg, g["dimg"]))


def test_torch_oracle_reproduces_its_fixtures():
    name = "torch_cbox_occluder_d"
    g = np.load(os.path.join(GOLD, name + ".npz"))
    _, _, img, dimg, _, _ = mtg.run_case(name)
    assert rel_l2(img, g["img"]) < 1e-12 and rel_l2(dimg, g["dimg"]) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mtg.CASES))
def test_gpu_reproduces_the_torch_fixtures(name):
    """HIP kernels (fp32, through the C ABI, product tables + JVP tangents) against the torch oracle's fixtures."""
    from helpers import GpuScene
    g = np.load(os.path.join(GOLD, name + ".npz"))
    scene, res, spp, sppe, sppse, mesh, direction = mtg.CASES[name]
    sc, P = load_scene(scene, res=res, spp=spp, sppe=sppe, sppse=sppse, translate=(mesh, direction))
    tb = sc.tables(0)
    img, dimg = GpuScene(tb).render_d_fwd(_abi.make_opts(spp=spp, sppe=sppe, sppse=sppse), [tangents_wrt(tb, P)])
    print("%s: GPU vs torch-oracle fixture: image rel-L2 %.2e, derivative image rel-L2 %.2e" % (name, rel_l2(img, g["img"]), rel_l2(dimg[0], g["dimg"])))
    assert rel_l2(img, g["img"]) < 1e-4 and rel_l2(dimg[0], g["dimg"]) < 1e-3


@pytest.mark.parametrize("scene,mesh", [("cbox_rough", 0), ("cbox", 1)])
def test_second_oracle_covers_rough_conductors_and_the_path_tracer(scene, mesh):
    """Round 6: GGX / RoughConductor (ggx.cpp:9-106, roughconductor.cpp:40-92, the conductor Fresnel term utils.h:148-164) and the PathTracer loop
     restated a SECOND time, in torch, from the reference's sources -- no line shared with the C++ oracle or the product.  Both oracles in fp64 on the
    product's fp32 tables: renderC, renderD w.r.t. EVERY texel (albedo, roughness, eta, k: a random tangent) and w.r.t. a mesh translation (interior term through the
    rough BSDF's sampling pdf and Fresnel term; all three terms for the DirectIntegrator), DirectIntegrator(1,1) / (2,2) and PathTracer(3)."""
    res, spp = 10, 2
    sc, P = load_scene(scene, res=res, spp=spp, sppe=2, sppse=2, translate=(mesh, (1.0, 0.5, 0.25)))
    tb = sc.tables(0)
    geo = {k: v for k, v in tangents_wrt(tb, P).items() if v is not None}
    tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
    g = torch.Generator().manual_seed(3)
    tex = {"texels": torch.rand(tbc["texels"].shape, generator=g) - 0.3}
    if scene == "cbox_rough":
        assert int((tbc["bsdf_rec"][:, 0] == _abi.BSDF_ROUGHCONDUCTOR).sum()) >= 1
    for name, kw_t, kw_c in (("direct11", dict(B=1, L=1), dict(bsdf_samples=1, light_samples=1)), ("direct22", dict(B=2, L=2), dict(bsdf_samples=2, light_samples=2)),
                             ("path3", dict(depth=3), dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3))):
        img = to.render({k: (v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v) for k, v in tbc.items()}, spp=spp, **kw_t).reshape(-1, 3).numpy()
        ref = oracle.render(tb, _abi.make_opts(spp=spp, **kw_c), precision=1, reference_form=True)
        assert ref.mean() > 0.05 and rel_l2(img, ref) < 1e-6, (name, rel_l2(img, ref))
        for what, tan, edges in (("texels", tex, False), ("geometry", geo, name != "path3")):
            se = 2 if edges else 0
            a_img, a_d = to.render_d_from_tables(tbc, {k: v.detach().cpu() for k, v in tan.items()}, spp=spp, sppe=se, sppse=se, **kw_t)
            b_img, b_d = oracle.render(tb, _abi.make_opts(spp=spp, sppe=se, sppse=se, **kw_c), mode=1, tangents=tan, precision=1, reference_form=True)
            assert np.abs(b_d).max() > 1e-4, (name, what)
            assert rel_l2(a_img.numpy(), b_img) < 1e-6 and rel_l2(a_d.numpy(), b_d) < 2e-6, (name, what, rel_l2(a_img.numpy(), b_img), rel_l2(a_d.numpy(), b_d))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mtg.TABLE_CASES))
def test_gpu_reproduces_the_torch_fixtures_of_rough_conductors_and_the_path_tracer(name):
    """tests/golden/torch_cbox_rough_*.npz: the torch oracle's renderer (GGX, conductor Fresnel, the PathTracer loop: its own restatement) on the product's tables, image
    and derivative image w.r.t. every texel.  The HIP kernels through the C ABI against them: an oracle-independent anchor for the rough-conductor path and the
    PathTracer, like the diffuse fixtures above."""
    from helpers import GpuScene
    g = np.load(os.path.join(GOLD, name + ".npz"))
    scene, res, spp, _, kw_c = mtg.TABLE_CASES[name]
    sc, _ = load_scene(scene, res=res, spp=spp)
    tb = sc.tables(0)
    _close(tb["texels"], g["texels"], 1e-7)
    tan = {"texels": mtg.texel_tangent(tb["texels"].numel())}
    img, dimg = GpuScene(tb).render_d_fwd(_abi.make_opts(spp=spp, **kw_c), [tan])
    print("%s: GPU vs torch-oracle fixture: image rel-L2 %.2e, derivative image rel-L2 %.2e" % (name, rel_l2(img, g["img"]), rel_l2(dimg[0], g["dimg"])))
    assert rel_l2(img, g["img"]) < 1e-4 and rel_l2(dimg[0], g["dimg"]) < 1e-3


@pytest.mark.parametrize("name", list(mtg.TABLE_CASES))
def test_cpp_oracle_reproduces_the_torch_fixtures_of_rough_conductors(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    scene, res, spp, _, kw_c = mtg.TABLE_CASES[name]
    sc, _ = load_scene(scene, res=res, spp=spp)
    tb = sc.tables(0)
    _close(tb["texels"], g["texels"], 1e-7)
    img, dimg = oracle.render(tb, _abi.make_opts(spp=spp, **kw_c), mode=1, tangents={"texels": mtg.texel_tangent(tb["texels"].numel())}, precision=1, reference_form=True)
    assert rel_l2(img, g["img"]) < 1e-6 and rel_l2(dimg, g["dimg"]) < 2e-6, (rel_l2(img, g["img"]), rel_l2(dimg, g["dimg"]))
