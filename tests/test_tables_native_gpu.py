"""The table chain of Scene::configure on the HIP library (csrc/psdr_tables.hip, psdr_cuda/tables_native.py) against its torch formulation
(scene.py process_mesh / _secondary_edges / PerspectiveCamera.configure -- itself pinned on the independent torch oracle by
tests/test_second_oracle.py): same tables, same reverse-mode gradients, forward mode (double backward) still available."""
import numpy as np
import pytest
import torch

import enoki as ek
import psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
from helpers import load_scene, rel_l2, tangents_wrt
from psdr_cuda import tables_native
from psdr_cuda.fixtures import scene_path

pytestmark = pytest.mark.gpu
KEYS = ("tri_info", "sec_edge", "prim_edge", "prim_edge_z", "sec_edge_faces", "sec_pmf", "prim_pmf")


def build(scene, native, grad):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(scene), False)
    sc.opts.width = sc.opts.height = 64
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 2, 2, 2, 0
    sc.opts.primary_edge_vis_check = True
    mesh = sc.param_map["Mesh[1]"]
    v = Vector3fD(ek.detach(mesh.vertex_positions))
    if grad:
        ek.set_requires_gradient(v)
    mesh.vertex_positions = v
    cam = sc.m_sensors[0]
    tw = cam._to_world.detach().clone().requires_grad_(grad)
    cam._to_world = tw
    if native:
        sc.configure()
    else:
        with tables_native.torch_formulation():
            sc.configure()
    return sc, v, tw


@pytest.mark.parametrize("scene", ["cbox_bunny", "cbox_occluder", "bunny_light"])
def test_native_tables_equal_the_torch_formulation(scene):
    a, _, _ = build(scene, True, False)
    b, _, _ = build(scene, False, False)
    ta, tb = a.tables(0), b.tables(0)
    assert ta["num_sec_edges"] > 0 and ta["num_prim_edges"] > 0

    def common(ka, kb):
        # an edge whose two faces are coplanar to within an ulp of the 1 - 1e-5 threshold may be kept by one formulation and dropped by the
        # other (14 724 against 14 725 edges on bunny_light): compare the edges both kept, allow two strays
        ia, ib = ta[ka].detach().cpu().numpy(), tb[kb].detach().cpu().numpy()
        key = lambda r: r[:, :6].round(3).astype(np.float32).tobytes() if False else [tuple(np.round(q[:6], 3)) for q in r]
        da = {q: i for i, q in enumerate(key(ia))}
        db = {q: i for i, q in enumerate(key(ib))}
        both = sorted(set(da) & set(db))
        assert len(set(da) ^ set(db)) <= 2, (len(da), len(db))
        return np.array([da[q] for q in both]), np.array([db[q] for q in both])
    sa, sb = common("sec_edge", "sec_edge")
    pa, pb = common("prim_edge", "prim_edge") if False else (np.arange(ta["num_prim_edges"]), np.arange(tb["num_prim_edges"]))
    assert ta["num_prim_edges"] == tb["num_prim_edges"]
    for k in KEYS:
        x, y = ta[k].detach().cpu().numpy(), tb[k].detach().cpu().numpy()
        if k.startswith("sec_"):
            x, y = x[sa], y[sb]
            if k == "sec_pmf":
                continue                                                       # cmf / pmf follow from the rows
        assert x.shape == y.shape, k
        if k.endswith("_pmf"):                    # the native chain normalises on the device (the descriptor's sum is 1), the torch chain keeps lengths + their sum
            x, y = x.astype(np.float64) / x.astype(np.float64).sum(), y.astype(np.float64) / y.astype(np.float64).sum()
            assert abs(float(ta[k].double().sum()) - 1.0) < 1e-5 and ta[k[:-3] + "sum"] == 1.0
        if x.dtype.kind in "iu":
            assert np.array_equal(x, y), k
        elif k == "prim_edge_z":
            assert np.array_equal(x[:, 2:].view(np.int32), y[:, 2:].view(np.int32)) and rel_l2(x[:, :2], y[:, :2]) < 1e-6
        else:
            # film records: a projection of points ~1000 units away in fp32, then normals of (short) film-space differences
            assert rel_l2(x, y) < (5e-5 if k == "prim_edge" else 2e-6), (k, rel_l2(x, y))


@pytest.mark.parametrize("scene", ["cbox_bunny", "bunny_light"])
def test_native_reverse_gradients_equal_torch(scene):
    """random cotangents on the three differentiable tables -> gradients of the vertices and of the camera pose"""
    outs = []
    for native in (True, False):
        sc, v, tw = build(scene, native, True)
        t = sc.tables(0)
        loss = 0.0
        for i, k in enumerate(("tri_info", "sec_edge", "prim_edge")):
            w = torch.sin(37.0 * t[k].detach() + i) + 0.5          # weights that follow the row content: a stray borderline edge moves ONE row's share
            loss = loss + (w * t[k]).sum()
        loss.backward()
        outs.append((v.t.grad.detach().cpu().numpy().copy(), tw.grad.detach().cpu().numpy().copy()))
    (gv_n, gc_n), (gv_t, gc_t) = outs
    assert np.abs(gv_t).max() > 0 and np.abs(gc_t).max() > 0
    print("%s: vertex gradient native vs torch %.2e, camera pose %.2e" % (scene, rel_l2(gv_n, gv_t), rel_l2(gc_n, gc_t)))
    assert rel_l2(gv_n, gv_t) < (1e-4 if scene == "cbox_bunny" else 1e-2), rel_l2(gv_n, gv_t)       # bunny_light: one stray borderline edge (see above) = 3e-3
    assert rel_l2(gc_n, gc_t) < 1e-3, rel_l2(gc_n, gc_t)             # sums of ~1e4 terms of mixed sign in fp32


def test_forward_mode_still_works_through_the_native_chain():
    """enoki.forward differentiates the chain by double backward: the native ops hand that case to their torch formulation"""
    res = {}
    for native in (True, False):
        import contextlib
        with (contextlib.nullcontext() if native else tables_native.torch_formulation()):
            sc, P = load_scene("cbox_bunny", res=32, spp=2, sppe=2, sppse=2, translate=(1, (1.0, 0.5, 0.0)))
            tb = sc.tables(0)
            res[native] = {k: (None if t is None else t.detach().cpu().numpy()) for k, t in tangents_wrt(tb, P).items()}
    for k in ("tri_info", "sec_edge", "prim_edge"):
        assert np.abs(res[False][k]).max() > 0 and rel_l2(res[True][k], res[False][k]) < 1e-5, k


def test_world_vertices_kernel_equals_the_torch_chain_bit_for_bit_and_in_its_adjoint():
    """psdr_geo_world_vertices_*: the positions an affine to_world gives are THE SAME floats as the eager chain's (three products, left-to-right
    sums, the translation, w = 1 divides exactly) -- an edge whose faces are coplanar to an ulp sees the same vertices; the adjoint to 1e-6"""
    g = torch.Generator().manual_seed(5)
    V, M = 5000, 4
    v = (torch.rand(V, 3, generator=g) * 400 - 200).cuda().requires_grad_(True)
    vmesh = torch.randint(0, M, (V,), generator=g).cuda()
    mats = torch.eye(4).repeat(M, 1, 1)
    for k in range(M):                                            # rotation * scale + translation per mesh
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        mats[k, :3, :3] = q * (0.5 + k)
        mats[k, :3, 3] = torch.randn(3, generator=g) * 100
    mats = mats.cuda()

    def to_world(vv, mm):
        mv = mm[vmesh]
        h = (mv[:, :3, :3] * vv.unsqueeze(1)).sum(-1) + mv[:, :3, 3]
        w = (mv[:, 3, :3] * vv).sum(-1) + mv[:, 3, 3]
        return h / w.unsqueeze(-1)
    ref = to_world(v, mats)
    out = tables_native.world_vertices(v, vmesh.to(torch.int32), mats, to_world)
    assert torch.equal(out, ref)
    a = torch.randn(V, 3, generator=g).cuda()
    g_ref, = torch.autograd.grad(ref, v, a)
    g_out, = torch.autograd.grad(out, v, a)
    assert rel_l2(g_out.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6
    # a projective matrix: the division is this unit's approximate one
    mats2 = mats.clone(); mats2[:, 3, :3] = 1e-3; mats2[:, 3, 3] = 1.5
    o2, r2 = tables_native.world_vertices(v, vmesh.to(torch.int32), mats2, to_world), to_world(v, mats2)
    assert rel_l2(o2.detach().cpu().numpy(), r2.detach().cpu().numpy()) < 1e-6
    ga, = torch.autograd.grad(o2, v, a); gb, = torch.autograd.grad(r2, v, a)
    assert rel_l2(ga.cpu().numpy(), gb.cpu().numpy()) < 1e-5
    # forward mode through the op (double backward falls back to the torch formulation)
    u = torch.zeros(V, 3, device="cuda", requires_grad=True)
    o3 = tables_native.world_vertices(v, vmesh.to(torch.int32), mats, to_world)
    gv, = torch.autograd.grad(o3, v, u, create_graph=True)
    t = torch.randn(V, 3, generator=g).cuda()
    jvp, = torch.autograd.grad(gv, u, t)
    jref = torch.autograd.functional.jvp(lambda x: to_world(x, mats), (v.detach(),), (t,))[1]
    assert rel_l2(jvp.cpu().numpy(), jref.cpu().numpy()) < 1e-6


def test_configure_and_render_are_reproducible_from_one_scene_object_to_the_next():
    """Two FRESH scenes of the same file in one process: the table chain must produce the SAME bits (the vertex normals are sums over the faces around a
    vertex -- accumulated in double so that the order of the atomics cannot show, csrc/psdr_tables.hip k_face_accum; with fp32 atomics 12 000 words of the
    cbox_bunny rows moved in their last bit from one configure() to the next and a PathTracer pixel with them by 3e-4), and so must the image."""
    imgs, rows, edges = [], [], []
    for _ in range(3):
        sc, _, _ = build("cbox_bunny", True, True)
        t = sc.tables(0)
        rows.append(t["tri_info"].detach().cpu().numpy().copy()); edges.append(t["sec_edge"].detach().cpu().numpy().copy())
        sc.opts.spp = 8
        imgs.append(psdr_cuda.PathTracer(3).renderC(sc, 0).numpy().copy())
    for k in (1, 2):
        assert rows[k].shape == rows[0].shape and int((rows[k] != rows[0]).sum()) == 0, int((rows[k] != rows[0]).sum())
        assert edges[k].shape == edges[0].shape and int((edges[k] != edges[0]).sum()) == 0
        # same tables, same sample streams: what is left is the order of the image atomics
        assert float(np.abs(imgs[k] - imgs[0]).max()) <= 2e-5 * max(1.0, float(np.abs(imgs[0]).max())), float(np.abs(imgs[k] - imgs[0]).max())


def test_configure_never_waits_for_the_device_and_stays_under_twenty_launches():
    """Scene.configure with vertex gradients: ZERO synchronising calls (torch's sync debug mode counts them; the numbers of kept edges, the
    distribution sums, mesh areas and emitter weights stay on the device: csrc/psdr_tables.hip k_compact_*, k_mesh_areas, k_emitter_rows) and at most
    20 device launches / copies (the reference's configure is one Enoki trace, scene.cpp:56-278).  Then a full optimisation
    iteration -- configure, renderD, loss, backward -- without a single wait either."""
    import warnings
    from torch.profiler import profile, ProfilerActivity
    sc, v, _ = build("cbox_bunny", True, False)
    mesh = sc.param_map["Mesh[1]"]
    integ = psdr_cuda.DirectIntegrator(1, 1)

    def fresh():
        vv = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(vv); mesh.vertex_positions = vv
        return vv
    for _ in range(2):                                            # steady state: caches of the first configure() in place
        fresh(); sc.configure()
    fresh()
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                sc.configure()
                torch.cuda.set_sync_debug_mode("default")
                torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("warn")
            vv = fresh()
            sc.configure()
            img = integ.renderD(sc, 0)
            ek.backward(FloatD._wrap(((img.t - 0.3) ** 2).sum().reshape(1)))
            g = ek.gradient(vv)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    syncs = [x for x in w if "called a synchronizing" in str(x.message)]
    assert len(syncs) == 0, [str(x.message)[:80] for x in syncs]
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print("configure(): %d device launches / copies" % len(ev))
    assert 0 < len(ev) <= 20, [e.name[:60] for e in ev]
    assert float(g.t.abs().max()) > 0
    # the public tables are cut to the kept rows (one read of the two counts), the render calls take the capacity tables
    t, tc = sc.tables(0), sc.tables(0, capacity=True)
    n_sec, n_prim = t["num_sec_edges"], t["num_prim_edges"]
    assert 0 < n_sec < tc["num_sec_edges"] == tc["sec_edge"].shape[0] and 0 < n_prim < tc["num_prim_edges"]
    assert t["sec_edge"].shape[0] == n_sec and torch.equal(t["sec_edge"], tc["sec_edge"][:n_sec])
    assert float(tc["sec_edge"].detach()[n_sec:].abs().max()) == 0 and float(tc["sec_pmf"][n_sec:].abs().max()) == 0 and float(tc["sec_cmf"][n_sec - 1:].min()) == 1.0
    assert float(tc["prim_pmf"][n_prim:].abs().max()) == 0 and float(tc["prim_cmf"][n_prim - 1:].min()) == 1.0 and float(tc["prim_pmf"][:n_prim].min()) > 0
    assert tc["sec_sum"] == tc["prim_sum"] == tc["emitter_sum"] == 1.0
    assert abs(float(tc["sec_pmf"].double().sum()) - 1) < 1e-5 and bool((tc["sec_cmf"][1:] >= tc["sec_cmf"][:-1]).all())


def test_compaction_kernel_against_a_boolean_mask_select():
    """psdr_geo_compact_edges_*: kept rows in order, aux words alongside, normalised pmf / cmf, count + sum header; adjoint = the scatter back"""
    g = torch.Generator(device="cuda").manual_seed(3)
    for E in (1, 5, 1024, 1025, 70001):
        rows = torch.randn(E, 16, device="cuda", generator=g)
        keep = (torch.rand(E, device="cuda", generator=g) < (0.0 if E == 5 else 0.37)).to(torch.uint8)
        aux = torch.randint(-5, 1 << 20, (E, 5), device="cuda", generator=g, dtype=torch.int32)
        rows.requires_grad_(True)
        out, aux_out, pos, pmf, cmf, hdr = tables_native.compact_edges(rows, keep, 3, 3, aux=aux[:, 2:4], aux_cols=2)
        k = keep.bool()
        n = int(k.sum())
        assert int(hdr[:1].view(torch.int32)) == n
        assert torch.equal(out[:n], rows[k]) and float(out[n:].abs().sum()) == 0
        assert torch.equal(aux_out[:n], aux[k][:, 2:4]) and int(aux_out[n:].abs().sum()) == 0
        ln = rows[k][:, 3:6].detach().double().norm(dim=1)
        assert abs(float(hdr[1]) - float(ln.sum())) <= 1e-5 * max(float(ln.sum()), 1e-30)
        if n:
            assert rel_l2(pmf[:n].cpu().numpy(), (ln / ln.sum()).cpu().numpy()) < 1e-6
            ref_cmf = torch.cumsum(ln / ln.sum(), 0)
            assert float((cmf[:n - 1].double() - ref_cmf[:n - 1]).abs().max()) < 1e-5 if n > 1 else True
        assert float(cmf[max(n - 1, 0):].min()) == 1.0 and float(pmf[n:].abs().sum()) == 0
        idx = torch.nonzero(k).reshape(-1)
        assert torch.equal(pos[k].long(), torch.arange(n, device="cuda")) and bool((pos[~k] == -1).all())
        w = torch.randn(E, 16, device="cuda", generator=g)
        (out * w).sum().backward()
        ref = torch.zeros(E, 16, device="cuda")
        ref[idx] = w[:n]
        assert torch.equal(rows.grad, ref)


def test_emitter_tables_kernel_against_the_host_formulation():
    """psdr_geo_emitter_tables (mesh areas, normalised emitter weights, face distributions) against the eager chain's numbers (scene.cpp:183-196)"""
    a, _, _ = build("bunny_light", True, False)
    with tables_native.torch_formulation():
        b, _, _ = build("bunny_light", False, False)
    ta, tb = a.tables(0), b.tables(0)
    for ma, mb in zip(a.m_meshes, b.m_meshes):
        assert abs(ma.m_total_area / mb.m_total_area - 1) < 1e-5 and abs(ma.m_inv_total_area * ma.m_total_area - 1) < 1e-12
    assert ta["num_emitters"] == tb["num_emitters"] >= 1
    ea, eb = ta["emitter_f"].cpu().numpy(), tb["emitter_f"].cpu().numpy()
    assert rel_l2(ea[:, :6], eb[:, :6]) < 1e-5
    assert torch.equal(ta["emitter_i"].cpu(), tb["emitter_i"].cpu()) and torch.equal(ta["mesh_emitter"].cpu(), tb["mesh_emitter"].cpu())
    assert rel_l2(ta["face_pmf"].cpu().numpy(), tb["face_pmf"].cpu().numpy()) < 1e-6 and rel_l2(ta["face_cmf"].cpu().numpy(), tb["face_cmf"].cpu().numpy()) < 1e-5
    pa = ta["emitter_pmf"].double().cpu().numpy()
    pb = tb["emitter_pmf"].double().cpu().numpy() / tb["emitter_sum"]
    assert abs(pa.sum() - 1) < 1e-6 and rel_l2(pa, pb) < 1e-6 and ta["emitter_sum"] == 1.0
    for x, y in zip(a.m_emitters, b.m_emitters):
        assert abs(x.m_sampling_weight - y.m_sampling_weight) < 1e-6
        fa, fb = x.m_mesh._face_distrb, y.m_mesh._face_distrb
        assert fa.m_size == fb.m_size and abs(fa.m_sum / fb.m_sum - 1) < 1e-5


def test_configure_reads_back_in_two_batches_on_the_eager_chain():
    """The eager torch chain (tables_native.torch_formulation: the formulation the native chain is checked against) synchronises with the device twice
    (sizes and sums, then the edge-distribution sums) -- it used to read seventeen values back one by one; torch's sync debug mode counts the calls"""
    import warnings
    with tables_native.torch_formulation():
        _check_two_batches()


def _check_two_batches():
    import warnings
    sc, v, _ = build("cbox_bunny", False, True)
    mesh = sc.param_map["Mesh[1]"]
    for _ in range(2):                                            # steady state: caches of the first configure() in place
        vv = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(vv); mesh.vertex_positions = vv
        sc.configure()
    vv = Vector3fD(ek.detach(mesh.vertex_positions)); ek.set_requires_gradient(vv); mesh.vertex_positions = vv
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            sc.configure()
        finally:
            torch.cuda.set_sync_debug_mode("default")
    syncs = [x for x in w if "called a synchronizing" in str(x.message)]
    assert len(syncs) <= 2, [str(x.message)[:80] for x in syncs]
    assert sc.tables(0)["num_sec_edges"] > 0 and sc.tables(0)["num_prim_edges"] > 0
