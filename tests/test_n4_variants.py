"""the reference's two compile-time variants (include/psdr/macros.h:11-13), built as run-time options.

  PSDR_MESH_ENABLE_1D_VERTEX_OFFSET   Mesh.vertex_offset: world positions = to_world * (raw + offset * raw vertex normal)
                                      (shape/mesh.h:37-39,71-80, mesh.cpp:226-232, psdr.cpp:259)
  PSDR_PRIMARY_EDGE_VIS_CHECK         RenderOption.primary_edge_vis_check -> psdr_scene_desc::prim_edge_z: a primary-edge
                                      sample counts only if the edge point is visible from the camera
                                      (integrator.cpp:105-108, perspective.cpp:91-96,171-196, edge.h:19-35)
"""
import numpy as np
import pytest
import torch

import enoki as ek
import oracle
import psdr_cuda
from enoki.cuda_autodiff import Float32 as FloatD
from helpers import host_render, load_scene, rel_l2, tangents_wrt
from psdr_cuda import _abi
from psdr_cuda.fixtures import scene_path


def _scene(name, res, spp, sppe, sppse, vis=False):
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    sc.opts.primary_edge_vis_check = vis
    return sc


def test_vertex_offset_moves_vertices_along_their_raw_normals():
    sc = _scene("cbox_occluder", 16, 1, 1, 1)
    sc.configure()
    base = sc.tables(0)["tri_info"].clone()
    m = sc.param_map["Mesh[1]"]
    v0, n0 = m.vertex_positions.t.clone(), m.vertex_normals.t.clone()
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    m.vertex_offset = FloatD(0.) + P                        # the same offset for every vertex
    sc.configure()
    tb = sc.tables(0)
    assert torch.allclose(tb["tri_info"].detach(), base)      # offset 0: unchanged
    tan = tangents_wrt(tb, P)
    # finite differences of the table chain
    eps = 1e-2
    tabs = []
    for s in (-eps, eps):
        m.vertex_offset = FloatD(float(s))
        sc.configure()
        tabs.append(sc.tables(0)["tri_info"].detach().clone())
        assert torch.allclose(m._raw_positions(), v0 + n0 * s, atol=1e-5)
    fd = (tabs[1] - tabs[0]) / (2 * eps)
    f0, f1 = sc.tables(0)["face_offset"][1], sc.tables(0)["face_offset"][2]
    assert float(tan["tri_info"][f0:f1, :9].abs().max()) > 0.1
    assert torch.allclose(tan["tri_info"][:, :9], fd[:, :9], atol=2e-3)          # p0, e1, e2 are linear in the offset
    assert float(tan["tri_info"][:f0].abs().max()) == 0.0                         # the other meshes do not move


def test_vis_check_host_product_matches_oracle():
    sc = _scene("cbox_bunny", 24, 0, 8, 0, vis=True)            # Mesh[1] = the bunny: part of its silhouette hides behind itself
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    from enoki.cuda_autodiff import Vector3f as Vector3fD, Matrix4f as Matrix4fD
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.5, 0.0]) * P))
    sc.configure()
    tb = sc.tables(0)
    assert tb["prim_edge_z"] is not None and tuple(tb["prim_edge_z"].shape) == (tb["num_prim_edges"], 4)
    o = _abi.make_opts(spp=0, sppe=8, sppse=0)
    tan = tangents_wrt(tb, P)
    _, d_host = host_render(tb, o, mode=1, tangents=tan)
    _, d_ref = oracle.render(tb, o, mode=1, tangents=tan)
    assert np.abs(d_ref).max() > 0 and rel_l2(d_host, d_ref) < 1e-3
    # without the check the estimator has the same expectation (both rays of a hidden edge point see the same surface) but the
    # hidden samples add noise: the two images differ, their sums agree
    tb2 = dict(tb); tb2["prim_edge_z"] = None
    _, d_plain = oracle.render(tb2, o, mode=1, tangents=tan)
    assert rel_l2(d_plain, d_ref) > 0.05 and abs(np.abs(d_plain).sum() - np.abs(d_ref).sum()) < 0.02 * np.abs(d_ref).sum()
    # the table: 1 / depth of the end points along the viewing direction, and the adjacent faces
    cam = tb["cam"].double()
    pos, cd = cam[48:51], cam[51:54] / cam[51:54].norm()
    w2s = cam[32:48].reshape(4, 4)
    z = tb["prim_edge_z"]
    faces = z[:, 2:4].contiguous().view(torch.int32)
    assert int(faces.min()) >= 0 and int(faces.max()) < tb["tri_info"].shape[0]
    info = tb["tri_info"].detach().double()
    pe = tb["prim_edge"].detach().double()
    for e in range(0, len(pe), max(1, len(pe) // 16)):
        # one end point of the edge is a vertex of its first face: its depth and film position match the row
        f = int(faces[e, 0])
        verts = torch.stack([info[f, 0:3], info[f, 0:3] + info[f, 3:6], info[f, 0:3] + info[f, 6:9]])
        depth = (verts - pos) @ cd
        q = torch.cat([verts, torch.ones(3, 1, dtype=torch.float64)], 1) @ w2s.T
        film = q[:, :2] / q[:, 3:4]
        k = int((film - pe[e, 0:2]).norm(dim=1).argmin())
        assert torch.allclose(film[k], pe[e, 0:2], atol=1e-5)
        assert abs(1.0 / float(depth[k]) - float(z[e, 0])) < 1e-6 * abs(float(z[e, 0]))


@pytest.mark.gpu
def test_vis_check_and_vertex_offset_on_the_gpu():
    from helpers import GpuScene
    from enoki.cuda_autodiff import Vector3f as Vector3fD, Matrix4f as Matrix4fD
    # vis check: GPU forward and reverse against the oracle / each other
    sc = _scene("cbox_bunny", 48, 4, 16, 0, vis=True)
    P = FloatD(0.)
    ek.set_requires_gradient(P)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD.translate(Vector3fD([1.0, 0.0, 0.5]) * P))
    sc.configure()
    tb = sc.tables(0)
    g = GpuScene(tb)
    o = _abi.make_opts(spp=4, sppe=16, sppse=0)
    tan = tangents_wrt(tb, P)
    img, dimg = g.render_d_fwd(o, [tan])
    rimg, rd = oracle.render(tb, o, mode=1, tangents=tan)
    # (an edge sample whose film point sits within an ulp of a pixel border lands left or right of it with the device's or the host's division: two
    # pixels of 2 304 trade ONE of 36 864 samples -- rel-L2 4e-3 on this tiny frame; so: at most a handful of pixels differ, the rest to 1e-3)
    off = float((np.abs(dimg[0] - rd).max(1) > 1e-3 * (1 + np.abs(rd).max(1))).mean())
    print("vis check: derivative image rel-L2 vs oracle %.2e, pixels off by > 1e-3: %.2e" % (rel_l2(dimg[0], rd), off))
    assert rel_l2(img, rimg) < 1e-4 and off < 2e-3 and rel_l2(dimg[0], rd) < 1e-2
    adj = np.random.default_rng(2).random((48 * 48, 3)).astype(np.float32)
    _, grads = g.render_d_rev(o, adj, with_image=False)
    lhs = float((adj.astype(np.float64) * dimg[0]).sum())
    rhs = float(sum((grads[k].astype(np.float64) * tan[k].detach().cpu().numpy()).sum() for k in tan if tan[k] is not None and k in grads))
    assert abs(lhs - rhs) < 2e-3 * float(np.abs(adj * dimg[0]).sum()), (lhs, rhs)
    rays_with = g.counters()[0]
    tb2 = dict(tb); tb2["prim_edge_z"] = None
    g2 = GpuScene(tb2)
    g2.render_d_rev(o, adj, with_image=False)
    assert rays_with != g2.counters()[0]                      # one more camera ray per edge sample, fewer Li evaluations
    # vertex offset: renderD + forward through the surface against the oracle
    sc = _scene("cbox_occluder", 32, 8, 8, 8)
    Q = FloatD(0.)
    ek.set_requires_gradient(Q)
    sc.param_map["Mesh[1]"].vertex_offset = FloatD(0.) + Q
    sc.configure()
    integ = psdr_cuda.DirectIntegrator(1, 1)
    im = integ.renderD(sc, 0)
    ek.forward(Q)
    d = ek.gradient(im).numpy()
    tb = sc.tables(0)
    _, rd = oracle.render(tb, _abi.make_opts(spp=8, sppe=8, sppse=8), mode=1, tangents=tangents_wrt(tb, Q))
    assert np.abs(rd).max() > 0 and rel_l2(d, rd) < 1e-3
