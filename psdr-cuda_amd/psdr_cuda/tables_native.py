"""The differentiable table chain of Scene::configure on the HIP library (csrc/psdr_tables.hip, include/psdr_hip.h psdr_geo_*): TriangleInfo
rows (process_mesh, reference src/shape/mesh.cpp:20-51), secondary-edge records (mesh.cpp:251-270 + scene.cpp:219-244) and primary-edge
records (perspective.cpp:39-111), each ONE forward and ONE reverse call wrapped in a torch.autograd.Function -- instead of ~150 eager torch
launches per configure() and ~250 in its backward.

Forward mode (enoki.forward) differentiates the chain by double backward (create_graph=True): a backward call whose incoming adjoint itself
requires a gradient re-runs the torch formulation of the op (scene.py), which torch can differentiate again; plain reverse mode
(enoki.backward, the optimisation loop) takes the kernels.  CPU tensors (no GPU: the host tests) always take the torch formulation."""
import ctypes as C

import torch

from . import _abi


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


import os

_enabled = True          # set_enabled(False) (tests, tools): the eager torch chain everywhere.  The package reads no environment variable.


def set_enabled(on):
    """Test / tool hook: False = Scene.configure() builds the triangle / edge tables with the eager torch chain instead of the psdr_geo_* kernels."""
    global _enabled
    _enabled = bool(on)


def available(t):
    return _enabled and t.is_cuda


class torch_formulation:
    """context manager: configure() inside it builds the tables with the eager torch chain (the formulation the committed fixtures were made with:
    an edge whose faces are coplanar to within an ulp of the 1 - 1e-5 filter threshold can be kept by one formulation and dropped by the other,
    and the sample streams of a fixture follow its edge list)"""

    def __enter__(self):
        global _enabled
        self.old, _enabled = _enabled, False

    def __exit__(self, *a):
        global _enabled
        _enabled = self.old


class _WorldVertices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_raw, vmesh_i32, mats, ref_fn):
        lib = _abi.load_hip()
        vc, mc = v_raw.detach().contiguous().float(), mats.detach().contiguous().float()
        out = torch.empty_like(vc)
        _abi.check(lib, lib.psdr_geo_world_vertices_fwd(vc.shape[0], vc.data_ptr(), vmesh_i32.data_ptr(), mc.data_ptr(), out.data_ptr(), _stream()))
        ctx.save_for_backward(v_raw, vmesh_i32, mc, out)
        ctx.ref_fn = ref_fn
        return out

    @staticmethod
    def backward(ctx, a_world):
        v_raw, vmesh, mats, out = ctx.saved_tensors
        if a_world.requires_grad:                        # double backward (forward-mode JVP): the torch formulation
            with torch.enable_grad():
                vv = v_raw if v_raw.requires_grad else v_raw.detach().requires_grad_(True)
                g, = torch.autograd.grad(ctx.ref_fn(vv, mats), vv, a_world, create_graph=True)
            return g, None, None, None
        lib = _abi.load_hip()
        a = a_world.contiguous().float()
        a_raw = torch.empty_like(a)
        _abi.check(lib, lib.psdr_geo_world_vertices_rev(a.shape[0], v_raw.detach().contiguous().float().data_ptr(), vmesh.data_ptr(), mats.data_ptr(), out.data_ptr(),
                                                        a.data_ptr(), a_raw.data_ptr(), _stream()))
        return a_raw, None, None, None


def world_vertices(v_raw, vmesh_i32, mats, ref_fn):
    """transform_pos of every vertex by its mesh's matrix (mats [M,4,4] WITHOUT a gradient: the caller keeps the torch chain when a transform is
    being optimised); ref_fn(v_raw, mats) = the torch formulation."""
    return _WorldVertices.apply(v_raw, vmesh_i32, mats, ref_fn)


class _TriRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, faces_i32, ref_fn, width):
        lib = _abi.load_hip()
        vc = v.detach().contiguous().float()
        V, T = vc.shape[0], faces_i32.shape[0]
        vsum = torch.empty(V, 3, dtype=torch.float32, device=v.device)
        rows = torch.empty(T, width, dtype=torch.float32, device=v.device)
        _abi.check(lib, lib.psdr_geo_tri_rows_fwd(V, T, vc.data_ptr(), faces_i32.data_ptr(), vsum.data_ptr(), rows.data_ptr(), width, _stream()))
        ctx.save_for_backward(v, faces_i32, vsum)
        ctx.ref_fn, ctx.width = ref_fn, width
        return rows

    @staticmethod
    def backward(ctx, a_rows):
        v, faces, vsum = ctx.saved_tensors
        if a_rows.requires_grad:                         # double backward (forward-mode JVP): the torch formulation
            with torch.enable_grad():
                vv = v.detach().requires_grad_(True) if not v.requires_grad else v
                rows = ctx.ref_fn(vv, faces)
                g, = torch.autograd.grad(rows, vv, a_rows[:, :22], create_graph=True)
            return g, None, None, None
        lib = _abi.load_hip()
        V, T = v.shape[0], faces.shape[0]
        a = a_rows.contiguous().float()
        a_v = torch.zeros(V, 3, dtype=torch.float32, device=v.device)
        a_vsum = torch.empty(V, 3, dtype=torch.float32, device=v.device)
        _abi.check(lib, lib.psdr_geo_tri_rows_rev(V, T, v.detach().contiguous().data_ptr(), faces.data_ptr(), vsum.data_ptr(), a.data_ptr(), ctx.width,
                                                  a_vsum.data_ptr(), a_v.data_ptr(), _stream()))
        return a_v, None, None, None


def tri_rows(v_world, faces_i32, ref_fn, width=22):
    """rows [T, width] of process_mesh (width 24 = PSDR_TRI_STRIDE: the rows as psdr_scene_desc::tri_info holds them, padding words zero);
    ref_fn(v, faces) = the torch formulation [T, 22] (double backward, CPU)."""
    return _TriRows.apply(v_world, faces_i32, ref_fn, width)


class _SecEdges(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, rows, edges_i32, ref_fn):
        lib = _abi.load_hip()
        E = edges_i32.shape[0]
        vc, rc = v.detach().contiguous().float(), rows.detach().contiguous().float()
        info = torch.empty(E, 16, dtype=torch.float32, device=v.device)
        keep = torch.empty(E, dtype=torch.uint8, device=v.device)
        _abi.check(lib, lib.psdr_geo_sec_edges_fwd(E, edges_i32.data_ptr(), vc.data_ptr(), rc.data_ptr(), rc.shape[1], info.data_ptr(), keep.data_ptr(), _stream()))
        ctx.save_for_backward(v, rows, edges_i32)
        ctx.ref_fn = ref_fn
        ctx.mark_non_differentiable(keep)
        return info, keep

    @staticmethod
    def backward(ctx, a_info, _a_keep):
        v, rows, edges = ctx.saved_tensors
        if a_info.requires_grad:
            with torch.enable_grad():
                vv = v if v.requires_grad else v.detach().requires_grad_(True)
                rr = rows if rows.requires_grad else rows.detach().requires_grad_(True)
                info = ctx.ref_fn(vv, rr, edges)
                gv, gr = torch.autograd.grad(info, (vv, rr), a_info, create_graph=True, allow_unused=True)
            return gv, gr, None, None
        lib = _abi.load_hip()
        a = a_info.contiguous().float()
        a_v = torch.zeros_like(v, dtype=torch.float32)
        a_rows = torch.zeros(rows.shape, dtype=torch.float32, device=v.device)
        _abi.check(lib, lib.psdr_geo_sec_edges_rev(edges.shape[0], edges.data_ptr(), a.data_ptr(), a_v.data_ptr(), a_rows.data_ptr(), a_rows.shape[1], _stream()))
        return a_v, a_rows, None, None


def sec_edges(v_world, rows, edges_i32, ref_fn):
    """(info [E, 16], keep [E] uint8) for every candidate edge; ref_fn(v, rows, edges) -> info (torch formulation)."""
    return _SecEdges.apply(v_world, rows, edges_i32, ref_fn)


class _PrimEdges(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, w2s, rows, edges_i32, face_normals_u8, cam_pos, cam_dir, ref_fn, cam22):
        lib = _abi.load_hip()
        E = edges_i32.shape[0]
        vc, rc = v.detach().contiguous().float(), rows.detach().contiguous().float()
        if cam22 is None:
            cam22 = torch.cat([w2s.detach().reshape(-1), cam_pos.detach().reshape(-1), cam_dir.detach().reshape(-1)]).float().contiguous()
        rows8 = torch.empty(E, 8, dtype=torch.float32, device=v.device)
        z4 = torch.empty(E, 4, dtype=torch.float32, device=v.device)
        keep = torch.empty(E, dtype=torch.uint8, device=v.device)
        _abi.check(lib, lib.psdr_geo_prim_edges_fwd(E, edges_i32.data_ptr(), face_normals_u8.data_ptr(), vc.data_ptr(), rc.data_ptr(), rc.shape[1],
                                                    cam22.data_ptr(), rows8.data_ptr(), z4.data_ptr(), keep.data_ptr(), _stream()))
        ctx.save_for_backward(v, w2s, edges_i32, cam22)
        ctx.ref_fn = ref_fn
        ctx.mark_non_differentiable(z4, keep)
        return rows8, z4, keep

    @staticmethod
    def backward(ctx, a_rows8, _a_z, _a_keep):
        v, w2s, edges, cam22 = ctx.saved_tensors
        if a_rows8.requires_grad:
            with torch.enable_grad():
                vv = v if v.requires_grad else v.detach().requires_grad_(True)
                ww = w2s if w2s.requires_grad else w2s.detach().requires_grad_(True)
                r8 = ctx.ref_fn(vv, ww, edges)
                gv, gw = torch.autograd.grad(r8, (vv, ww), a_rows8, create_graph=True, allow_unused=True)
            return gv, gw, None, None, None, None, None, None, None
        lib = _abi.load_hip()
        a = a_rows8.contiguous().float()
        a_v = torch.zeros_like(v, dtype=torch.float32)
        a_w = torch.zeros(16, dtype=torch.float32, device=v.device)
        _abi.check(lib, lib.psdr_geo_prim_edges_rev(edges.shape[0], edges.data_ptr(), v.detach().contiguous().data_ptr(), cam22.data_ptr(), a.data_ptr(),
                                                    a_v.data_ptr(), a_w.data_ptr(), _stream()))
        return a_v, a_w.reshape(4, 4), None, None, None, None, None, None, None


def prim_edges(v_world, w2s, rows, edges_i32, face_normals_u8, cam_pos, cam_dir, ref_fn, cam22=None):
    """(rows8 [E, 8], z4 [E, 4], keep [E]) for every candidate edge of one sensor; ref_fn(v, w2s, edges) -> rows8 (torch formulation).
    cam22: world_to_sample | position | direction as one contiguous detached tensor when the caller holds it (words 32..53 of the camera record)."""
    return _PrimEdges.apply(v_world, w2s, rows, edges_i32, face_normals_u8, cam_pos, cam_dir, ref_fn, cam22)


class _CompactEdges(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, keep_u8, w0, wn, aux, aux_cols):
        lib = _abi.load_hip()
        rc = rows.detach().contiguous().float()
        E, S = rc.shape
        dev = rows.device
        A = int(aux_cols) if aux is not None else 0
        rows_out = torch.empty(E, S, dtype=torch.float32, device=dev)
        aux_out = torch.empty(E, max(A, 1), dtype=torch.int32, device=dev)
        pos = torch.empty(E, dtype=torch.int32, device=dev)
        dist = torch.empty(2 * E + 2, dtype=torch.float32, device=dev)              # pmf | cmf | header
        scratch = torch.empty(4 * ((E + 1023) // 1024), dtype=torch.int32, device=dev)
        _abi.check(lib, lib.psdr_geo_compact_edges_fwd(E, rc.data_ptr(), S, keep_u8.data_ptr(), int(w0), int(wn), aux.data_ptr() if A else None,
                                                       aux.stride(0) if A else 0, A, scratch.data_ptr(), rows_out.data_ptr(), aux_out.data_ptr(), pos.data_ptr(),
                                                       dist.data_ptr(), dist.data_ptr() + 4 * E, dist.data_ptr() + 8 * E, _stream()))
        ctx.save_for_backward(pos)
        ctx.mark_non_differentiable(aux_out, pos, dist)
        return rows_out, aux_out, pos, dist

    @staticmethod
    def backward(ctx, a_out, _a_aux, _a_pos, _a_dist):
        pos, = ctx.saved_tensors
        if a_out.requires_grad:                          # double backward (forward-mode JVP): a differentiable gather
            p = pos.long()
            return torch.where((p >= 0).unsqueeze(-1), a_out.index_select(0, p.clamp(min=0)), torch.zeros_like(a_out)), None, None, None, None, None
        lib = _abi.load_hip()
        a = a_out.contiguous().float()
        a_rows = torch.empty_like(a)
        _abi.check(lib, lib.psdr_geo_compact_edges_rev(a.shape[0], a.shape[1], pos.data_ptr(), a.data_ptr(), a_rows.data_ptr(), _stream()))
        return a_rows, None, None, None, None, None


def compact_edges(rows, keep_u8, w0, wn, aux=None, aux_cols=0):
    """The kept rows of a candidate edge table, in order, in a table of the same capacity E (zero behind them), their normalised length
    distribution and the count -- all left on the device (csrc/psdr_tables.hip k_compact_*).  aux: an int32 / float32 tensor [E, >= aux_cols]
    whose first aux_cols words per row travel along (row stride = aux.stride(0), unit column stride).
    Returns (rows_out [E, S], aux_out [E, aux_cols] int32 bits, pos [E] int32, pmf [E], cmf [E], header [2] = {count as int bits, sum})."""
    E = rows.shape[0]
    rows_out, aux_out, pos, dist = _CompactEdges.apply(rows, keep_u8, w0, wn, aux, aux_cols)
    return rows_out, aux_out, pos, dist[:E], dist[E:2 * E], dist[2 * E:]


def emitter_tables(rows, face_offset_i32, mesh_emitter_i32, emitter_i, radiance, env_weight, n_face_words):
    """mesh areas [M] + emitter_f [Ne, 8], emitter_pmf / emitter_cmf [Ne] (normalised), face_pmf / face_cmf [n_face_words] on the device
    (csrc/psdr_tables.hip k_mesh_areas, k_emitter_rows); nothing here carries a gradient (the reference detaches all of it)."""
    lib = _abi.load_hip()
    dev = rows.device
    M, Ne = mesh_emitter_i32.shape[0], emitter_i.shape[0] if emitter_i is not None else 0
    rc = rows.detach()
    out = torch.empty(M + Ne * (_abi.EMITTER_F_STRIDE + 2) + 2 * max(n_face_words, 1), dtype=torch.float32, device=dev)
    o = [0]
    def take(n):
        t = out[o[0]:o[0] + n]; o[0] += n
        return t
    area, ef, epmf, ecmf = take(M), take(Ne * _abi.EMITTER_F_STRIDE), take(Ne), take(Ne)
    fpmf, fcmf = take(max(n_face_words, 1)), take(max(n_face_words, 1))
    p = lambda t: t.data_ptr() if t is not None and t.numel() else None
    _abi.check(lib, lib.psdr_geo_emitter_tables(M, rc.data_ptr(), rc.stride(0), face_offset_i32.data_ptr(), mesh_emitter_i32.data_ptr(), Ne, p(emitter_i),
                                                p(radiance), p(env_weight), area.data_ptr(), p(ef), p(epmf), p(ecmf), fpmf.data_ptr(), fcmf.data_ptr(), _stream()))
    return area, ef.reshape(Ne, _abi.EMITTER_F_STRIDE), epmf, ecmf, fpmf, fcmf
