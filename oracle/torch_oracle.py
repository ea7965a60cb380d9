"""SECOND, independent restatement of the hot path (test infrastructure, like oracle/psdr_oracle.cpp): torch on the
CPU, fp64, automatic differentiation by torch's forward mode -- written from the behavioural spec (SURVEY.md App. A,
which cites the reference file:line for every formula), sharing NO code with the product's table builder
(psdr_cuda/scene.py) or with the C++ oracle.

What it restates
  * the table chain from raw inputs (vertices, faces, transforms, camera parameters):
        process_mesh                    src/shape/mesh.cpp:20-51        -> triangle table
        Mesh::configure                 src/shape/mesh.cpp:215-274      -> areas, face distribution, secondary edges
        edge topology                   src/shape/mesh.cpp:154-196
        Scene::configure                src/scene/scene.cpp:183-244     -> emitter table, concatenation, coplanar filter
        PerspectiveCamera::configure    src/sensor/perspective.cpp:11-111 -> matrices, inv_area, primary-edge list
  * the sampler (TEA + PCG32, src/core/sampler.cpp:7-40)
  * closest hit by brute force over all triangles (cuda/psdr_cuda.cu:9-45), hit reconstruction (scene.cpp:290-384)
  * DirectIntegrator::__Li with diffuse BSDFs and area lights (direct.cpp:47-163), the interior estimator
    (integrator.cpp:64-95), the primary-edge term (integrator.cpp:98-119, perspective.cpp:158-200) and the
    secondary-edge term (direct.cpp:207-316, scene.cpp:456-492), all in the reference's LITERAL forms.

Its only inputs from the product are what the XML / OBJ loader read (raw arrays).  Tests use it three ways
(tests/test_second_oracle.py): product tables == these tables; C++ oracle (fp64) == this renderer on the same
tables, image and derivative image; GPU == both.
"""
import math

import numpy as np
import torch
import torch.autograd.forward_ad as fwAD

F64 = torch.float64
EPS, RAY_EPS, SHADOW_EPS, EDGE_EPS = 1e-5, 1e-3, 1e-3, 1e-5
TRI_FACE_NORMALS = 0x40000000


# --------------------------------------------------------------------------- inputs
def scene_inputs(scene):
    """What the loader read, as plain float64 / int64 arrays (no tables)."""
    meshes = []
    for m in scene.m_meshes:
        tw = (m._to_world_left.double() @ m._to_world_raw.double() @ m._to_world_right.double()).detach().cpu()
        meshes.append(dict(v=m._vertex_positions_raw.detach().double().cpu(), f=m._face_indices.long().cpu(), to_world=tw,
                           face_normals=bool(m.use_face_normals), enable_edges=bool(m.enable_edges),
                           bsdf=scene.m_bsdfs.index(m.bsdf) if m.bsdf is not None else -1,
                           emitter=scene.m_emitters.index(m.m_emitter) if m.m_emitter is not None else -1))
    bsdfs = [b.reflectance.tensor().detach().double().cpu().reshape(-1) for b in scene.m_bsdfs]
    emitters = [e.radiance.t.detach().double().cpu().reshape(3) for e in scene.m_emitters]
    s = scene.m_sensors[0]
    cam = dict(fov_x=s.m_fov_x, near=s.m_near_clip, far=s.m_far_clip, to_world=s._to_world.detach().double().cpu())
    o = scene.opts
    return dict(meshes=meshes, bsdfs=bsdfs, emitters=emitters, cam=cam, width=o.width, height=o.height, sppe=o.sppe, sppse=o.sppse)


# --------------------------------------------------------------------------- tables
def edge_topology(faces):
    """mesh.cpp:154-196: key = sorted vertex pair -> (v0, v1, face0, face1 | -1, opposite vertex of face0), key order."""
    table = {}
    for fi, (a, b, c) in enumerate(faces.tolist()):
        for (p, q, r) in ((a, b, c), (b, c, a), (c, a, b)):
            k = (min(p, q), max(p, q))
            if k not in table:
                table[k] = [r, fi]
            else:
                if len(table[k]) >= 3:
                    raise RuntimeError("Edge shared by more than 2 faces")
                table[k].append(fi)
    rows = [[k[0], k[1], v[1], v[2] if len(v) > 2 else -1, v[0]] for k, v in sorted(table.items())]
    return torch.tensor(rows, dtype=torch.long).reshape(-1, 5)


def transform_pos(m, v):
    h = v @ m[:3, :3].T + m[:3, 3]
    w = v @ m[3, :3] + m[3, 3]
    return h / w.unsqueeze(-1)


def discrete(pmf):
    """pmf.cpp:10-27 on the detached masses; fp32 like the tables the kernels read."""
    p = pmf.detach().to(torch.float32)
    cmf = torch.cumsum(p, 0)
    return cmf, p, float(cmf[-1])


def build_tables(inp, mesh_transform=None):
    """All scene tables in fp64 from the raw inputs.  mesh_transform: {mesh id: 4x4 torch matrix (may carry a forward-mode
    tangent)} applied on the LEFT of that mesh's to_world (Mesh::set_transform, mesh.h:19-35)."""
    W, H = inp["width"], inp["height"]
    tri_rows, tri_mesh, mesh_bsdf, mesh_emitter, sec_rows, sec_faces = [], [], [], [], [], []
    face_off, vworld, fn_all, p0_all, areas = [0], [], [], [], []
    for i, m in enumerate(inp["meshes"]):
        tw = m["to_world"]
        if mesh_transform and i in mesh_transform:
            tw = mesh_transform[i] @ tw
        v = transform_pos(tw, m["v"])
        f = m["f"]
        p0, e1, e2 = v[f[:, 0]], v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
        c = torch.cross(e1, e2, dim=1)
        a2 = c.norm(dim=1)
        nv_num = torch.zeros_like(v).index_add(0, f.reshape(-1), c.repeat_interleave(3, 0))
        nv_den = torch.zeros(v.shape[0], dtype=v.dtype).index_add(0, f.reshape(-1), a2.repeat_interleave(3))
        nv = nv_num / nv_den.unsqueeze(-1)
        nv = nv / nv.norm(dim=1, keepdim=True)
        fn, area = c / a2.unsqueeze(-1), 0.5 * a2
        tri_rows.append(torch.cat([p0, e1, e2, nv[f[:, 0]], nv[f[:, 1]], nv[f[:, 2]], fn, area.unsqueeze(-1), torch.zeros(len(f), 2, dtype=v.dtype)], 1))
        tri_mesh.append(torch.full((len(f),), i | (TRI_FACE_NORMALS if m["face_normals"] else 0), dtype=torch.int32))
        mesh_bsdf.append(m["bsdf"]); mesh_emitter.append(m["emitter"])
        vworld.append(v); fn_all.append(fn); p0_all.append(p0); areas.append(area)
        if m["enable_edges"]:
            e = edge_topology(f)
            m["_edges"] = e
            is_b = e[:, 3] < 0
            n0 = fn[e[:, 2]]
            n1 = fn[e[:, 3].clamp(min=0)] * (~is_b).unsqueeze(-1)
            keep = ((n0 * n1).sum(1) < 1.0 - EDGE_EPS).detach()
            row = torch.cat([v[e[:, 0]], v[e[:, 1]] - v[e[:, 0]], n0, n1, v[e[:, 4]], is_b.double().unsqueeze(-1)], 1)
            sec_rows.append(row[keep])
            gf = torch.stack([e[:, 2] + face_off[-1], torch.where(is_b, e[:, 3], e[:, 3] + face_off[-1])], 1)
            sec_faces.append(gf[keep])
        face_off.append(face_off[-1] + len(f))
    tb = dict(width=W, height=H, num_meshes=len(inp["meshes"]), num_bsdfs=len(inp["bsdfs"]), num_tris=face_off[-1], face_offset=face_off)
    tb["tri_info"] = torch.cat(tri_rows)
    tb["tri_mesh"] = torch.cat(tri_mesh)
    tb["tri_uv"] = None
    tb["mesh_bsdf"] = torch.tensor(mesh_bsdf, dtype=torch.int32)
    tb["mesh_emitter"] = torch.tensor(mesh_emitter, dtype=torch.int32)
    # BSDF records + texels: constant diffuse reflectances (scene.cpp: BSDF list order)
    rec, off = [], 0
    for b in inp["bsdfs"]:
        rec.append([0, off, 1, 1] + [0, 1, 1] * 4)
        off += 3
    tb["bsdf_rec"] = torch.tensor(rec, dtype=torch.int32)
    tb["texels"] = torch.cat(inp["bsdfs"])
    tb["material_mask"] = 1
    # emitters: weight = area * luminance, normalised (area.cpp:10-16, scene.cpp:183-196)
    Ne = len(inp["emitters"])
    weights, ef, ei, cmfs, pmfs, coff = [], [], [], [], [], 0
    for k, rad in enumerate(inp["emitters"]):
        mi = mesh_emitter.index(k)
        A = float(np.float32(areas[mi].detach().sum().item()))            # m_total_area is a host float (mesh.cpp:239)
        lum = float(rad[0] * .2126 + rad[1] * .7152 + rad[2] * .0722)
        weights.append(A * lum)
        cmf, pmf, s = discrete(areas[mi])
        ef.append([float(rad[0]), float(rad[1]), float(rad[2]), 0.0, 1.0 / A, s, 0.0, 0.0])
        ei.append([mi, face_off[mi], len(inp["meshes"][mi]["f"]), coff])
        cmfs.append(cmf); pmfs.append(pmf); coff += len(cmf)
    ecmf, epmf, esum = discrete(torch.tensor(weights, dtype=F64))
    for k in range(Ne):
        ef[k][3] = float(np.float32(weights[k]) * (np.float32(1.0) / np.float32(esum)))
    tb["emitter_f"], tb["emitter_i"] = torch.tensor(ef, dtype=F64), torch.tensor(ei, dtype=torch.int32)
    tb["emitter_rad"] = torch.stack(list(inp["emitters"]))
    tb["face_cmf"], tb["face_pmf"] = torch.cat(cmfs), torch.cat(pmfs)
    tb["emitter_cmf"], tb["emitter_pmf"], tb["emitter_sum"], tb["num_emitters"] = ecmf, epmf, esum, Ne
    tb.update(env_emitter=-1, env_tex=[0, 0, 0], env_reso=[0, 0], env_f=None, env_cmf=None, env_pmf=None, env_sum=0.0)
    # secondary edges: concatenation in mesh order, pmf = |e1| (scene.cpp:219-244)
    if inp["sppse"] > 0 and sec_rows and sum(len(r) for r in sec_rows) > 0:
        se = torch.cat(sec_rows)
        cmf, pmf, s = discrete(se[:, 3:6].norm(dim=1))
        tb.update(sec_edge=se, sec_cmf=cmf, sec_pmf=pmf, sec_sum=s, num_sec_edges=len(se), sec_edge_faces=torch.cat(sec_faces).to(torch.int32))
    else:
        tb.update(sec_edge=None, sec_cmf=None, sec_pmf=None, sec_sum=0.0, num_sec_edges=0, sec_edge_faces=None)
    # camera (perspective.cpp:11-33, transform.h:45-60)
    c = inp["cam"]
    aspect = W / H
    n_, f_ = c["near"], c["far"]
    cot = 1.0 / math.tan(math.radians(c["fov_x"] * 0.5))
    Pm = torch.tensor([[cot, 0, 0, 0], [0, cot, 0, 0], [0, 0, f_ / (f_ - n_), -n_ * f_ / (f_ - n_)], [0, 0, 1, 0]], dtype=F64)
    scale = torch.diag(torch.tensor([-0.5, -0.5 * aspect, 1.0, 1.0], dtype=F64))
    trans = torch.eye(4, dtype=F64); trans[0, 3] = -1.0; trans[1, 3] = -1.0 / aspect
    c2s = scale @ trans @ Pm
    s2c = torch.linalg.inv(c2s)
    tw = c["to_world"]
    w2s = c2s @ torch.linalg.inv(tw)
    cam_pos = transform_pos(tw, torch.zeros(1, 3, dtype=F64))[0]
    cam_dir = tw[:3, :3] @ torch.tensor([0.0, 0.0, 1.0], dtype=F64)

    def corner(x, y):
        v = s2c @ torch.tensor([x, y, 0.0, 1.0], dtype=F64)
        return v[:3] / v[3]
    v00, v10, v11, vc = corner(0, 0), corner(1, 0), corner(1, 1), corner(.5, .5)
    inv_area = float((vc.norm() ** 2) / ((v00 - v10).norm() * (v11 - v10).norm()))
    cam = torch.zeros(64, dtype=F64)
    cam[0:16], cam[16:32], cam[32:48] = s2c.reshape(-1), tw.reshape(-1), w2s.reshape(-1)
    cam[48:51], cam[51:54], cam[54] = cam_pos, cam_dir, inv_area
    tb["cam"] = cam
    # primary edges of this sensor (perspective.cpp:39-111)
    pe = []
    if inp["sppe"] > 0:
        for i, m in enumerate(inp["meshes"]):
            if not m["enable_edges"]:
                continue
            e = m["_edges"]
            valid = e[:, 3] >= 0
            fn, p0 = fn_all[i].detach(), p0_all[i].detach()
            def nrm(x): return x / x.norm(dim=1, keepdim=True)
            e0 = nrm(cam_pos - p0[e[:, 2]])
            e1 = nrm(cam_pos - p0[e[:, 3].clamp(min=0)] * valid.unsqueeze(-1))
            n0, n1 = fn[e[:, 2]], fn[e[:, 3].clamp(min=0)] * valid.unsqueeze(-1)
            d0, d1, dn = (e0 * n0).sum(1), (e1 * n1).sum(1), (n0 * n1).sum(1)
            if m["face_normals"]:
                keep = ~(valid & (((d0 < EPS) & (d1 < EPS)) | (dn > 1.0 - EPS)))
            else:
                keep = (~valid) | ((d0 > EPS) ^ (d1 > EPS))
            ek = e[keep]
            q0 = transform_pos(w2s, vworld[i][ek[:, 0]])[:, :2]
            q1 = transform_pos(w2s, vworld[i][ek[:, 1]])[:, :2]
            d = (q1 - q0).detach()
            ln = d.norm(dim=1)
            d = d / ln.unsqueeze(-1)
            pe.append(torch.cat([q0, q1, torch.stack([-d[:, 1], d[:, 0]], 1), ln.unsqueeze(-1), torch.zeros(len(ek), 1, dtype=F64)], 1))
    if pe and sum(len(r) for r in pe) > 0:
        pr = torch.cat(pe)
        cmf, pmf, s = discrete(pr[:, 6])
        tb.update(prim_edge=pr, prim_cmf=cmf, prim_pmf=pmf, prim_sum=s, num_prim_edges=len(pr))
    else:
        tb.update(prim_edge=None, prim_cmf=None, prim_pmf=None, prim_sum=0.0, num_prim_edges=0)
    return tb


def to_float_tables(tb):
    """fp32 copies (what a kernel / the C++ oracle reads) of the fp64 tables."""
    out = {}
    for k, v in tb.items():
        out[k] = v.detach().to(torch.float32) if isinstance(v, torch.Tensor) and v.dtype == F64 else (v.detach() if isinstance(v, torch.Tensor) else v)
    return out


# --------------------------------------------------------------------------- sampler
M64 = (1 << 64) - 1


def _tea(v0, v1):
    s = 0
    for _ in range(4):
        s = (s + 0x9e3779b9) & 0xffffffff
        v0 = (v0 + ((((v1 << 4) & M64) + 0xa341316c) ^ ((v1 + s) & M64) ^ ((v1 >> 5) + 0xc8013ea4))) & M64
        v1 = (v1 + ((((v0 << 4) & M64) + 0xad90777d) ^ ((v0 + s) & M64) ^ ((v0 >> 5) + 0x7e95761e))) & M64
    return (v0 + ((v1 << 32) & M64)) & M64


class Streams:
    """One PCG32 stream per slot (sampler.cpp:29-40); python integers: exact 64-bit arithmetic, small slot counts."""
    MULT = 0x5851f42d4c957f2d

    def __init__(self, slots, offset=0):
        self.state, self.inc = [], []
        for i in slots:
            seed = (int(i) + 0x853c49e6748fea9b) & M64
            initstate, initseq = _tea(seed, int(i)), _tea(int(i), seed)
            inc = ((initseq << 1) | 1) & M64
            st = (0 * self.MULT + inc) & M64
            st = (st + initstate) & M64
            st = (st * self.MULT + inc) & M64
            self.state.append(st); self.inc.append(inc)
        for _ in range(offset):
            self.next()

    def next(self):
        out = np.empty(len(self.state), dtype=np.float64)
        for k in range(len(self.state)):
            old = self.state[k]
            self.state[k] = (old * self.MULT + self.inc[k]) & M64
            xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
            rot = old >> 59
            u = ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff
            bits = (u >> 9) | 0x3f800000
            out[k] = float(np.array([bits], dtype=np.uint32).view(np.float32)[0]) - 1.0
        return torch.from_numpy(out)


# --------------------------------------------------------------------------- geometry kernels
def dot(a, b): return (a * b).sum(-1)
def normalize(a): return a / a.norm(dim=-1, keepdim=True)


def closest_hit(tb, o, d):
    """Brute force over all triangles, t in [RayEpsilon, inf), both faces (psdr_cuda.cu:9-45).  Detached."""
    T = tb["tri_info"].detach()
    o, d = o.detach(), d.detach()
    p0, e1, e2 = T[:, 0:3], T[:, 3:6], T[:, 6:9]
    h = torch.cross(d.unsqueeze(1), e2.unsqueeze(0), dim=-1)
    a = dot(e1.unsqueeze(0), h)
    f = 1.0 / a
    s = o.unsqueeze(1) - p0.unsqueeze(0)
    u = f * dot(s, h)
    q = torch.cross(s, e1.unsqueeze(0).expand_as(s), dim=-1)
    v = f * dot(d.unsqueeze(1), q)
    t = f * dot(e2.unsqueeze(0), q)
    ok = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= RAY_EPS) & torch.isfinite(t)
    t = torch.where(ok, t, torch.full_like(t, float("inf")))
    tmin, idx = t.min(dim=1)
    hit = torch.isfinite(tmin)
    r = torch.arange(len(o))
    return hit, torch.where(hit, idx, torch.zeros_like(idx)), u[r, idx], v[r, idx]


def frame(n):
    """coordinate_system, frame.h:9-28"""
    sg = torch.where(n[..., 2].detach() >= 0, 1.0, -1.0).to(n.dtype)
    a = -1.0 / (sg + n[..., 2])
    b = n[..., 0] * n[..., 1] * a
    s = torch.stack([n[..., 0] ** 2 * a * sg + 1.0, b * sg, -(n[..., 0] * sg)], -1)
    t = torch.stack([b, sg + n[..., 1] ** 2 * a, -n[..., 1]], -1)
    return s, t


class Its:
    pass


def intersect(tb, o, d, active, mode):
    """scene.cpp:290-384.  mode 'C' (detached), 'path' (D, path space), 'solid' (D, differentiable Moeller-Trumbore)."""
    hit, tri, hu, hv = closest_hit(tb, o, d)
    its = Its()
    its.valid = active & hit
    T = tb["tri_info"][tri]
    if mode == "C":
        T, o, d = T.detach(), o.detach(), d.detach()
    p0, e1, e2, n0, n1, n2, fn, area = T[:, 0:3], T[:, 3:6], T[:, 6:9], T[:, 9:12], T[:, 12:15], T[:, 15:18], T[:, 18:21], T[:, 21]
    tm = tb["tri_mesh"][tri]
    its.tri, its.mesh = tri, (tm & ~TRI_FACE_NORMALS).long()
    face = (tm & TRI_FACE_NORMALS) != 0
    its.n = fn
    its.J = torch.ones(len(o), dtype=F64)
    if mode == "solid":
        h = torch.cross(d, e2, dim=-1)
        f = 1.0 / dot(e1, h)
        s = o - p0
        u = f * dot(s, h)
        q = torch.cross(s, e1, dim=-1)
        v = f * dot(d, q)
        t = f * dot(e2, q)
        its.p = o + d * t.unsqueeze(-1)
        its.t = t
        dirv = d
    else:
        u, v = hu, hv
        if mode == "path":
            its.J = area / area.detach()
        its.p = p0 + e1 * u.unsqueeze(-1) + e2 * v.unsqueeze(-1)
        dd = its.p - o
        its.t = dd.norm(dim=-1)
        dirv = dd / its.t.unsqueeze(-1)
    sh = normalize(n0 + (n1 - n0) * u.unsqueeze(-1) + (n2 - n0) * v.unsqueeze(-1))
    sh = torch.where(face.unsqueeze(-1), fn, sh)
    its.sh_n = sh
    its.sh_s, its.sh_t = frame(sh)
    its.wi = torch.stack([dot(-dirv, its.sh_s), dot(-dirv, its.sh_t), dot(-dirv, sh)], -1)
    its.emitter = torch.where(its.valid, tb["mesh_emitter"].long()[its.mesh], torch.full_like(tri, -1))
    return its


def sample_reuse(cmf, pmf, total, u):
    """pmf.cpp:30-50 on fp32 tables; u float64 values.  Returns (index, pmf/sum, reused u)."""
    n = len(cmf)
    if n == 1:
        return torch.zeros(len(u), dtype=torch.long), torch.ones(len(u), dtype=F64), u
    c, p = cmf.double(), pmf.double()
    x = u * total
    idx = torch.searchsorted(c, x.contiguous(), right=False).clamp(max=n - 1)
    x = x - torch.where(idx > 0, c[(idx - 1).clamp(min=0)], torch.zeros_like(x))
    pi = p[idx]
    x = torch.where(pi > 0, x / pi, x).clamp(0.0, 1.0)
    return idx, pi / total, x


def concentric_disk(sx, sy):
    x, y = 2 * sx - 1, 2 * sy - 1
    zero = (x == 0) & (y == 0)
    q13 = x.abs() < y.abs()
    r, rp = torch.where(q13, y, x), torch.where(q13, x, y)
    phi = 0.25 * math.pi * rp / r
    phi = torch.where(q13, 0.5 * math.pi - phi, phi)
    phi = torch.where(zero, torch.zeros_like(phi), phi)
    return r * torch.cos(phi), r * torch.sin(phi)


# ------------------------------------------------------------------------------ BSDFs
BSDF_DIFFUSE, BSDF_ROUGHCONDUCTOR = 0, 1
SLOT_REFLECTANCE, SLOT_ALPHA_U, SLOT_ALPHA_V, SLOT_ETA, SLOT_K = range(5)


def _rec(tb, its):
    return tb["bsdf_rec"].long()[tb["mesh_bsdf"].long()[its.mesh].clamp(min=0)]


def _tex(tb, rec, slot, ch, ad):
    """Bitmap::eval of a CONSTANT (1 x 1) texture, bitmap.cpp:41-48: the texel itself (AD to it).  Larger bitmaps are not restated here."""
    tex = tb["texels"] if ad else tb["texels"].detach()
    return tex[rec[:, 1 + 3 * slot].unsqueeze(-1) + torch.arange(ch)]


def _check_constant_textures(tb):
    rec = tb["bsdf_rec"].long()
    for r in rec:
        slots = (SLOT_REFLECTANCE,) if int(r[0]) == BSDF_DIFFUSE else (SLOT_REFLECTANCE, SLOT_ALPHA_U, SLOT_ALPHA_V, SLOT_ETA, SLOT_K)
        assert all(int(r[2 + 3 * sl]) == 1 and int(r[3 + 3 * sl]) == 1 for sl in slots), "the torch oracle restates constant textures only"
    assert int(tb.get("env_emitter", -1)) < 0, "the torch oracle has no environment map"


def safe_sqrt(x): return torch.sqrt(x.clamp(min=0))


def diffuse_eval(tb, its, wo, ad):
    """diffuse.cpp:25-39"""
    rho = _tex(tb, _rec(tb, its), SLOT_REFLECTANCE, 3, ad)
    ok = (its.wi[:, 2].detach() > 0) & (wo[:, 2].detach() > 0)
    return torch.where(ok.unsqueeze(-1), rho * (wo[:, 2] / math.pi).unsqueeze(-1), torch.zeros_like(rho))


def ggx_D(m, au, av):
    """GGXDistribution::eval, ggx.cpp:15-33"""
    r = 1.0 / (math.pi * au * av * ((m[:, 0] / au) ** 2 + (m[:, 1] / av) ** 2 + m[:, 2] ** 2) ** 2)
    return torch.where((r * m[:, 2]).detach() > 1e-5, r, torch.zeros_like(r))


def ggx_g1(v, m, au, av):
    """GGXDistribution::smith_g1, ggx.cpp:77-91"""
    xy = (au * v[:, 0]) ** 2 + (av * v[:, 1]) ** 2
    r = 2.0 / (1.0 + torch.sqrt(1.0 + xy / v[:, 2] ** 2))
    r = torch.where(xy.detach() == 0, torch.ones_like(r), r)
    return torch.where((dot(v, m) * v[:, 2]).detach() <= 0, torch.zeros_like(r), r)


def ggx_visible_11(c, sx, sy):
    """GGXDistribution::sample_visible_11, ggx.cpp:94-104"""
    px, py = concentric_disk(sx, sy)
    s = 0.5 * (1.0 + c)
    a = safe_sqrt(1.0 - px * px)
    py = a + (py - a) * s                                   # lerp(a, py, s)
    z = safe_sqrt(1.0 - px * px - py * py)
    sn = safe_sqrt(1.0 - c * c)
    norm = 1.0 / (sn * py + c * z)
    return (c * py - sn * z) * norm, px * norm


def ggx_sample(wi, sx, sy, au, av):
    """GGXDistribution::sample, ggx.cpp:37-74 (sin / cos of the azimuth: frame.h:100-116, guarded at sin^2 theta <= 4 Epsilon)"""
    wp = normalize(torch.stack([au * wi[:, 0], av * wi[:, 1], wi[:, 2]], -1))
    st2 = 1.0 - wp[:, 2] ** 2
    small = st2.detach().abs() <= 4.0 * EPS
    inv = 1.0 / torch.sqrt(torch.where(small, torch.ones_like(st2), st2))
    sinp = torch.where(small, torch.zeros_like(st2), (wp[:, 1] * inv).clamp(-1.0, 1.0))
    cosp = torch.where(small, torch.ones_like(st2), (wp[:, 0] * inv).clamp(-1.0, 1.0))
    slx, sly = ggx_visible_11(wp[:, 2], sx, sy)
    slx, sly = (cosp * slx - sinp * sly) * au, (sinp * slx + cosp * sly) * av
    return normalize(torch.stack([-slx, -sly, torch.ones_like(slx)], -1))


def fresnel_conductor(eta, k, c):
    """fresnel<ad>(eta_r, eta_i, cos_theta_i), utils.h:148-164; eta, k [n, 3], c [n]"""
    c = c.unsqueeze(-1)
    c2 = c * c
    s2 = 1.0 - c2
    s4 = s2 * s2
    t1 = eta * eta - k * k - s2
    a2pb2 = safe_sqrt(t1 * t1 + 4.0 * (k * eta) ** 2)
    a = safe_sqrt(0.5 * (a2pb2 + t1))
    T1, T2 = a2pb2 + c2, 2.0 * c * a
    rs = (T1 - T2) / (T1 + T2)
    T3, T4 = a2pb2 * c2 + s4, T2 * s2
    rp = rs * (T3 - T4) / (T3 + T4)
    return 0.5 * (rs + rp)


def _alphas(tb, rec, ad):
    rough = rec[:, 0] == BSDF_ROUGHCONDUCTOR
    au, av = _tex(tb, rec, SLOT_ALPHA_U, 1, ad)[:, 0], _tex(tb, rec, SLOT_ALPHA_V, 1, ad)[:, 0]
    one = torch.ones(len(rec), dtype=F64)
    return torch.where(rough, au, one), torch.where(rough, av, one), rough          # (other lanes: a harmless alpha; their results are never selected)


def rough_eval(tb, its, wo, ad):
    """RoughConductor::__eval, roughconductor.cpp:40-57 (the value includes cos theta_o)"""
    rec = _rec(tb, its)
    au, av, _ = _alphas(tb, rec, ad)
    ci, co = its.wi[:, 2], wo[:, 2]
    ok = (ci.detach() > 0) & (co.detach() > 0)
    Hh = normalize(wo + its.wi)
    D = ggx_D(Hh, au, av)
    ok = ok & (D.detach() != 0)
    G = ggx_g1(its.wi, Hh, au, av) * ggx_g1(wo, Hh, au, av)
    res = D * G / (4.0 * ci)
    F = fresnel_conductor(_tex(tb, rec, SLOT_ETA, 3, ad), _tex(tb, rec, SLOT_K, 3, ad), dot(its.wi, Hh))
    val = F * res.unsqueeze(-1) * _tex(tb, rec, SLOT_REFLECTANCE, 3, ad)
    return torch.where(ok.unsqueeze(-1), val, torch.zeros_like(val))


def rough_pdf(tb, its, wo, ad):
    """RoughConductor::__pdf, roughconductor.cpp:61-76: the mask is computed there and NOT applied"""
    rec = _rec(tb, its)
    au, av, _ = _alphas(tb, rec, ad)
    m = normalize(wo + its.wi)
    return ggx_D(m, au, av) * ggx_g1(its.wi, m, au, av) / (4.0 * its.wi[:, 2])


def bsdf_sample(tb, its, s, active, ad):
    """Diffuse::__sample (diffuse.cpp:42-57: the LAST two of the three numbers) / RoughConductor::__sample (roughconductor.cpp:79-92: the first two).
    Returns the local direction, its pdf (keeps the derivative w.r.t. wi / alpha in D mode, as bs.pdf does) and the validity mask."""
    rec = _rec(tb, its)
    px, py = concentric_disk(s[1], s[2])
    wz = safe_sqrt(1.0 - px * px - py * py)
    wo_d = torch.stack([px, py, wz], -1)
    pdf_d = wz / math.pi
    ok_d = active & (its.wi[:, 2].detach() > 0)
    au, av, rough = _alphas(tb, rec, ad)
    if not bool(rough.any()):
        return wo_d, pdf_d, ok_d
    m = ggx_sample(its.wi, s[0], s[1], au, av)
    wo_r = m * (2.0 * dot(its.wi, m)).unsqueeze(-1) - its.wi
    pdf_r = rough_pdf(tb, its, wo_r, ad)
    ok_r = active & (its.wi[:, 2].detach() > 0) & (pdf_r.detach() != 0) & (wo_r[:, 2].detach() > 0)
    return torch.where(rough.unsqueeze(-1), wo_r, wo_d), torch.where(rough, pdf_r, pdf_d), torch.where(rough, ok_r, ok_d)


def bsdf_eval(tb, its, wo, ad):
    rough = _rec(tb, its)[:, 0] == BSDF_ROUGHCONDUCTOR
    d = diffuse_eval(tb, its, wo, ad)
    return torch.where(rough.unsqueeze(-1), rough_eval(tb, its, wo, ad), d) if bool(rough.any()) else d


def bsdf_pdf(tb, its, wo, ad):
    """Diffuse::__pdf from DETACHED cosines under its mask (diffuse.cpp:70-81); RoughConductor::__pdf"""
    rough = _rec(tb, its)[:, 0] == BSDF_ROUGHCONDUCTOR
    ok = (its.wi[:, 2].detach() > 0) & (wo[:, 2].detach() > 0)
    d = torch.where(ok, wo[:, 2].detach() / math.pi, torch.zeros_like(wo[:, 2].detach()))
    return torch.where(rough, rough_pdf(tb, its, wo, ad), d) if bool(rough.any()) else d


def sample_emitter_position(tb, u0, u1, ad):
    """scene.cpp:427-447 -> area.cpp:32-46 -> mesh.cpp:306-330 (single emitter: direct call)."""
    ef, ei = tb["emitter_f"], tb["emitter_i"].long()
    e = torch.zeros(len(u0), dtype=torch.long)
    epdf = torch.ones(len(u0), dtype=F64)
    if tb["num_emitters"] > 1:
        e, epdf, u1 = sample_reuse(tb["emitter_cmf"], tb["emitter_pmf"], tb["emitter_sum"], u1)
    assert tb["num_emitters"] == 1, "the torch oracle samples one area light"
    off, cnt, first = int(ei[0, 3]), int(ei[0, 2]), int(ei[0, 1])
    f, _, u0 = sample_reuse(tb["face_cmf"][off:off + cnt], tb["face_pmf"][off:off + cnt], float(ef[0, 5]), u0)
    t = torch.sqrt((1.0 - u0).clamp(min=0))
    a, b = 1.0 - t, t * u1
    T = tb["tri_info"][first + f]
    if not ad:
        T = T.detach()
    p = T[:, 0:3] + T[:, 3:6] * a.unsqueeze(-1) + T[:, 6:9] * b.unsqueeze(-1)
    J = T[:, 21] / T[:, 21].detach() if ad else torch.ones(len(u0), dtype=F64)
    return p, T[:, 18:21], float(ef[0, 4]) * epdf, J


def Le(tb, its, ad):
    rad = tb["emitter_rad"] if ad else tb["emitter_rad"].detach()
    r = rad[its.emitter.clamp(min=0)]
    ok = its.valid & (its.emitter >= 0) & (its.wi[:, 2].detach() > 0)
    return torch.where(ok.unsqueeze(-1), r, torch.zeros_like(r))


def direct_step(tb, rng, its, active, ad, B, L):
    """The two loops of DirectIntegrator::__Li at the vertex `its` (direct.cpp:64-160).  Returns the gathered radiance and, of the FIRST BSDF sample, the vertex
    it found, its throughput (already divided by the pdf) and whether it found one -- what the PathTracer continues with."""
    result = torch.zeros(len(active), 3, dtype=F64)
    mode1 = "path" if ad else "C"
    ef = tb["emitter_f"]
    nxt = None
    for i in range(B):
        s = [rng.next(), rng.next(), rng.next()]
        wo_s, pdf_s, a1 = bsdf_sample(tb, its, s, active, ad)
        wd = wo_s.detach()
        dir1 = its.sh_s.detach() * wd[:, 0:1] + its.sh_t.detach() * wd[:, 1:2] + its.sh_n.detach() * wd[:, 2:3]
        its1 = intersect(tb, its.p, dir1, a1, mode1)
        a_hit = a1 & its1.valid
        a1 = a_hit & (its1.emitter >= 0)
        if ad:
            wo = (its1.p - its.p) / its1.t.unsqueeze(-1)
            wl = torch.stack([dot(wo, its.sh_s), dot(wo, its.sh_t), dot(wo, its.sh_n)], -1)
            f = bsdf_eval(tb, its, wl, True)
            G = dot(its1.n, -wo).abs() / its1.t ** 2
            pdf0 = pdf_s * G.detach()
            f = f * (G * its1.J / pdf0).unsqueeze(-1)
        else:
            f = bsdf_eval(tb, its, wd, False)
            G = dot(its1.n, -dir1).abs() / its1.t ** 2
            pdf0 = pdf_s * G
            f = f / pdf_s.unsqueeze(-1)
        w = torch.full_like(pdf0, 1.0 / B)
        if L > 0:
            pe = float(ef[0, 3]) * float(ef[0, 4])
            w = w * pdf0 ** 2 / (pdf0 ** 2 + pe ** 2)
        c = Le(tb, its1, ad) * f * w.unsqueeze(-1)
        result = result + torch.where(a1.unsqueeze(-1), c, torch.zeros_like(c))
        if i == 0:
            nxt = (its1, torch.where(a_hit.unsqueeze(-1), f, torch.zeros_like(f)), a_hit)
    for _ in range(L):
        s0, s1 = rng.next(), rng.next()
        p, n, ppdf, J = sample_emitter_position(tb, s0, s1, ad)
        wo = p - its.p
        d2 = dot(wo, wo)
        dist = torch.sqrt(d2.clamp(min=0))
        wo = wo / dist.unsqueeze(-1)
        its1 = intersect(tb, its.p, wo, active, mode1)
        a1 = active & its1.valid & (its1.t.detach() > dist.detach() - SHADOW_EPS) & (its1.emitter >= 0)
        G = dot(its1.n, -wo).abs() / d2
        wl = torch.stack([dot(wo, its.sh_s), dot(wo, its.sh_t), dot(wo, its.sh_n)], -1)
        f = bsdf_eval(tb, its, wl, ad) * (G * J / ppdf).unsqueeze(-1)
        pdf1 = bsdf_pdf(tb, its, wl, ad) * (G.detach() if ad else G)
        w = torch.full_like(d2, 1.0 / L)
        if B > 0:
            w = w * ppdf ** 2 / (ppdf ** 2 + pdf1 ** 2)
        c = Le(tb, its1, ad) * f * w.unsqueeze(-1)
        result = result + torch.where(a1.unsqueeze(-1), c, torch.zeros_like(c))
    return result, nxt


def Li(tb, rng, o, d, active, ad, B=1, L=1, depth=0):
    """DirectIntegrator::__Li, direct.cpp:47-163 (depth = 0), or the PathTracer of SURVEY App. F (depth >= 1: the one-bounce step iterated from the vertex its first
    BSDF sample found; PathTracer(1) == DirectIntegrator(1, 1) sample for sample).  Diffuse and rough-conductor BSDFs with constant textures, one area light."""
    its = intersect(tb, o, d, active, "solid" if ad else "C")
    active = active & its.valid
    result = Le(tb, its, ad)
    if depth <= 0:
        c, _ = direct_step(tb, rng, its, active, ad, B, L)
        return result + torch.where(active.unsqueeze(-1), c, torch.zeros_like(c))
    beta = torch.ones(len(active), 3, dtype=F64)
    for _ in range(depth):
        c, (its1, f, a_hit) = direct_step(tb, rng, its, active, ad, 1, 1)
        result = result + torch.where(active.unsqueeze(-1), beta * c, torch.zeros_like(c))
        active = active & a_hit
        beta = torch.where(active.unsqueeze(-1), beta * f, beta)
        active = active & (beta.detach() != 0).any(-1)
        its = its1
    return result


def primary_ray(tb, sx, sy, ad):
    cam = tb["cam"] if ad else tb["cam"].detach()
    s2c, tw = cam[0:16].reshape(4, 4).detach(), cam[16:32].reshape(4, 4)
    v = torch.stack([sx, sy, torch.zeros_like(sx), torch.ones_like(sx)], -1) @ s2c.T
    dc = normalize(v[:, :3] / v[:, 3:4])
    o = (tw[:3, 3] / tw[3, 3]).expand(len(sx), 3)
    return o, dc @ tw[:3, :3].T


def zero_nonfinite(v):
    ok = torch.isfinite(v.detach())
    return torch.where(ok, v, torch.zeros_like(v))


def render(tb, spp=1, sppe=0, sppse=0, B=1, L=1, ad=False, rng_offset=(0, 0, 0), depth=0):
    """renderC (ad=False: image) or renderD (ad=True: image whose forward-mode tangent is the derivative image),
    integrator.cpp:64-119 + direct.cpp:207-221."""
    W, H = tb["width"], tb["height"]
    _check_constant_textures(tb)
    img = torch.zeros(W * H * 3, dtype=F64)
    if spp > 0:
        slots = np.arange(W * H * spp)
        rng = Streams(slots, rng_offset[0])
        pix = torch.from_numpy(slots // spp)
        j0, j1 = rng.next(), rng.next()
        sx, sy = ((pix % W) + j0) / W, ((pix // W) + j1) / H
        o, d = primary_ray(tb, sx, sy, ad)
        v = zero_nonfinite(Li(tb, rng, o, d, torch.ones(len(slots), dtype=torch.bool), ad, B, L, depth)) / spp
        img = img + torch.zeros(W * H, 3, dtype=F64).index_add(0, pix, v).reshape(-1)
    if ad and sppe > 0 and tb["num_prim_edges"] > 0:
        slots = np.arange(W * H * sppe)
        rng = Streams(slots, rng_offset[1])
        k, pmf, u = sample_reuse(tb["prim_cmf"], tb["prim_pmf"], tb["prim_sum"], rng.next())
        pe = tb["prim_edge"][k]
        nrm = pe[:, 4:6].detach()
        pdf = pmf / pe[:, 6].detach()
        p_ = pe[:, 0:2] * (1 - u).unsqueeze(-1) + pe[:, 2:4] * u.unsqueeze(-1)
        xdn = dot(p_, nrm)
        pd = p_.detach()
        ix, iy = torch.floor(pd[:, 0] * W).long(), torch.floor(pd[:, 1] * H).long()
        valid = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        Ls = []
        for sg in (-EDGE_EPS, EDGE_EPS):                       # ray_n first, then ray_p (integrator.cpp:107-112)
            o, d = primary_ray(tb, pd[:, 0] + sg * nrm[:, 0], pd[:, 1] + sg * nrm[:, 1], False)
            Ls.append(Li(tb, rng, o, d, valid, False, B, L, depth))
        dL = (Ls[0] - Ls[1]) / pdf.unsqueeze(-1)
        val = zero_nonfinite(xdn.unsqueeze(-1) * dL) / sppe
        val = val - val.detach()
        idx = torch.where(valid, iy * W + ix, torch.zeros_like(ix))
        img = img + torch.zeros(W * H, 3, dtype=F64).index_add(0, idx, torch.where(valid.unsqueeze(-1), val, torch.zeros_like(val))).reshape(-1)
    if ad and sppse > 0 and tb["num_sec_edges"] > 0:
        slots = np.arange(W * H * sppse)
        rng = Streams(slots, rng_offset[2])
        s0, s1, s2 = rng.next(), rng.next(), rng.next()
        k, pdf0, s0 = sample_reuse(tb["sec_cmf"], tb["sec_pmf"], tb["sec_sum"], s0)
        E = tb["sec_edge"][k]
        bp0 = E[:, 0:3] + E[:, 3:6] * s0.unsqueeze(-1)
        Ed = E.detach()
        e1len = Ed[:, 3:6].norm(dim=1)
        edge, edge2, p0 = Ed[:, 3:6] / e1len.unsqueeze(-1), Ed[:, 12:15] - Ed[:, 0:3], bp0.detach()
        pdf0 = pdf0 / e1len
        p2, bn, pdf2, _ = sample_emitter_position(tb, s1, s2, False)
        e = p2 - p0
        d2 = dot(e, e)
        e = e / torch.sqrt(d2.clamp(min=0)).unsqueeze(-1)
        cos_t = -dot(bn, e)
        def sgn(x): return (x > EDGE_EPS).long() - (x < -EDGE_EPS).long()
        g0, g1 = sgn(dot(Ed[:, 6:9], e)), sgn(dot(Ed[:, 9:12], e))
        valid = (cos_t > EPS) & torch.where(Ed[:, 15] != 0, g0 != 0, g0 * g1 < 0)
        bpdf = pdf0 * pdf2 * d2 / cos_t
        dirv = normalize(p2 - p0)
        its2 = intersect(tb, p0, dirv, valid, "C")
        valid = valid & its2.valid & ((its2.p - p2).norm(dim=1) < SHADOW_EPS)
        its1c = intersect(tb, p0, -dirv, valid, "C")
        valid = valid & its1c.valid
        p1 = its1c.p
        cam = tb["cam"].detach()
        q = torch.cat([p1, torch.ones(len(p1), 1, dtype=F64)], 1) @ cam[32:48].reshape(4, 4).T
        qx, qy = q[:, 0] / q[:, 3], q[:, 1] / q[:, 3]
        ix, iy = torch.floor(qx * W).long(), torch.floor(qy * H).long()
        valid = valid & (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        dc = p1 - cam[48:51]
        dist2 = dot(dc, dc)
        dc = dc / torch.sqrt(dist2).unsqueeze(-1)
        sensor_val = cam[54] / (dist2 * dot(dc, cam[51:54].expand_as(dc)) ** 3)
        co, cd = primary_ray(tb, qx, qy, True)
        its1 = intersect(tb, co, cd, valid, "solid")
        valid = valid & its1.valid & ((its1.p.detach() - p1).norm(dim=1) < SHADOW_EPS)
        dist, cos2 = (p2 - p1).norm(dim=1), dot(bn, dirv).abs()
        ev = torch.cross(edge, dirv, dim=1)
        sinphi = ev.norm(dim=1)
        proj = normalize(torch.cross(ev, bn, dim=1))
        sinphi2 = torch.cross(dirv, proj, dim=1).norm(dim=1)
        base_v = (its1c.t / dist) * (sinphi / sinphi2) * cos2
        valid = valid & (sinphi > EPS) & (sinphi2 > EPS)
        d0 = -cd.detach()
        d0l = torch.stack([dot(d0, its1c.sh_s), dot(d0, its1c.sh_t), dot(d0, its1c.sh_n)], -1)
        f = bsdf_eval(tb, its1c, d0l, False)
        corr = ((its1c.wi[:, 2] * dot(d0, its1c.n)) / (d0l[:, 2] * dot(dirv, its1c.n))).abs()
        value0 = f * corr.unsqueeze(-1) * Le(tb, its2, False) * (base_v * sensor_val / bpdf).unsqueeze(-1)
        nn = normalize(torch.cross(bn, proj, dim=1))
        value0 = value0 * (torch.sign(dot(ev, edge2)) * torch.sign(dot(ev, nn))).unsqueeze(-1)
        TA = tb["tri_info"][its2.tri]
        so, sd = its1.p, normalize(bp0 - its1.p)
        h = torch.cross(sd, TA[:, 6:9], dim=1)
        ff = 1.0 / dot(TA[:, 3:6], h)
        s_ = so - TA[:, 0:3]
        u = ff * dot(s_, h)
        qq = torch.cross(s_, TA[:, 3:6], dim=1)
        v = ff * dot(sd, qq)
        TAd = TA.detach()
        u2 = TAd[:, 0:3] + TAd[:, 3:6] * u.unsqueeze(-1) + TAd[:, 6:9] * v.unsqueeze(-1)
        res = value0 * dot(nn, u2).unsqueeze(-1)
        res = zero_nonfinite(res - res.detach()) / sppse
        idx = torch.where(valid, iy * W + ix, torch.zeros_like(ix))
        img = img + torch.zeros(W * H, 3, dtype=F64).index_add(0, idx, torch.where(valid.unsqueeze(-1), res, torch.zeros_like(res))).reshape(-1)
    return img


def translate(vec, P):
    m = torch.eye(4, dtype=F64)
    return torch.cat([torch.cat([m[:3, :3], (torch.tensor(vec, dtype=F64) * P).reshape(3, 1)], 1), m[3:4]], 0)


def render_d(inp, mesh_id, direction, **kw):
    """Image, derivative image and (tables, tangent tables) w.r.t. P, the mesh `mesh_id` translated by direction * P
    (examples/run_test.py mesh_transform), by torch forward-mode AD through THIS module's table chain and renderer."""
    with fwAD.dual_level():
        P = fwAD.make_dual(torch.zeros((), dtype=F64), torch.ones((), dtype=F64))
        tb = build_tables(inp, {mesh_id: translate(direction, P)})
        out = render(tb, ad=True, **kw)
        img, dimg = fwAD.unpack_dual(out)
        prim, tang = {}, {}
        for k, v in tb.items():
            if isinstance(v, torch.Tensor) and v.dtype == F64:
                p, t = fwAD.unpack_dual(v)
                prim[k], tang[k] = p.detach().clone(), (None if t is None else t.detach().clone())
            else:
                prim[k] = v
        return img.detach().reshape(-1, 3), (torch.zeros_like(img) if dimg is None else dimg.detach()).reshape(-1, 3), prim, tang


def render_d_from_tables(tb, tangents, **kw):
    """renderD on GIVEN tables and tangent tables (dict name -> tensor or None; e.g. fp32-rounded copies, the very
    numbers another implementation consumes): image and derivative image."""
    with fwAD.dual_level():
        d = {}
        for k, v in tb.items():
            if isinstance(v, torch.Tensor) and v.dtype in (torch.float32, F64):
                t = tangents.get(k)
                d[k] = fwAD.make_dual(v.double(), t.double()) if t is not None else v.double()
            else:
                d[k] = v
        img, dimg = fwAD.unpack_dual(render(d, ad=True, **kw))
        return img.detach().reshape(-1, 3), (torch.zeros_like(img) if dimg is None else dimg.detach()).reshape(-1, 3)
