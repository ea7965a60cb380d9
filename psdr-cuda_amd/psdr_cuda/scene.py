"""Scene objects and Scene::configure(): builds every device table the HIP kernels gather from.

Host-side mirror of the reference's L2/L3 layers.  All O(V+T) work that the
reference does with Enoki DiffArrays (world transform, per-face info, vertex normals, edge
records, camera matrices, distributions) is done here with torch ops on the render device, so the
chain  user parameters -> tables  stays differentiable by torch autograd, while the O(W*H*spp)
work is done by the hand-written HIP kernels behind include/psdr_hip.h.
"""
import math
import os
import time
import xml.etree.ElementTree as ET

import numpy as np
import torch

import enoki as ek
from . import _abi
from . import tables_native
from .core import (Object, RenderOption, Bitmap1fD, Bitmap3fD, DiscreteDistribution, psdr_assert, PositionSampleC, PositionSampleD,
                   FloatC, FloatD, Vector2fD, Vector3fC, Vector3fD, Matrix4fD, IntC)

Epsilon = 1e-5
EdgeEpsilon = 1e-5


def _dev():
    return ek.default_device()


def _mat(x):
    """4x4 float32 tensor on the render device from a shim matrix / tensor / nested list."""
    if isinstance(x, ek.ArrayBase):
        t = x.t
    elif isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(np.asarray(x, dtype=np.float32))
    return t.to(device=_dev(), dtype=torch.float32).reshape(4, 4)


def transform_pos(m, v):
    """reference include/psdr/core/transform.h:84-88"""
    h = v @ m[:3, :3].T + m[:3, 3]
    w = v @ m[3, :3] + m[3, 3]
    return h / w.unsqueeze(-1)


def transform_dir(m, v):
    return v @ m[:3, :3].T


def _normalize(v):
    return v / torch.sqrt((v * v).sum(dim=-1, keepdim=True))


# ------------------------------------------------------------------------------ BSDFs
class BSDF(Object):
    _type_name = "BSDF"

    def anisotropic(self):
        return False


class Diffuse(BSDF):
    """reference include/psdr/bsdf/diffuse.h, src/bsdf/diffuse.cpp"""
    _type_name = "DiffuseBSDF"

    def __init__(self, reflectance=None):
        super().__init__()
        self.reflectance = reflectance if isinstance(reflectance, Bitmap3fD) else Bitmap3fD(
            0.5 if reflectance is None else reflectance)

    def to_string(self):
        return "Diffuse[id=%s]" % self.id


DiffuseBSDF = Diffuse


class RoughConductor(BSDF):
    """reference include/psdr/bsdf/roughconductor.h, src/bsdf/roughconductor.cpp"""
    _type_name = "RoughConductorBSDF"

    def __init__(self, alpha=0.1, eta=(0.0, 0.0, 0.0), k=(1.0, 1.0, 1.0)):
        super().__init__()
        self.alpha_u = Bitmap1fD(alpha)
        self.alpha_v = Bitmap1fD(alpha)
        self.eta = Bitmap3fD(eta)
        self.k = Bitmap3fD(k)
        self.specular_reflectance = Bitmap3fD(1.0)
        self.m_anisotropic = False     # roughconductor.h:12-18: single-alpha constructors are isotropic

    def anisotropic(self):
        return self.m_anisotropic

    def to_string(self):
        return "RoughConductor[id=%s]" % self.id


RoughConductorBSDF = RoughConductor


# ---------------------------------------------------------------------------- emitters
class Emitter(Object):
    _type_name = "Emitter"
    _weight_host, _weight_dev = 1.0, None

    # the sampling weight (emitter.h:27; normalised by Scene.configure, scene.cpp:183-196) stays on the device after a native configure():
    # one element of the emitter distribution, read when somebody asks
    @property
    def m_sampling_weight(self):
        if self._weight_dev is not None:
            self._weight_host, self._weight_dev = float(self._weight_dev.item()), None
        return self._weight_host

    @m_sampling_weight.setter
    def m_sampling_weight(self, v):
        self._weight_host, self._weight_dev = float(v), None


class AreaLight(Emitter):
    """reference include/psdr/emitter/area.h, src/emitter/area.cpp:10-62"""
    _type_name = "AreaLight"

    def __init__(self, radiance, mesh):
        super().__init__()
        r = np.asarray(radiance, dtype=np.float32).reshape(-1)
        if r.size == 1:
            r = np.repeat(r, 3)
        self.radiance = Vector3fD([float(r[0]), float(r[1]), float(r[2])])
        self.m_mesh = mesh
        self.m_sampling_weight = 1.0
        self.m_ready = False

    def _luminance_t(self):
        psdr_assert(ek.slices(self.radiance) == 1)
        r = self.radiance.t.detach().reshape(3)
        return (r[0] * .2126 + r[1] * .7152 + r[2] * .0722).reshape(1)

    def configure(self, lum=None):
        """lum: the luminance of the radiance when the caller read it back already (Scene.configure batches its reads)"""
        psdr_assert(self.m_mesh is not None and self.m_mesh.m_ready)
        if lum is None:
            lum = float(self._luminance_t().item())
        self.m_sampling_weight = self.m_mesh.m_total_area * lum
        self.m_ready = True

    def to_string(self):
        return "AreaLight[radiance = %s, sampling_weight = %g]" % (self.radiance.numpy().tolist(), self.m_sampling_weight)


class EnvironmentMap(Emitter):
    """reference include/psdr/emitter/envmap.h:11-58, src/emitter/envmap.cpp:10-27 (configure).
    eval / sample_position / sample_position_pdf run per sample inside the kernels
    (csrc/psdr_device.h env_*); this class owns the parameters and builds the tables."""
    _type_name = "EnvironmentMap"

    def __init__(self, file_name=None):
        super().__init__()
        self.radiance = Bitmap3fD()
        if file_name is not None:
            self.radiance.load_openexr(file_name)
        self.scale = FloatD(1.0)
        d = _dev()
        self._to_world_raw = torch.eye(4, device=d)
        self._to_world_left = torch.eye(4, device=d)
        self.m_sampling_weight = 1.0          # never reset by configure (emitter.h:27, envmap.cpp:10-27)
        self.m_lower = self.m_upper = None    # scene AABB + margin, set once by Scene.configure
        self.m_mesh = None                    # the bounding mesh
        self.m_ready = False

    @property
    def to_world(self):
        return Matrix4fD._wrap(self._to_world_raw)

    def set_transform(self, mat):
        self._to_world_left = _mat(mat)
        self.m_ready = False

    def configure(self):
        w, h = self.radiance.resolution
        psdr_assert(w > 1 and h > 1)
        width, height = (w - 1) << 1, (h - 1) << 1
        d = _dev()
        idx = torch.arange(width * height, device=d)
        cx, cy = idx // height, idx % height          # HyperCubeDistribution<2> cell order, cube_distrb.cpp:19-26
        uv = torch.stack([(cx.to(torch.float32) + .5) * np.float32(1.0 / width),
                          (cy.to(torch.float32) + .5) * np.float32(1.0 / height)], dim=-1)
        with torch.no_grad():
            val = self.radiance.eval(uv, False).t.detach()
        theta = (cy.to(torch.float32) + .5) * np.float32(np.pi / height)
        lum = val[:, 0] * .2126 + val[:, 1] * .7152 + val[:, 2] * .0722
        self._cell_reso = (width, height)
        self._cell_distrb = DiscreteDistribution()
        self._cell_distrb.init((lum * torch.sin(theta)).to(torch.float32))
        self._to_world = self._to_world_left @ self._to_world_raw
        self._from_world = torch.linalg.inv(self._to_world)
        self.m_ready = True

    def record(self):
        """env_f (include/psdr_hip.h PSDR_ENV_*), differentiable w.r.t. the transform and the scale."""
        sc = self.scale.t.reshape(-1)[:1].to(torch.float32)
        return torch.cat([self._from_world[:3, :3].reshape(-1), self._to_world[:3, :3].detach().reshape(-1), sc,
                          self.m_lower.reshape(3), self.m_upper.reshape(3),
                          torch.zeros(_abi.ENV_WORDS - 25, device=sc.device)]).contiguous()

    def to_string(self):
        return "EnvironmentMap[sampling_weight = %g]" % self.m_sampling_weight


# ----------------------------------------------------------------------------- sensors
class Sensor(Object):
    _type_name = "Sensor"

    def __init__(self):
        super().__init__()
        self._to_world = torch.eye(4, dtype=torch.float32, device=_dev())
        self.m_enable_edges = False

    @property
    def to_world(self):
        return Matrix4fD._wrap(self._to_world)

    @to_world.setter
    def to_world(self, m):
        self._to_world = _mat(m)


def perspective(fov, near_, far_):
    """reference include/psdr/core/transform.h:45-60"""
    recip = 1.0 / (far_ - near_)
    cot = 1.0 / math.tan(math.radians(fov * 0.5))
    m = np.diag([cot, cot, far_ * recip, 0.0])
    m[2, 3] = -near_ * far_ * recip
    m[3, 2] = 1.0
    return m


def look_at(origin, target, up):
    """reference transform.h:68-79"""
    o, t, u = (np.asarray(x, dtype=np.float64) for x in (origin, target, up))
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross(u, d); left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


_NONZERO_STATIC = {}


def _has_nonzero_static(device):
    """Probed ONCE per device type on a dummy mask: whether torch.nonzero_static serves this device.  The real call is never wrapped in
    try/except -- an asynchronous device error that surfaces there must not be swallowed by a fallback."""
    key = device.type
    if key not in _NONZERO_STATIC:
        ok = hasattr(torch, "nonzero_static")
        if ok:
            try:
                torch.nonzero_static(torch.ones(2, dtype=torch.bool, device=device), size=2)
            except (RuntimeError, NotImplementedError):
                ok = False
        _NONZERO_STATIC[key] = ok
    return _NONZERO_STATIC[key]


def compact_indices(keep, count):
    """Indices of the True entries of `keep`, in order, when their number is already known on the host: no device-to-host read
    (a boolean-mask index reads the count back)."""
    if count == 0:
        return torch.zeros(0, dtype=torch.long, device=keep.device)
    if _has_nonzero_static(keep.device):
        return torch.nonzero_static(keep, size=int(count)).reshape(-1)
    pos = torch.cumsum(keep.to(torch.long), dim=0) - 1
    out = torch.empty(int(count) + 1, dtype=torch.long, device=keep.device)
    out.scatter_(0, torch.where(keep, pos, torch.full_like(pos, int(count))), torch.arange(keep.shape[0], device=keep.device))
    return out[:int(count)]


class PerspectiveCamera(Sensor):
    """reference include/psdr/sensor/perspective.h, src/sensor/perspective.cpp"""
    _type_name = "PerspectiveCamera"

    def __init__(self, fov_x, near=0.1, far=1e4):
        super().__init__()
        self.m_fov_x, self.m_near_clip, self.m_far_clip = float(fov_x), float(near), float(far)

    def to_string(self):
        return "PerspectiveCamera"

    def configure(self, scene):
        """perspective.cpp:11-112 -> dict of tensors for this sensor (one call: reads the edge count and the distribution sum back itself;
        Scene.configure uses configure_begin / configure_finish around its batched read-back)."""
        st = self.configure_begin(scene)
        if st.get("kind") == "device":
            return st["out"]
        count = int(st["count_t"].item()) if st["count_t"] is not None else 0
        out = self.configure_finish(st, count)
        if st.get("sum_t") is not None:
            self.configure_sum(st, float(st["sum_t"].item()))
        return out

    def configure_begin(self, scene):
        """Everything up to the silhouette mask of the primary-edge list, enqueued without a read-back.  count_t: number of kept edges (device)."""
        W, H = scene.opts.width, scene.opts.height
        aspect = float(W) / float(H)
        tw = self._to_world
        # what depends on the pose only is looked at again when the pose tensor changed (the determinant check reads a value back to
        # the host, the inverse of a pose without gradient is a constant), what depends on the lens only when the lens changed
        pose_key = (id(tw), tw.data_ptr(), tw._version)
        if getattr(self, "_pose_key", None) != pose_key:
            # ONE 36-byte read-back and a host determinant: torch.det on the device is an LU factorisation plus a dozen pivot-sign kernels in front of the same read
            det = float(np.linalg.det(tw.detach()[:3, :3].cpu().double().numpy()))
            psdr_assert(abs(det - 1.0) < 1e-4, "Sensor transformation should not involve scaling!")   # sensor.cpp:8-12
            self._pose_key, self._pose_inv = pose_key, (None if tw.requires_grad else torch.linalg.inv(tw))
            self._pose_alive = tw                  # the keyed tensor stays alive: its id / allocator block cannot be recycled under the key
        lens_key = (self.m_fov_x, self.m_near_clip, self.m_far_clip, aspect, str(tw.device))
        if getattr(self, "_lens_key", None) != lens_key:
            c2s = (np.diag([-0.5, -0.5 * aspect, 1.0, 1.0]) @
                   np.array([[1, 0, 0, -1.0], [0, 1, 0, -1.0 / aspect], [0, 0, 1, 0], [0, 0, 0, 1.0]]) @
                   perspective(self.m_fov_x, self.m_near_clip, self.m_far_clip))
            s2c = np.linalg.inv(c2s)
            self._lens_key = lens_key
            self._lens = (c2s, s2c, torch.as_tensor(c2s, dtype=torch.float32, device=tw.device), torch.as_tensor(s2c, dtype=torch.float32, device=tw.device))
        c2s, s2c, c2s_t, s2c_t = self._lens
        # (inv_ex: torch.linalg.inv reads its error flag back -- a device-to-host read per configure() when the pose carries a gradient; the
        # pose was checked when it changed: the determinant test above)
        # a pose without a gradient under an unchanged lens: world_to_sample, the camera record and the 22 words the edge kernel reads stand
        cam_key = (pose_key, lens_key) if self._pose_inv is not None else None
        cached = getattr(self, "_cam_cache", None)
        cached = cached if cached is not None and cam_key is not None and cached[0] == cam_key else None
        if cached is not None:
            w2s, cam_pos, cam_dir = cached[1], cached[2], cached[3]
        else:
            w2s = c2s_t @ (self._pose_inv if self._pose_inv is not None else torch.linalg.inv_ex(tw)[0])
            cam_pos = tw[:3, 3] / tw[3, 3]            # transform_pos(to_world, origin)
            cam_dir = tw[:3, 2]                       # transform_dir(to_world, (0, 0, 1))

        if getattr(self, "_cam_tail_key", None) != lens_key:
            def tp(x, y):
                v = np.array([x, y, 0.0, 1.0]); r = s2c @ v
                return r[:3] / r[3]
            v00, v10, v11, vc = tp(0, 0), tp(1, 0), tp(1, 1), tp(.5, .5)
            inv_area = float(np.dot(vc, vc) / (np.linalg.norm(v00 - v10) * np.linalg.norm(v11 - v10)))
            self._cam_tail_key = lens_key
            self._cam_tail = torch.tensor([inv_area] + [0.0] * (_abi.CAM_WORDS - 55), dtype=torch.float32, device=tw.device)
        # words 0..15 sample_to_camera, 16..31 to_world, 32..47 world_to_sample, 48..50 position, 51..53 direction, 54 1 / film area
        if cached is not None:
            cam = cached[4]
        else:
            cam = torch.cat([s2c_t.reshape(-1), tw.detach().reshape(-1), w2s.detach().reshape(-1), cam_pos.detach(), cam_dir.detach(), self._cam_tail]).contiguous()
            self._cam_cache = (cam_key, w2s, cam_pos, cam_dir, cam) if cam_key is not None else None
        out = {"cam": cam, "cam_to_world": tw, "prim_edge": None, "prim_cmf": None, "prim_pmf": None,
               "prim_sum": 0.0, "num_prim_edges": 0, "prim_edge_z": None}

        # ---- primary-edge list, perspective.cpp:39-111
        self.m_enable_edges = False
        st = {"out": out, "count_t": None, "sum_t": None, "vis": getattr(scene.opts, "primary_edge_vis_check", False)}
        if scene.opts.sppe > 0:
            bt = scene._batch
            ei = bt["tp"]["edges"]
            if ei is not None and tables_native.available(bt["v_world"]):
                # every candidate edge in ONE launch (csrc/psdr_tables.hip k_prim_edges): film records, 1 / depth rows, silhouette test
                def film_records(v, m, edges):
                    edges = edges.long()
                    q0, q1 = transform_pos(m, v.index_select(0, edges[:, 0]))[:, :2], transform_pos(m, v.index_select(0, edges[:, 1]))[:, :2]
                    e = (q1 - q0).detach()
                    ln = torch.sqrt((e * e).sum(-1))
                    e = e / ln.unsqueeze(-1)
                    return torch.cat([q0, q1, torch.stack([-e[:, 1], e[:, 0]], dim=-1), ln.unsqueeze(-1), torch.zeros_like(ln).unsqueeze(-1)], dim=-1)
                r8, z4, keep8 = tables_native.prim_edges(bt["v_world"], w2s, bt["tri_info"], bt["tp"]["edges_i32"], bt["tp"]["edge_face_normals_u8"],
                                                        cam_pos, cam_dir, film_records, cam22=cam[32:54])
                # the kept edges first, in a table of the same capacity; their number and the sum of their lengths stay on the device
                # (csrc/psdr_tables.hip k_compact_*): no read-back, no host-sized gather
                pe, zs, _pos, pmf, cmf, hdr = tables_native.compact_edges(r8, keep8, 6, 1, aux=z4, aux_cols=4)
                E = int(pe.shape[0])
                out.update(prim_edge=pe, prim_cmf=cmf, prim_pmf=pmf, prim_sum=1.0, num_prim_edges=E, prim_header=hdr)
                if st["vis"]:
                    out["prim_edge_z"] = zs.view(torch.float32)
                self.m_enable_edges = True
                st.update(kind="device")
            elif ei is not None:
                tinfo, facen = bt["tri_info"], bt["tp"]["edge_face_normals"]
                valid = ei[:, 3] >= 0
                f1 = torch.where(valid, ei[:, 3], torch.zeros_like(ei[:, 3]))
                f0 = ei[:, 2]
                vm = valid.unsqueeze(-1).to(torch.float32)
                e0 = _normalize(cam_pos - tinfo[f0, 0:3])
                # masked gather reads zeros for invalid lanes: normalize(cam_pos - 0)
                e1 = _normalize(cam_pos - tinfo[f1, 0:3] * vm)
                n0 = tinfo[f0, 18:21]
                n1 = tinfo[f1, 18:21] * vm
                d0, d1, dn = (e0 * n0).sum(-1), (e1 * n1).sum(-1), (n0 * n1).sum(-1)
                keep_face = ~(valid & (((d0 < Epsilon) & (d1 < Epsilon)) | (dn > 1.0 - Epsilon)))      # face-normal meshes
                keep_smooth = (~valid) | ((d0 > Epsilon) ^ (d1 > Epsilon))                             # smooth-shaded meshes
                keep = torch.where(facen, keep_face, keep_smooth).detach()
                st.update(kind="torch", keep=keep, count_t=keep.sum().reshape(1), w2s=w2s, cam_pos=cam_pos, cam_dir=cam_dir, bt=bt)
        return st

    def configure_finish(self, st, count):
        """The compacted primary-edge table from configure_begin's state and the number of kept edges.  st["sum_t"]: the sum of the edge
        lengths on the device (configure_sum stores it once read back)."""
        out = st["out"]
        if st["count_t"] is None or count <= 0:
            return out
        bt_keep = st["keep"]
        idx = compact_indices(bt_keep, count)
        if st["kind"] == "native":
            pe = st["r8"][idx].contiguous()
            zs = st["z4"][idx]
        else:
            ei, vpos = st["bt"]["tp"]["edges"], st["bt"]["v_world"]
            w2s, cam_pos, cam_dir = st["w2s"], st["cam_pos"], st["cam_dir"]
            info = ei[idx]
            p0 = vpos[info[:, 0]]
            p1 = vpos[info[:, 1]]
            q0f, q1f = transform_pos(w2s, p0), transform_pos(w2s, p1)
            q0, q1 = q0f[:, :2], q1f[:, :2]
            # PSDR_PRIMARY_EDGE_VIS_CHECK: (1 / camera-space depth of the two end points, the adjacent faces as int bits).
            # 1 / depth is affine along the film segment like the reference's sample-space z, but keeps fp32
            # precision (sample-space z = 1 - near / depth loses 3-4 digits at near = 0.1, depth = 500)
            cd = _normalize(cam_dir.detach())
            iz = [1.0 / ((pp.detach() - cam_pos.detach()) * cd).sum(-1) for pp in (p0, p1)]
            fb = [torch.where(info[:, 3] >= 0, info[:, c], info[:, 2]).to(torch.int32).view(torch.float32) for c in (2, 3)]
            zs = torch.stack(iz + fb, dim=-1)
            e = (q1 - q0).detach()
            ln = torch.sqrt((e * e).sum(-1))
            e = e / ln.unsqueeze(-1)
            nrm = torch.stack([-e[:, 1], e[:, 0]], dim=-1)
            pe = torch.cat([q0, q1, nrm, ln.unsqueeze(-1), torch.zeros_like(ln).unsqueeze(-1)], dim=-1).contiguous()
        pmf = pe[:, 6].detach().to(torch.float32).contiguous()
        st["sum_t"] = pmf.sum().reshape(1)
        d = DiscreteDistribution(); d.init(pmf, total=0.0)          # m_sum: configure_sum
        st["distrb"] = d
        out.update(prim_edge=pe, prim_cmf=d.m_cmf, prim_pmf=d.m_pmf, prim_sum=0.0, num_prim_edges=int(pe.shape[0]))
        if st["vis"]:
            out["prim_edge_z"] = zs.contiguous()
        self.m_enable_edges = True
        return out

    def configure_sum(self, st, total):
        st["distrb"].m_sum = float(total)
        st["out"]["prim_sum"] = float(total)


PositionSample = PositionSampleD      # kept as an alias (older name of this build)


class BoundarySegSampleDirect:
    """records.h:35-44: pdf, is_valid, p0 (differentiable), edge, edge2, p2, n."""
    pdf = is_valid = p0 = edge = edge2 = p2 = n = None


# ------------------------------------------------------------------------------- mesh
def load_obj(fname):
    """Minimal OBJ reader standing in for tinyobj::LoadObj(triangulate=true)
    (reference src/shape/mesh.cpp:62-138): v / vt / f with v, v/vt, v//vn, v/vt/vn corner
    forms, negative indices, polygons fan-triangulated as (a,b,c),(a,c,d) like tinyobj does
    for convex quads.  Returns (verts [V,3] f32, uvs [n,2] f32 or None, faces [F,3] i32, uv_faces or None)."""
    verts, uvs, faces, uvfaces = [], [], [], []
    with open(fname, "r") as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            parts = line.split()
            if not parts:
                continue
            tag = parts[0]
            if tag == "v":
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif tag == "vt":
                uvs.append((float(parts[1]), float(parts[2]) if len(parts) > 2 else 0.0))
            elif tag == "f":
                vi, ti = [], []
                for c in parts[1:]:
                    sp = c.split("/")
                    i = int(sp[0]); vi.append(i - 1 if i > 0 else len(verts) + i)
                    if len(sp) > 1 and sp[1]:
                        j = int(sp[1]); ti.append(j - 1 if j > 0 else len(uvs) + j)
                    else:
                        ti.append(-1)
                for k in range(1, len(vi) - 1):
                    faces.append((vi[0], vi[k], vi[k + 1]))
                    uvfaces.append((ti[0], ti[k], ti[k + 1]))
    v = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
    fa = np.asarray(faces, dtype=np.int32).reshape(-1, 3)
    if uvs:
        return v, np.asarray(uvs, dtype=np.float32).reshape(-1, 2), fa, np.asarray(uvfaces, dtype=np.int32).reshape(-1, 3)
    return v, None, fa, None


def build_edge_indices(faces, fname="<mesh>"):
    """Edge topology, reference src/shape/mesh.cpp:154-196: std::map keyed by the sorted vertex
    pair (iteration = lexicographic order); value = [opposite vertex of the first face, face ids].
    Returns int32 [E,5] = v0, v1, face0, face1 (-1 on boundary), opposite vertex of face0."""
    F = faces.shape[0]
    if F == 0:
        return np.zeros((0, 5), dtype=np.int32)
    a = faces[:, [0, 1, 2]].reshape(-1)
    b = faces[:, [1, 2, 0]].reshape(-1)
    c = faces[:, [2, 0, 1]].reshape(-1)
    fid = np.repeat(np.arange(F, dtype=np.int64), 3)
    k0, k1 = np.minimum(a, b).astype(np.int64), np.maximum(a, b).astype(np.int64)
    order = np.lexsort((np.arange(3 * F), k1, k0))           # stable within a key: insertion order
    k0s, k1s, cs, fs = k0[order], k1[order], c[order], fid[order]
    new = np.ones(3 * F, dtype=bool)
    new[1:] = (k0s[1:] != k0s[:-1]) | (k1s[1:] != k1s[:-1])
    start = np.nonzero(new)[0]
    count = np.diff(np.append(start, 3 * F))
    if (count > 2).any():
        raise RuntimeError("Edge shared by more than 2 faces: " + fname)
    E = start.shape[0]
    out = np.empty((E, 5), dtype=np.int32)
    out[:, 0], out[:, 1] = k0s[start], k1s[start]
    out[:, 2] = fs[start]
    out[:, 4] = cs[start]
    two = count == 2
    out[:, 3] = -1
    out[two, 3] = fs[start[two] + 1]
    if (out[two, 2] == out[two, 3]).any():
        raise RuntimeError("Duplicated faces: " + fname)
    return out


def process_mesh(verts, faces):
    """reference src/shape/mesh.cpp:20-51 -> (triangle_info [F,22], vertex_normals [V,3])."""
    f0, f1, f2 = faces[:, 0].long(), faces[:, 1].long(), faces[:, 2].long()
    # (index_select, not verts[f0]: its backward is an atomic index_add; advanced indexing goes back through torch's sort-based index_put(accumulate),
    #  0.25 ms a call on the GPU -- 22 of them in one forward-mode step through the surface, tools/r06_index_sites.py)
    p0 = verts.index_select(0, f0)
    e1 = verts.index_select(0, f1) - p0
    e2 = verts.index_select(0, f2) - p0
    fn = torch.cross(e1, e2, dim=-1)
    fa = torch.sqrt((fn * fn).sum(-1))
    vn = torch.zeros_like(verts)
    vw = torch.zeros(verts.shape[0], dtype=verts.dtype, device=verts.device)
    for fi in (f0, f1, f2):
        vn = vn.index_add(0, fi, fn)
        vw = vw.index_add(0, fi, fa)
    vn = _normalize(vn / vw.unsqueeze(-1))
    n0, n1, n2 = vn.index_select(0, f0), vn.index_select(0, f1), vn.index_select(0, f2)
    fn = fn / fa.unsqueeze(-1)
    fa = fa * 0.5
    return torch.cat([p0, e1, e2, n0, n1, n2, fn, fa.unsqueeze(-1)], dim=-1), vn


class Mesh(Object):
    """reference include/psdr/shape/mesh.h, src/shape/mesh.cpp"""
    _type_name = "Mesh"

    def __init__(self):
        super().__init__()
        self.m_ready = False
        self.use_face_normals = False
        self.m_has_uv = False
        self.enable_edges = True
        d = _dev()
        self._to_world_raw = torch.eye(4, device=d)
        self._to_world_left = torch.eye(4, device=d)
        self._to_world_right = torch.eye(4, device=d)
        self.bsdf = None
        self.m_emitter = None
        self.num_vertices = 0
        self.num_faces = 0
        self._vertex_positions_raw = None
        self._vertex_normals_raw = None
        self._vertex_offset = None          # [V] displacement along the raw vertex normal (None = 0), see vertex_offset
        self._vertex_uv = None
        self._face_indices = None
        self._face_uv_indices = None
        self._edge_indices = None       # numpy [E,5]
        self._edge_indices_dev = None
        self._area_host, self._area_dev = 0.0, None
        self._triangle_info = None
        self._vertex_positions = None
        self._sec_edge_info = None
        self._face_distrb_cache = None

    # total area (mesh.cpp:244-246): Scene.configure leaves it on the device (one element of the native chain's area table); the host value
    # is read when somebody asks
    @property
    def m_total_area(self):
        if self._area_dev is not None:
            self._area_host, self._area_dev = float(self._area_dev.item()), None
        return self._area_host

    @m_total_area.setter
    def m_total_area(self, v):
        self._area_host, self._area_dev = float(v), None

    @property
    def m_inv_total_area(self):
        a = self.m_total_area
        return 1.0 / a if a != 0.0 else 0.0

    @m_inv_total_area.setter
    def m_inv_total_area(self, v):          # derived from m_total_area
        pass

    # -- loading ---------------------------------------------------------------
    def load(self, filename, verbose=False):
        if not os.path.exists(filename):
            raise RuntimeError("Failed to load OBJ from: " + filename)
        v, uv, f, uvf = load_obj(filename)
        self.set_geometry(v, f, uv, uvf, fname=filename)
        if verbose:
            print("Loaded %d vertices, %d faces, %d edges. " % (self.num_vertices, self.num_faces,
                                                              0 if self._edge_indices is None else len(self._edge_indices)))

    def set_geometry(self, verts, faces, uv=None, uv_faces=None, fname="<mesh>"):
        d = _dev()
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        self.num_vertices, self.num_faces = verts.shape[0], faces.shape[0]
        self._vertex_positions_raw = torch.as_tensor(verts, device=d)
        self._face_indices = torch.as_tensor(faces, device=d)
        self.m_has_uv = uv is not None and len(uv) > 0
        if self.m_has_uv:
            self._vertex_uv = torch.as_tensor(np.asarray(uv, dtype=np.float32), device=d)
            self._face_uv_indices = torch.as_tensor(np.asarray(uv_faces, dtype=np.int32), device=d)
        self._edge_indices = build_edge_indices(faces, fname) if self.enable_edges else None
        self._edge_indices_dev = None if self._edge_indices is None else torch.as_tensor(self._edge_indices, device=d)
        self._topo_version = getattr(self, "_topo_version", 0) + 1      # the scene's batched topology is rebuilt
        self.m_ready = False

    # -- python surface (src/psdr.cpp:242-265) ----------------------------------
    def set_transform(self, mat, set_left=True):
        if set_left:
            self._to_world_left = _mat(mat)
        else:
            self._to_world_right = _mat(mat)
        self.m_ready = False

    def append_transform(self, mat, append_left=True):
        if append_left:
            self._to_world_left = _mat(mat) @ self._to_world_left
        else:
            self._to_world_right = self._to_world_right @ _mat(mat)
        self.m_ready = False

    @property
    def to_world(self):
        return Matrix4fD._wrap(self._to_world_raw)

    @property
    def vertex_positions(self):
        return Vector3fD._wrap(self._vertex_positions_raw)

    @vertex_positions.setter
    def vertex_positions(self, v):
        t = v.t if isinstance(v, ek.ArrayBase) else torch.as_tensor(np.asarray(v, dtype=np.float32), device=_dev())
        psdr_assert(t.shape == (self.num_vertices, 3), "vertex_positions: wrong size")
        self._vertex_positions_raw = t
        self.m_ready = False

    @property
    def vertex_offset(self):
        """Per-vertex displacement along the raw vertex normal: m_vertex_positions = to_world * (raw + offset * normal_raw).
        The reference builds this parameter only when PSDR_MESH_ENABLE_1D_VERTEX_OFFSET is defined (include/psdr/macros.h:12,
        shape/mesh.h:37-39,71-80, mesh.cpp:226-232, psdr.cpp:259); here it is always available and costs nothing while unset."""
        if self._vertex_offset is None:
            return FloatD._wrap(torch.zeros(self.num_vertices, device=_dev()))
        return FloatD._wrap(self._vertex_offset)

    @vertex_offset.setter
    def vertex_offset(self, v):
        t = v.t if isinstance(v, ek.ArrayBase) else torch.as_tensor(np.asarray(v, dtype=np.float32), device=_dev())
        t = t.reshape(-1)
        if t.numel() == 1:
            t = t.expand(self.num_vertices)
        psdr_assert(t.shape[0] == self.num_vertices, "vertex_offset: wrong size")
        self._vertex_offset = t
        self.m_ready = False

    def _raw_positions(self):
        """Object-space positions the transform is applied to (mesh.cpp:226-232)."""
        if self._vertex_offset is None:
            return self._vertex_positions_raw
        _, n_raw = process_mesh(self._vertex_positions_raw, self._face_indices)
        return self._vertex_positions_raw + n_raw * self._vertex_offset.unsqueeze(-1)

    @property
    def vertex_normals(self):
        if self._vertex_normals_raw is None and self._vertex_positions_raw is not None:
            _, self._vertex_normals_raw = process_mesh(self._vertex_positions_raw, self._face_indices)
        return Vector3fD._wrap(self._vertex_normals_raw)

    @property
    def _face_distrb(self):
        """face-area distribution (mesh.cpp:248-249), built on first use after a configure"""
        if self._face_distrb_cache is None and self._triangle_info is not None:
            self._face_distrb_cache = DiscreteDistribution()
            self._face_distrb_cache.init(self._triangle_info[:, 21].detach())
        return self._face_distrb_cache

    @_face_distrb.setter
    def _face_distrb(self, v):
        self._face_distrb_cache = v

    @property
    def vertex_uv(self):
        return None if self._vertex_uv is None else Vector2fD._wrap(self._vertex_uv)

    @vertex_uv.setter
    def vertex_uv(self, v):
        self._vertex_uv = v.t if isinstance(v, ek.ArrayBase) else torch.as_tensor(v, device=_dev())

    @property
    def face_indices(self):
        return self._face_indices

    @property
    def face_uv_indices(self):
        return self._face_uv_indices

    def edge_indices(self):
        return None if self._edge_indices is None else self._edge_indices[:, :4].copy()

    def to_string(self):
        return "Mesh[nv=%d, nf=%d, id=%s]" % (self.num_vertices, self.num_faces, self.id)

    # -- configure, mesh.cpp:215-274 -------------------------------------------
    def configure(self):
        if self.bsdf is not None:
            psdr_assert(not self.bsdf.anisotropic() or not self.use_face_normals)
        if self.enable_edges and self._edge_indices is None:
            self._edge_indices = build_edge_indices(self._face_indices.cpu().numpy())
            self._edge_indices_dev = torch.as_tensor(self._edge_indices, device=_dev())
        self._vertex_normals_raw = None       # lazy (vertex_normals property)
        to_world = self._to_world_left @ self._to_world_raw @ self._to_world_right
        self._vertex_positions = transform_pos(to_world, self._raw_positions())
        self._triangle_info, _ = process_mesh(self._vertex_positions, self._face_indices)
        face_areas = self._triangle_info[:, 21]
        self.m_total_area = float(face_areas.detach().sum().item())
        self.m_inv_total_area = 1.0 / self.m_total_area
        self._triangle_uv = None
        if self.m_has_uv:
            fu = self._face_uv_indices.long()
            self._triangle_uv = torch.cat([self._vertex_uv[fu[:, 0]], self._vertex_uv[fu[:, 1]],
                                           self._vertex_uv[fu[:, 2]]], dim=-1)
        self._face_distrb = None              # lazy (property above)
        self._sec_edge_info = None
        if self.enable_edges and self._edge_indices is not None and self._edge_indices.shape[0] > 0:
            ei = self._edge_indices_dev
            is_b = ei[:, 3] < 0
            vp, ti = self._vertex_positions, self._triangle_info
            fnr = ti[:, 18:21]
            p0 = vp.index_select(0, ei[:, 0].long())
            e1 = vp.index_select(0, ei[:, 1].long()) - p0
            n0 = fnr.index_select(0, ei[:, 2].long())
            f1 = torch.where(is_b, torch.zeros_like(ei[:, 3]), ei[:, 3]).long()
            n1 = fnr.index_select(0, f1) * (~is_b).unsqueeze(-1).to(torch.float32)
            p2 = vp.index_select(0, ei[:, 4].long())
            keep = ((n0 * n1).sum(-1) < 1.0 - EdgeEpsilon).detach()
            info = torch.cat([p0, e1, n0, n1, p2, is_b.to(torch.float32).unsqueeze(-1)], dim=-1)
            self._sec_edge_info = info[keep]
        self.m_ready = True

    def sample_position(self, sample2, active=True):
        """Mesh::__sample_position (mesh.cpp:306-330), host/torch mirror for API parity: face by the
        area pmf with sample reuse, uniform point on the triangle.  Returns a PositionSample."""
        psdr_assert(self.m_ready and self._triangle_info is not None)
        ad = isinstance(sample2, ek.ArrayBase) and sample2._ad
        s = (sample2.t if isinstance(sample2, ek.ArrayBase) else torch.as_tensor(sample2, device=_dev())).to(torch.float32)
        d = self._face_distrb
        u = s[:, 0].detach().clone()
        if d.m_size > 1:
            x = u * d.m_sum
            idx = torch.searchsorted(d.m_cmf, x.contiguous(), right=False).clamp(max=d.m_size - 1)
            prev = torch.where(idx > 0, d.m_cmf[(idx - 1).clamp(min=0)], torch.zeros_like(x))
            p = d.m_pmf[idx]
            u = torch.where(p > 0, (x - prev) / p, x - prev).clamp(0.0, 1.0)
        else:
            idx = torch.zeros_like(u, dtype=torch.long)
        t = torch.sqrt(torch.clamp(1.0 - u, min=0.0))
        a, b = 1.0 - t, t * s[:, 1].detach()
        ti = self._triangle_info if ad else self._triangle_info.detach()
        row = ti[idx]
        ps = PositionSampleD() if ad else PositionSampleC()
        ps.p = (Vector3fD if ad else Vector3fC)._wrap(row[:, 0:3] + a.unsqueeze(-1) * row[:, 3:6] + b.unsqueeze(-1) * row[:, 6:9])
        ps.n = (Vector3fD if ad else Vector3fC)._wrap(row[:, 18:21])
        ps.J = (FloatD if ad else FloatC)._wrap(row[:, 21] / row[:, 21].detach() if ad else torch.ones_like(u))
        ps.pdf = FloatC._wrap(torch.full_like(u, self.m_inv_total_area))
        ps.is_valid = torch.ones_like(u, dtype=torch.bool)
        return ps

    def dump(self, fname):
        """OBJ writer, reference mesh.cpp:354-418 (raw vertices, 1-based faces)."""
        v = self._vertex_positions_raw.detach().cpu().numpy()
        f = self._face_indices.cpu().numpy()
        with open(fname, "w") as o:
            for p in v:
                o.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
            if self.m_has_uv:
                for t in self._vertex_uv.detach().cpu().numpy():
                    o.write("vt %.9g %.9g\n" % (t[0], t[1]))
                fu = self._face_uv_indices.cpu().numpy()
                for a, b in zip(f, fu):
                    o.write("f %d/%d %d/%d %d/%d\n" % (a[0] + 1, b[0] + 1, a[1] + 1, b[1] + 1, a[2] + 1, b[2] + 1))
            else:
                for a in f:
                    o.write("f %d %d %d\n" % (a[0] + 1, a[1] + 1, a[2] + 1))


# ------------------------------------------------------------------------------ scene
def _parse_vector(s, length, allow_empty=False):
    vals = [float(x) for x in s.replace(",", " ").split()]
    psdr_assert(len(vals) <= length)
    if len(vals) < length:
        if not allow_empty:
            raise RuntimeError("Vector too short: [%s]" % s)
        vals = vals + [vals[-1] if vals else 0.0] * (length - len(vals))
    return vals


def _rotate_deg(axis, angle_deg):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)        # enoki::rotate normalises nothing; XML axes are unit vectors
    ang = math.radians(angle_deg)
    s, c = math.sin(ang), math.cos(ang)
    x, y, z = a
    cm = 1 - c
    return np.array([[c + x * x * cm, x * y * cm - z * s, x * z * cm + y * s, 0],
                     [y * x * cm + z * s, c + y * y * cm, y * z * cm - x * s, 0],
                     [z * x * cm - y * s, z * y * cm + x * s, c + z * z * cm, 0],
                     [0, 0, 0, 1.0]])


def _load_transform(node):
    """reference src/scene/scene_loader.cpp:80-127: result = T_child * result in document order."""
    result = np.eye(4)
    if node is None:
        return result
    name = node.get("name", "")
    psdr_assert(name in ("to_world", "toWorld"), "Invalid transformation name: " + name)
    for c in node:
        if c.tag == "translate":
            m = np.eye(4); m[:3, 3] = [float(c.get("x", 0)), float(c.get("y", 0)), float(c.get("z", 0))]
        elif c.tag == "rotate":
            m = _rotate_deg([float(c.get("x", 0)), float(c.get("y", 0)), float(c.get("z", 0))], float(c.get("angle", 0)))
        elif c.tag == "scale":
            m = np.diag([float(c.get("x", 1)), float(c.get("y", 1)), float(c.get("z", 1)), 1.0])
        elif c.tag in ("look_at", "lookAt", "lookat"):
            m = look_at(_parse_vector(c.get("origin"), 3), _parse_vector(c.get("target"), 3), _parse_vector(c.get("up"), 3))
        elif c.tag == "matrix":
            m = np.asarray(_parse_vector(c.get("value"), 16)).reshape(4, 4)
        else:
            raise RuntimeError("Unsupported transformation: " + c.tag)
        result = m @ result
    return result


def _find_child(node, names, allow_empty=False):
    for c in node:
        if c.get("name") in names:
            return c
    if not allow_empty:
        raise RuntimeError("Missing child node: " + sorted(names)[0])
    return None


def _load_rgb(node):
    if node.tag == "float":
        return [float(node.get("value"))] * 3
    if node.tag == "rgb":
        return _parse_vector(node.get("value"), 3, True)
    raise RuntimeError("Unsupported RGB type: " + node.tag)


def _load_texture(node, bitmap, base_dir):
    if node.tag == "texture":
        psdr_assert(node.get("type") == "bitmap", "Unsupported texture type: %s" % node.get("type"))
        fn = node.find("string")
        psdr_assert(fn is not None and fn.get("name") == "filename", "Failed to retrieve bitmap filename")
        bitmap.load_openexr(_resolve(fn.get("value"), base_dir))
    elif bitmap.channels == 1:
        bitmap.fill(float(node.get("value")))
    else:
        bitmap.fill(_load_rgb(node))


def _resolve(path, base_dir):
    """Scene files name assets relative to the process cwd (reference behaviour); fall back to
    the XML's directory and its parents so fixtures load from anywhere."""
    if os.path.isabs(path) or os.path.exists(path):
        return path
    d = base_dir
    for _ in range(4):
        if d is None:
            break
        cand = os.path.normpath(os.path.join(d, path))
        if os.path.exists(cand):
            return cand
        d = os.path.dirname(d)
    cand = os.path.normpath(os.path.join(_abi.PKG_ROOT, path))      # bundled fixtures (./data/...)
    return cand if os.path.exists(cand) else path


class Scene(Object):
    """reference include/psdr/scene/scene.h, src/scene/scene.cpp"""
    _type_name = "Scene"

    def __init__(self):
        super().__init__()
        self.opts = RenderOption(0, 0, 0, 0, 0)
        self.m_loaded = False
        self.m_sensors, self.m_emitters, self.m_bsdfs, self.m_meshes = [], [], [], []
        self.m_emitter_env = None
        self.m_has_bound_mesh = False
        self.param_map = {}
        self.num_sensors = 0
        self.num_meshes = 0
        self._tables = None
        self._sensor_tables = []
        self._native = None            # psdr_scene_t
        self.native_options = {}       # developer options applied to the handle when it is created (psdr_scene_set_option: name -> value)
        self._sample_count = [0, 0, 0]
        self._rng_offset = [0, 0, 0]
        self._configured = False
        self._version = 0
        self._topo = None
        self._batch = None

    def __del__(self):
        try:
            if self._native is not None:
                _abi.load_hip().psdr_scene_destroy(self._native)
        except Exception:
            pass

    # -- loading (src/scene/scene_loader.cpp) -----------------------------------
    def load_file(self, file_name, auto_configure=True):
        try:
            root = ET.parse(file_name).getroot()
        except (ET.ParseError, OSError):
            raise RuntimeError("XML parsing failed")
        self._load_scene(root, os.path.dirname(os.path.abspath(file_name)))
        if auto_configure:
            self.configure()

    def load_string(self, scene_xml, auto_configure=True):
        try:
            root = ET.fromstring(scene_xml)
        except ET.ParseError:
            raise RuntimeError("XML parsing failed")
        self._load_scene(root, None)
        if auto_configure:
            self.configure()

    def _load_scene(self, root, base_dir):
        psdr_assert(not self.m_loaded, "Scene already loaded!")
        if root.tag != "scene":
            root = root.find("scene")
        for node in root.findall("sensor"):
            self._load_sensor(node)
        for node in root.findall("bsdf"):
            self._load_bsdf(node, base_dir)
        for node in root.findall("emitter"):
            self._load_emitter(node, base_dir)
        for node in root.findall("shape"):
            self._load_shape(node, base_dir)
        self._build_param_map()
        self.num_sensors, self.num_meshes = len(self.m_sensors), len(self.m_meshes)
        self.m_loaded = True

    def _build_param_map(self):
        for name, arr in (("Mesh", self.m_meshes), ("Emitter", self.m_emitters), ("Sensor", self.m_sensors)):
            for i, obj in enumerate(arr):
                self.param_map["%s[%d]" % (name, i)] = obj
                if obj.id:
                    key = "%s[id=%s]" % (name, obj.id)
                    psdr_assert(key not in self.param_map, "Duplicate id: " + obj.id)
                    self.param_map[key] = obj

    def _load_sensor(self, node):
        film, sampler = node.find("film"), node.find("sampler")
        if not self.m_sensors:
            psdr_assert(film is not None, "Missing film node")
            psdr_assert(sampler is not None, "Missing sampler node")
            self.opts.width = int(_find_child(film, {"width"}).get("value"))
            self.opts.height = int(_find_child(film, {"height"}).get("value"))
            spp = int(sampler.find("integer").get("value"))
            self.opts.spp = self.opts.sppe = self.opts.sppse = spp
        else:
            psdr_assert(film is None, "Duplicate film node")
            psdr_assert(sampler is None, "Duplicate sampler node")
        if node.get("type") != "perspective":
            raise RuntimeError("Unsupported sensor: " + str(node.get("type")))
        to_world = _load_transform(node.find("transform"))
        fov_x = float(_find_child(node, {"fov"}).get("value"))
        ax = _find_child(node, {"fov_axis", "fovAxis"}, True)
        if ax is not None and ax.get("value") != "x":
            raise RuntimeError("Unsupported fov-axis: " + ax.get("value"))
        nn = _find_child(node, {"near_clip", "nearClip"}, True)
        ff = _find_child(node, {"far_clip", "farClip"}, True)
        s = PerspectiveCamera(fov_x, float(nn.get("value", 0.1)) if nn is not None else 0.1,
                              float(ff.get("value", 1e4)) if ff is not None else 1e4)
        s.to_world = to_world
        self.m_sensors.append(s)

    def _load_emitter(self, node, base_dir):
        """SceneLoader::load_emitter, scene_loader.cpp:291-315 (top-level emitters: envmap only)."""
        if node.get("type") != "envmap":
            raise RuntimeError("Unsupported emitter: " + str(node.get("type")))
        psdr_assert(self.m_emitter_env is None, "A scene is only allowed to have one envmap!")
        fn = node.find("string")
        psdr_assert(fn is not None and fn.get("name") == "filename" and fn.get("value"), "Failed to retrieve bitmap filename")
        sc = _find_child(node, {"scale"}, True)
        e = EnvironmentMap(_resolve(fn.get("value"), base_dir))
        e.scale = FloatD(float(sc.get("value", 1.0)) if sc is not None else 1.0)
        e._to_world_raw = torch.as_tensor(_load_transform(node.find("transform")), dtype=torch.float32, device=_dev())
        self.m_emitters.append(e)
        self.m_emitter_env = e

    def add_environment_map(self, env):
        """programmatic construction (fixtures / tests)"""
        psdr_assert(self.m_emitter_env is None, "A scene is only allowed to have one envmap!")
        self.m_emitters.append(env)
        self.m_emitter_env = env

    def _load_bsdf(self, node, base_dir):
        bid = node.get("id")
        psdr_assert(bid, "BSDF must have an id")
        t = node.get("type")
        if t == "diffuse":
            b = Diffuse()
            _load_texture(_find_child(node, {"reflectance"}), b.reflectance, base_dir)
        elif t == "roughconductor":
            b = RoughConductor()
            alpha, eta, k = (_find_child(node, {n}) for n in ("alpha", "eta", "k"))
            _load_texture(alpha, b.alpha_u, base_dir); _load_texture(alpha, b.alpha_v, base_dir)
            _load_texture(eta, b.eta, base_dir); _load_texture(k, b.k, base_dir)
        else:
            raise RuntimeError("Unsupported BSDF: " + str(t))
        b.id = bid
        self.add_bsdf(b)

    def add_bsdf(self, b):
        self.m_bsdfs.append(b)
        self.param_map["BSDF[%d]" % (len(self.m_bsdfs) - 1)] = b
        key = "BSDF[id=%s]" % b.id
        psdr_assert(key not in self.param_map, "Duplicate BSDF id: " + b.id)
        self.param_map[key] = b

    def _load_shape(self, node, base_dir):
        if node.get("type") != "obj":
            raise RuntimeError("Unsupported shape: " + str(node.get("type")))
        name = node.find("string")
        psdr_assert(name is not None and name.get("name") == "filename")
        mesh = Mesh()
        mesh.load(_resolve(name.get("value"), base_dir))
        ref = node.find("ref")
        psdr_assert(ref is not None, "Missing BSDF reference")
        key = "BSDF[id=%s]" % ref.get("id")
        psdr_assert(key in self.param_map, "Unknown BSDF id: " + str(ref.get("id")))
        mesh.bsdf = self.param_map[key]
        psdr_assert(node.find("bsdf") is None, "BSDFs declared under shapes are not supported.")
        fn = _find_child(node, {"face_normals", "faceNormals"}, True)
        mesh.use_face_normals = fn is not None and fn.get("value") == "true"
        if node.get("id"):
            mesh.id = node.get("id")
        em = node.find("emitter")
        if em is not None:
            psdr_assert(em.get("type") == "area", "Unsupported emitter: " + str(em.get("type")))
            e = AreaLight(_load_rgb(_find_child(em, {"radiance"})), mesh)
            self.m_emitters.append(e)
            mesh.m_emitter = e
        mesh._to_world_raw = torch.as_tensor(_load_transform(node.find("transform")), dtype=torch.float32, device=_dev())
        self.m_meshes.append(mesh)

    # programmatic construction (used by fixtures / tests)
    def add_mesh(self, mesh, bsdf, emitter_radiance=None):
        mesh.bsdf = bsdf
        if emitter_radiance is not None:
            e = AreaLight(emitter_radiance, mesh)
            self.m_emitters.append(e)
            mesh.m_emitter = e
        self.m_meshes.append(mesh)

    def add_sensor(self, sensor):
        self.m_sensors.append(sensor)

    def finalize(self):
        self._build_param_map()
        self.num_sensors, self.num_meshes = len(self.m_sensors), len(self.m_meshes)
        self.m_loaded = True

    # -- batched mesh pipeline ----------------------------------------------------
    # The reference configures mesh by mesh (mesh.cpp:215-274) inside one Enoki trace; as eager torch ops
    # that is ~90 tiny kernels per mesh.  Here the static topology of ALL meshes is concatenated once
    # (global vertex / face / edge ids) and every configure() runs ONE transform, ONE process_mesh and ONE
    # edge pass over the whole scene; the per-mesh views (`mesh._triangle_info`, ...) are slices of it.
    def _batch_topology(self):
        key = tuple((id(m), getattr(m, "_topo_version", 0), m.num_vertices, m.num_faces, m.enable_edges, m.use_face_normals,
                     None if m._edge_indices is None else m._edge_indices.shape[0]) for m in self.m_meshes)
        if self._topo is not None and self._topo["key"] == key:
            return self._topo
        d = _dev()
        v_off, f_off, faces, vmesh, tmesh, edges, eface = [0], [0], [], [], [], [], []
        for i, m in enumerate(self.m_meshes):
            if m.enable_edges and m._edge_indices is None:
                m._edge_indices = build_edge_indices(m._face_indices.cpu().numpy())
                m._edge_indices_dev = torch.as_tensor(m._edge_indices, device=d)
            faces.append(m._face_indices.long() + v_off[-1])
            vmesh.append(torch.full((m.num_vertices,), i, dtype=torch.int64, device=d))
            tmesh.append(torch.full((m.num_faces,), i, dtype=torch.int64, device=d))
            if m.enable_edges and m._edge_indices is not None and m._edge_indices.shape[0] > 0:
                e = m._edge_indices_dev.long()
                g = torch.stack([e[:, 0] + v_off[-1], e[:, 1] + v_off[-1], e[:, 2] + f_off[-1],
                                 torch.where(e[:, 3] >= 0, e[:, 3] + f_off[-1], e[:, 3]), e[:, 4] + v_off[-1]], dim=-1)
                edges.append(g)
                eface.append(torch.full((g.shape[0],), bool(m.use_face_normals), dtype=torch.bool, device=d))
            v_off.append(v_off[-1] + m.num_vertices)
            f_off.append(f_off[-1] + m.num_faces)
        flags = [(i | (_abi.TRI_FACE_NORMALS if m.use_face_normals else 0)) for i, m in enumerate(self.m_meshes)]
        self._topo = {
            "key": key, "v_off": v_off, "f_off": f_off, "faces": torch.cat(faces), "vmesh": torch.cat(vmesh),
            "tmesh": torch.cat(tmesh),
            "tri_mesh": torch.cat([torch.full((m.num_faces,), fl, dtype=torch.int32, device=d) for m, fl in zip(self.m_meshes, flags)]).contiguous(),
            "edges": torch.cat(edges) if edges else None, "edge_face_normals": torch.cat(eface) if eface else None,
        }
        # what the native table chain reads (tables_native / csrc/psdr_tables.hip): int32 ids, uint8 flags
        self._topo["faces_i32"] = self._topo["faces"].to(torch.int32).contiguous()
        self._topo["vmesh_i32"] = self._topo["vmesh"].to(torch.int32).contiguous()
        self._topo["edges_i32"] = self._topo["edges"].to(torch.int32).contiguous() if edges else None
        self._topo["edge_face_normals_u8"] = self._topo["edge_face_normals"].to(torch.uint8).contiguous() if eface else None
        return self._topo

    def _configure_meshes(self):
        """world positions, triangle table, areas and per-mesh views for all meshes at once"""
        tp = self._batch_topology()
        meshes = self.m_meshes
        for m in meshes:
            if m.bsdf is not None:
                psdr_assert(not m.bsdf.anisotropic() or not m.use_face_normals)
        # the per-mesh transforms: the product is kept while none of its factors changed or carries a gradient (an optimisation of vertex
        # positions leaves them alone: two batched 4x4 products and three stacks less per configure)
        parts = [t for m in meshes for t in (m._to_world_left, m._to_world_raw, m._to_world_right)]
        mats_key = None if any(t.requires_grad for t in parts) else tuple((id(t), t.data_ptr(), t._version) for t in parts)
        if mats_key is not None and getattr(self, "_mats_key", None) == mats_key:
            mats = self._mats
        else:
            mats = torch.bmm(torch.bmm(torch.stack([m._to_world_left for m in meshes]), torch.stack([m._to_world_raw for m in meshes])),
                             torch.stack([m._to_world_right for m in meshes]))
            self._mats_key, self._mats = mats_key, (mats if mats_key is not None else None)
            self._mats_alive = parts if mats_key is not None else None        # keyed tensors stay alive (ids / blocks stay unique), as _static_key does
        v_raw = torch.cat([m._raw_positions() for m in meshes], dim=0)
        def to_world(v, mm):
            mv = mm.index_select(0, tp["vmesh"].long())                                                   # [V,4,4]
            h = (mv[:, :3, :3] * v.unsqueeze(1)).sum(-1) + mv[:, :3, 3]
            w = (mv[:, 3, :3] * v).sum(-1) + mv[:, 3, 3]
            return h / w.unsqueeze(-1)                                             # transform_pos, transform.h:84-88
        if tables_native.available(v_raw) and not mats.requires_grad:              # one launch (and one in the backward) instead of nine (fifteen)
            v_world = tables_native.world_vertices(v_raw, tp["vmesh_i32"], mats, to_world)
        else:
            v_world = to_world(v_raw, mats)
        if tables_native.available(v_world):          # one forward (and one reverse) launch sequence on the HIP library
            # rows of PSDR_TRI_STRIDE words (padding zero): psdr_scene_desc::tri_info as it stands; the areas are summed with the emitter
            # tables (_finish_native), nothing is read back
            tri_info = tables_native.tri_rows(v_world, tp["faces_i32"], lambda v, f: process_mesh(v, f)[0], width=_abi.TRI_STRIDE)
            self._areas_t = None
        else:
            tri_info, _ = process_mesh(v_world, tp["faces"])
            # total areas: on the device until _mesh_areas_to_host (configure() reads every size and sum it needs back in ONE batch)
            self._areas_t = torch.zeros(len(meshes), device=v_world.device).index_add(0, tp["tmesh"], tri_info[:, 21].detach())
        for i, m in enumerate(meshes):
            m._vertex_positions = v_world[tp["v_off"][i]:tp["v_off"][i + 1]]
            m._triangle_info = tri_info[tp["f_off"][i]:tp["f_off"][i + 1]]
            m._vertex_normals_raw = None
            m._face_distrb = None
            m._sec_edge_info = None
            if m.m_has_uv:
                uv_key = (id(m._face_uv_indices), m._face_uv_indices._version, id(m._vertex_uv), m._vertex_uv._version)
                if getattr(m, "_triangle_uv_key", None) != uv_key or m._triangle_uv is None:        # the uv tables do not move with the vertices
                    fu = m._face_uv_indices.long()
                    m._triangle_uv = torch.cat([m._vertex_uv[fu[:, 0]], m._vertex_uv[fu[:, 1]], m._vertex_uv[fu[:, 2]]], dim=-1)
                    m._triangle_uv_key = uv_key
            else:
                m._triangle_uv = None
            m.m_ready = True
        return tp, v_world, tri_info

    def _mesh_areas_to_host(self, areas):
        for m, a in zip(self.m_meshes, areas):
            m.m_total_area = float(np.float32(a))
            m.m_inv_total_area = 1.0 / m.m_total_area

    def _secondary_edges(self, tp, v_world, tri_info):
        """SecondaryEdgeInfo of every mesh with edges (mesh.cpp:251-270) and the coplanar filter's mask, one pass"""
        ei = tp["edges"]
        if ei is None:
            return None

        def records(v, rows, edges):
            edges = edges.long()
            is_b = edges[:, 3] < 0
            fnr = rows[:, 18:21]
            p0 = v.index_select(0, edges[:, 0])
            e1 = v.index_select(0, edges[:, 1]) - p0
            n0 = fnr.index_select(0, edges[:, 2])
            n1 = fnr.index_select(0, torch.where(is_b, torch.zeros_like(edges[:, 3]), edges[:, 3])) * (~is_b).unsqueeze(-1).to(torch.float32)
            p2 = v.index_select(0, edges[:, 4])
            return torch.cat([p0, e1, n0, n1, p2, is_b.to(torch.float32).unsqueeze(-1)], dim=-1)
        if tables_native.available(v_world):
            info, keep8 = tables_native.sec_edges(v_world, tri_info, tp["edges_i32"], records)
            return info, keep8          # _finish_native compacts on the device
        else:
            info = records(v_world, tri_info, ei)
            is_b = ei[:, 3] < 0           # (the filter on the gathered normals themselves, as this chain always evaluated it: the rounding of the
            n0 = tri_info[ei[:, 2], 18:21]     # reduction depends on the memory layout of its operand, and the committed fixtures follow these decisions)
            n1 = tri_info[torch.where(is_b, torch.zeros_like(ei[:, 3]), ei[:, 3]), 18:21] * (~is_b).unsqueeze(-1).to(torch.float32)
            keep = ((n0 * n1).sum(-1) < 1.0 - EdgeEpsilon).detach()
        return info, keep          # configure() compacts once it holds the count (no read-back here)

    def _upload_small(self, arrays, d):
        """Host float32 arrays -> device tensors through ONE staging buffer and one copy (pinned + asynchronous on a GPU: a pageable
        host-to-device copy would wait for everything queued on the stream)."""
        sizes = [int(a.size) for a in arrays]
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in arrays]) if arrays else np.zeros(0, np.float32)
        if d.type == "cpu":
            dev = torch.from_numpy(flat.copy())
        else:
            pin = getattr(self, "_pin", None)
            # two staging buffers used alternately: the copy out of one is complete before it is written again (every configure() reads
            # results back after queueing its copy)
            if pin is None or pin[0].numel() < flat.size:
                pin = [torch.empty(max(256, 2 * flat.size), dtype=torch.float32).pin_memory() for _ in range(2)]
                self._pin, self._pin_turn = pin, 0
            self._pin_turn ^= 1
            buf = pin[self._pin_turn]
            buf[:flat.size].copy_(torch.from_numpy(flat))
            dev = buf[:flat.size].to(d, non_blocking=True)
        out, o = [], 0
        for n in sizes:
            out.append(dev[o:o + n]); o += n
        return out

    def _material_tables(self, d):
        """BSDF records + texel pool (+ the environment map's texels behind them): the part of the tables that a
        material parameter changes."""
        bsdf_ids = {id(b): i for i, b in enumerate(self.m_bsdfs)}
        pool, rec, off = [], [], 0

        def put(bm):
            nonlocal off
            t = bm.tensor()
            w, h = bm.resolution
            pool.append(t.reshape(-1).to(d))
            o_ = off
            off += t.numel()
            return [o_, w, h]
        for b in self.m_bsdfs:
            if isinstance(b, Diffuse):
                r = [_abi.BSDF_DIFFUSE] + put(b.reflectance) + [0, 1, 1] * 4
            elif isinstance(b, RoughConductor):
                r = ([_abi.BSDF_ROUGHCONDUCTOR] + put(b.specular_reflectance) + put(b.alpha_u) + put(b.alpha_v) +
                     put(b.eta) + put(b.k))
            else:
                raise RuntimeError("Unsupported BSDF: " + b.type_name())
            rec.append(r)
        out = {}
        key = tuple(map(tuple, rec))
        if getattr(self, "_bsdf_rec_key", None) != key:            # the int table only changes with the texture layout
            self._bsdf_rec_key = key
            self._bsdf_rec_t = torch.tensor(rec if rec else [[0] * 16], dtype=torch.int32, device=d).contiguous()
        out["bsdf_rec"] = self._bsdf_rec_t
        out["material_mask"] = sum({1 << r[0] for r in rec}) if rec else 0      # BSDF types present (psdr_scene_desc.material_mask)
        out["env_tex"] = put(self.m_emitter_env.radiance) if self.m_emitter_env is not None else [0, 0, 0]
        out["texels"] = (torch.cat(pool) if pool else torch.zeros(1, device=d)).to(torch.float32).contiguous()
        mb = tuple(bsdf_ids.get(id(m.bsdf), -1) for m in self.m_meshes)
        if getattr(self, "_mesh_bsdf_key", None) != mb:
            self._mesh_bsdf_key = mb
            self._mesh_bsdf_t = torch.tensor(mb, dtype=torch.int32, device=d)
        out["mesh_bsdf"] = self._mesh_bsdf_t
        return out

    def _static_key(self):
        """Identity + version of everything configure() reads EXCEPT the BSDF parameters, or None when some of it
        carries a gradient (its torch graph must then be rebuilt every time).  While the key is unchanged the
        geometry / camera / emitter / edge tables of the last configure() are still right: a material-only
        optimisation loop (examples/run_test.py material_roughness, the headline albedo benchmark) then pays for the
        texel pool only -- 1.4 ms -> 0.3 ms per configure() on the Cornell box."""
        if self.m_emitter_env is not None:          # the environment map owns derived state (bounding mesh, cell masses): always rebuilt
            return None, None
        ts = []
        for m in self.m_meshes:
            ts += [m._vertex_positions_raw, m._to_world_raw, m._to_world_left, m._to_world_right, m._vertex_offset]
        for sn in self.m_sensors:
            ts.append(sn._to_world)
        for e in self.m_emitters:
            ts.append(e.radiance.t)
        if any(t is not None and t.requires_grad for t in ts):
            return None, None
        o = self.opts
        scalars = (o.width, o.height, o.sppe > 0, o.sppse > 0, bool(getattr(o, "primary_edge_vis_check", False)), len(self.m_meshes), len(self.m_sensors), len(self.m_emitters),
                   tuple((id(m), m.enable_edges, m.use_face_normals, m.m_has_uv, id(m.m_emitter), getattr(m, "_topo_version", 0), m._vertex_offset is None) for m in self.m_meshes),
                   tuple((type(sn).__name__, getattr(sn, "m_fov_x", None), getattr(sn, "m_near_clip", None), getattr(sn, "m_far_clip", None)) for sn in self.m_sensors),
                   tuple((type(e).__name__, id(e.m_mesh)) for e in self.m_emitters),
                   tuple(self._sample_count))
        return (scalars, tuple((id(t), t._version) for t in ts if t is not None)), ts      # ts keeps the tensors alive: ids stay unique

    # -- configure (src/scene/scene.cpp:56-278) ---------------------------------
    def configure(self):
        psdr_assert(self.m_loaded, "Scene not loaded yet!")
        t_start = time.perf_counter()
        o = self.opts
        d = _dev()
        # samplers: re-seed (offset 0) only when the slot count changed, scene.cpp:65-79
        for k, n in enumerate((o.spp, o.sppe, o.sppse)):
            if n > 0:
                count = o.height * o.width * n
                if self._sample_count[k] != count:
                    self._sample_count[k] = count
                    self._rng_offset[k] = 0
        psdr_assert(self.m_meshes, "Missing meshes!")
        psdr_assert(self.m_sensors, "Missing sensor!")
        key, alive = self._static_key()
        if key is not None and self._configured and getattr(self, "_static_cache", None) is not None and self._static_cache[0] == key:
            # nothing but BSDF parameters changed since the last configure(): its geometry-side tables stand
            tb = dict(self._static_cache[2])
            mt = self._material_tables(d)
            if self.m_emitter_env is not None:
                tb["env_tex"] = mt["env_tex"]
            mt.pop("env_tex")
            tb.update(mt)
            self._version += 1
            tb["version"] = self._version
            self._tables = tb           # _bvh_version stays: the tables keep their geo_version stamp, the tree on the handle is still theirs
            if o.log_level > 0:
                self.log("Configured in %g seconds (material tables only)." % (time.perf_counter() - t_start))
            return
        has_uv = any(m.m_has_uv for m in self.m_meshes)
        tp, v_world, tri_info22 = self._configure_meshes()
        self._batch = {"tp": tp, "v_world": v_world, "tri_info": tri_info22}       # the sensors' edge pass reads it
        # AABB over the meshes, scene.cpp:88-101 (m_upper starts at numeric_limits<float>::min(), the
        # smallest POSITIVE float: kept as is)
        # sensors (+ camera positions into the AABB, scene.cpp:104-119): everything up to the silhouette masks, no read-back yet
        sensor_states = [s.configure_begin(self) for s in self.m_sensors]
        # the box itself is formed when somebody asks for it (m_lower / m_upper: the environment map's bounding mesh, the log line) -- seven launches
        # per configure() that an optimisation loop never looks at
        self._aabb_pending = (v_world.detach(), [st["out"]["cam"][_abi.CAM_POS:_abi.CAM_POS + 3].detach() for st in sensor_states])

        # environment lighting: bounding mesh added once, scene.cpp:135-180
        if self.m_emitter_env is not None and not self.m_has_bound_mesh:
            margin = ((self.m_upper - self.m_lower) * np.float32(0.05)).min()
            self.m_lower, self.m_upper = self.m_lower - margin, self.m_upper + margin
            env = self.m_emitter_env
            env.m_lower, env.m_upper = self.m_lower.clone(), self.m_upper.clone()
            lo, hi = self.m_lower.cpu().numpy(), self.m_upper.cpu().numpy()
            verts = np.array([[hi[j] if (i >> j) & 1 else lo[j] for j in range(3)] for i in range(8)], dtype=np.float32)
            faces = np.array([[0, 1, 3], [0, 3, 2], [1, 5, 7], [1, 7, 3], [2, 3, 7], [2, 7, 6],
                              [0, 5, 1], [0, 4, 5], [0, 2, 6], [0, 6, 4], [4, 7, 5], [4, 6, 7]], dtype=np.int32)
            bound = Mesh()
            bound.enable_edges = False
            bound.use_face_normals = True
            bound.set_geometry(verts, faces, fname="<envmap bounding mesh>")
            bound.bsdf, bound.m_emitter = None, env
            env.m_mesh = bound
            self.m_meshes.append(bound)
            self.num_meshes = len(self.m_meshes)
            self.m_has_bound_mesh = True
            tp, v_world, tri_info22 = self._configure_meshes()        # once: the mesh list just grew (no edges on the box)
            self._batch = {"tp": tp, "v_world": v_world, "tri_info": tri_info22}
            for st in sensor_states:                                  # the edge states hold the batch they were made from (same edges: the box has none)
                if "bt" in st:
                    st["bt"] = self._batch
            if o.log_level > 0:
                self.log("Bounding mesh added for environmental lighting.")

        face_offset = tp["f_off"]
        T = face_offset[-1]
        # secondary edges, scene.cpp:219-244: records and the coplanar filter's mask (no read-back yet)
        sec = self._secondary_edges(tp, v_world, tri_info22) if o.sppse > 0 else None
        if tables_native.available(v_world):
            return self._finish_native(d, tp, tri_info22, sec, sensor_states, key, alive, has_uv, t_start)

        # ---- ONE read-back for every size and sum the host needs from here on: mesh areas, the luminance of the area lights, the
        # sums of their face distributions, the numbers of kept secondary / primary edges
        area_lights = [e for e in self.m_emitters if not isinstance(e, EnvironmentMap)]
        for e in area_lights:          # the reference's assertion (arealight.cpp configure), BEFORE the batched read-back dereferences the mesh
            psdr_assert(e.m_mesh is not None and e.m_mesh.m_ready)
        parts = [self._areas_t]
        parts += [e._luminance_t() for e in area_lights]
        parts += [e.m_mesh._triangle_info[:, 21].detach().to(torch.float32).sum().reshape(1) for e in area_lights]
        counts = [sec[1].sum().reshape(1)] if sec is not None else []
        counts += [st["count_t"] for st in sensor_states if st["count_t"] is not None]
        # floats and (bit-cast) 32-bit counts in one float32 buffer, one copy
        buf = torch.cat(parts + [c.to(torch.int32).view(torch.float32) for c in counts]).cpu()
        nf = buf.numel() - len(counts)
        stats = buf[:nf].tolist() + [float(x) for x in buf[nf:].view(torch.int32).tolist()]
        if sec is None:
            stats.insert(nf, 0.0)
        k = nf + 1
        for st in sensor_states:                                   # a sensor without an edge pass reads as zero kept edges
            if st["count_t"] is None:
                stats.insert(k, 0.0)
            k += 1
        M, L = len(self.m_meshes), len(area_lights)
        self._mesh_areas_to_host(stats[:M])
        lum_of = {id(e): stats[M + i] for i, e in enumerate(area_lights)}
        face_sum_of = {id(e): stats[M + L + i] for i, e in enumerate(area_lights)}
        n_sec = int(round(stats[M + 2 * L]))
        n_prim = [int(round(x)) for x in stats[M + 2 * L + 1:]]

        # sensors: compacted primary-edge tables (their sums are read back with the secondary edges' at the end)
        self._sensor_tables = [s.configure_finish(st, n) for s, st, n in zip(self.m_sensors, sensor_states, n_prim)]

        tri_info = torch.cat([tri_info22, torch.zeros(T, 2, device=d)], dim=-1).contiguous()
        tb = {"tri_info": tri_info, "tri_mesh": tp["tri_mesh"], "num_tris": T,
              "tri_uv": None, "face_offset": face_offset}
        if has_uv:
            uv_rows = [m._triangle_uv if m._triangle_uv is not None else torch.zeros(m.num_faces, 6, device=d) for m in self.m_meshes]
            tb["tri_uv"] = torch.cat([torch.cat(uv_rows, dim=0).detach(), torch.zeros(T, 2, device=d)], dim=-1).contiguous()

        mt = self._material_tables(d)
        env_tex = mt.pop("env_tex")
        tb.update(mt)

        # emitters, scene.cpp:183-196 + area.cpp:10-16.  The small host-made tables travel in one pinned staging buffer (asynchronous copy:
        # a pageable host-to-device copy waits for the stream)
        em_ids = {id(e): i for i, e in enumerate(self.m_emitters)}
        mesh_emitter = [em_ids.get(id(m.m_emitter), -1) for m in self.m_meshes]
        Ne = len(self.m_emitters)
        ef_h = np.zeros((max(Ne, 1), _abi.EMITTER_F_STRIDE), dtype=np.float32)
        ei_h = np.zeros((max(Ne, 1), _abi.EMITTER_I_STRIDE), dtype=np.int32)
        rad = torch.zeros(max(Ne, 1), 3, device=d)
        cmfs, pmfs, coff = [], [], 0
        ed_pmf = np.zeros(1, dtype=np.float32)
        ed_sum = 0.0
        if Ne:
            weights = []
            for e in self.m_emitters:
                if isinstance(e, EnvironmentMap):
                    e.configure()
                else:
                    e.configure(lum_of[id(e)])
                weights.append(e.m_sampling_weight)
            ed_pmf = np.asarray(weights, dtype=np.float32)
            ed_sum = float(torch.from_numpy(ed_pmf).sum().item())           # host tensor: the float32 sum torch forms
            inv_total = np.float32(1.0) / np.float32(ed_sum)
            rads = []
            for i, e in enumerate(self.m_emitters):
                e.m_sampling_weight = float(np.float32(e.m_sampling_weight) * inv_total)
                mi = self.m_meshes.index(e.m_mesh)
                if isinstance(e, EnvironmentMap):
                    ef_h[i, 3] = e.m_sampling_weight
                    ei_h[i] = [mi, face_offset[mi], e.m_mesh.num_faces, 0]
                    rads.append(torch.zeros(3, device=d))
                    continue
                fd = DiscreteDistribution(); fd.init(e.m_mesh._triangle_info[:, 21].detach(), total=face_sum_of[id(e)])
                e.m_mesh._face_distrb = fd
                ef_h[i, 3], ef_h[i, 4], ef_h[i, 5] = e.m_sampling_weight, e.m_mesh.m_inv_total_area, fd.m_sum
                ei_h[i] = [mi, face_offset[mi], e.m_mesh.num_faces, coff]
                cmfs.append(fd.m_cmf); pmfs.append(fd.m_pmf); coff += fd.m_size
                rads.append(e.radiance.t.reshape(3))
            rad = torch.stack(rads)
        ed_cmf = np.cumsum(ed_pmf, dtype=np.float32)
        small = self._upload_small([np.asarray(mesh_emitter, dtype=np.int32).view(np.float32), ef_h.reshape(-1), ei_h.reshape(-1).view(np.float32),
                                    ed_pmf, ed_cmf], d)
        tb["mesh_emitter"] = small[0].view(torch.int32)
        ef = small[1].reshape(max(Ne, 1), _abi.EMITTER_F_STRIDE)
        ei = small[2].view(torch.int32).reshape(max(Ne, 1), _abi.EMITTER_I_STRIDE)
        if Ne:
            ef = torch.cat([rad.detach(), ef[:, 3:]], dim=-1)
            tb["emitter_cmf"], tb["emitter_pmf"], tb["emitter_sum"] = small[4], small[3], ed_sum
        else:
            z = torch.zeros(1, device=d)
            tb["emitter_cmf"], tb["emitter_pmf"], tb["emitter_sum"] = z, z, 0.0
        tb["emitter_f"], tb["emitter_i"], tb["emitter_rad"] = ef.contiguous(), ei.contiguous(), rad
        tb["face_cmf"] = torch.cat(cmfs).contiguous() if cmfs else torch.zeros(1, device=d)
        tb["face_pmf"] = torch.cat(pmfs).contiguous() if pmfs else torch.zeros(1, device=d)
        tb["num_emitters"] = Ne
        if self.m_emitter_env is not None:
            env = self.m_emitter_env
            cd = env._cell_distrb
            tb.update(env_emitter=self.m_emitters.index(env), env_tex=env_tex, env_reso=list(env._cell_reso),
                      env_f=env.record(), env_cmf=cd.m_cmf, env_pmf=cd.m_pmf, env_sum=cd.m_sum)
        else:
            tb.update(env_emitter=-1, env_tex=[0, 0, 0], env_reso=[0, 0], env_f=None, env_cmf=None, env_pmf=None, env_sum=0.0)

        # secondary edges: the kept records, their length distribution
        sums = [st["sum_t"] for st in sensor_states if st.get("sum_t") is not None]
        sd = None
        if sec is not None and n_sec > 0:
            idx = compact_indices(sec[1], n_sec)
            se = sec[0][idx].contiguous()
            self._sec_edge_faces = tp["edges_i32"][idx][:, 2:4].contiguous()      # adjacent faces (global ids; -1 = none)
            e1 = se[:, 3:6].detach()
            ln = torch.sqrt((e1 * e1).sum(-1))
            sd = DiscreteDistribution(); sd.init(ln, total=0.0)
            sums.append(sd.m_pmf.sum().reshape(1))
            tb.update(sec_edge=se, sec_cmf=sd.m_cmf, sec_pmf=sd.m_pmf, sec_sum=0.0, num_sec_edges=int(se.shape[0]),
                      sec_edge_faces=self._sec_edge_faces)
        else:
            if sec is not None:
                self._sec_edge_faces = tp["edges_i32"][:0, 2:4].contiguous()
            tb.update(sec_edge=None, sec_cmf=None, sec_pmf=None, sec_sum=0.0, num_sec_edges=0, sec_edge_faces=None)
        # ---- second (and last) read-back: the sums of the edge distributions, formed over the compacted tables
        if sums:
            vals = torch.cat(sums).tolist()
            k = 0
            for s_, st in zip(self.m_sensors, sensor_states):
                if st.get("sum_t") is not None:
                    s_.configure_sum(st, vals[k]); k += 1
            if sd is not None:
                sd.m_sum = float(vals[k]); tb["sec_sum"] = sd.m_sum
        self._version += 1
        tb["version"] = self._version
        tb["geo_version"] = self._version      # stamps the geometry: the BVH on the native handle is tied to it (Integrator._prepare)
        self._tables = tb
        self._configured = True
        self._bvh_version = None
        self._static_cache = (key, alive, dict(tb)) if key is not None else None
        if o.log_level > 0:
            self.log("AABB: [lower = %s, upper = %s]" % (self.m_lower.tolist(), self.m_upper.tolist()))
            if o.sppe > 0:
                self.log("(%s) primary edges initialized." % ", ".join(str(s["num_prim_edges"]) for s in self._sensor_tables))
            if o.sppse > 0:
                self.log("%d secondary edges initialized." % tb["num_sec_edges"])
            self.log("Configured in %g seconds." % (time.perf_counter() - t_start))

    def _finish_native(self, d, tp, tri_info, sec, sensor_states, key, alive, has_uv, t_start):
        """The rest of configure() on a GPU: every count and sum stays on the device.  Edge tables keep the capacity of their candidate lists
        (kept rows first, csrc/psdr_tables.hip k_compact_*), the distributions are normalised there (the descriptor's sums are 1), mesh areas
        and the emitter tables come from one kernel pair; host attributes that mirror device values (Mesh.m_total_area,
        AreaLight.m_sampling_weight, DiscreteDistribution.m_sum, the numbers of kept edges) are read on first use."""
        o = self.opts
        face_offset = tp["f_off"]
        T = face_offset[-1]
        self._sensor_tables = [st["out"] for st in sensor_states]
        tb = {"tri_info": tri_info, "tri_mesh": tp["tri_mesh"], "num_tris": T, "tri_uv": None, "face_offset": face_offset, "device_counts": True}
        if has_uv:
            uvk = tuple(id(m._triangle_uv) for m in self.m_meshes)
            if getattr(self, "_tri_uv_key", None) != uvk:               # the uv rows do not move with the vertices
                uv_rows = [m._triangle_uv if m._triangle_uv is not None else torch.zeros(m.num_faces, 6, device=d) for m in self.m_meshes]
                self._tri_uv_key = uvk
                self._tri_uv_t = torch.cat([torch.cat(uv_rows, dim=0).detach(), torch.zeros(T, 2, device=d)], dim=-1).contiguous()
            tb["tri_uv"] = self._tri_uv_t
        mt = self._material_tables(d)
        env_tex = mt.pop("env_tex")
        tb.update(mt)

        # ---- emitters, scene.cpp:183-196 + area.cpp:10-16: static ids once per topology, areas / weights / face distributions by kernel
        Ne, M = len(self.m_emitters), len(self.m_meshes)
        for e in self.m_emitters:          # the reference's assertion (AreaLight::configure) before any table dereferences the emitter's mesh
            psdr_assert(e.m_mesh is not None and e.m_mesh.m_ready)
        ekey = (tp["key"], tuple((id(e), id(e.m_mesh)) for e in self.m_emitters), tuple(id(m.m_emitter) for m in self.m_meshes))
        if getattr(self, "_emit_static", None) is None or self._emit_static[0] != ekey:
            em_ids = {id(e): i for i, e in enumerate(self.m_emitters)}
            mesh_emitter = np.asarray([em_ids.get(id(m.m_emitter), -1) for m in self.m_meshes], dtype=np.int32)
            ei_h = np.zeros((max(Ne, 1), _abi.EMITTER_I_STRIDE), dtype=np.int32)
            coff = 0
            for i, e in enumerate(self.m_emitters):
                mi = self.m_meshes.index(e.m_mesh)
                env = isinstance(e, EnvironmentMap)
                ei_h[i] = [mi, face_offset[mi], e.m_mesh.num_faces, 0 if env else coff]
                coff += 0 if env else e.m_mesh.num_faces
            ints = torch.from_numpy(np.concatenate([mesh_emitter, ei_h.reshape(-1), np.asarray(face_offset, dtype=np.int32)])).to(d)
            self._emit_static = (ekey, ints[:M], ints[M:M + ei_h.size].reshape(-1, _abi.EMITTER_I_STRIDE), ints[M + ei_h.size:], coff,
                                 torch.full((max(Ne, 1),), -1.0, device=d))
        _, mesh_emitter_t, ei_t, foff_t, n_face_words, no_env = self._emit_static
        env_w = no_env
        if self.m_emitter_env is not None:
            self.m_emitter_env.configure()
            env_w = torch.tensor([float(e.m_sampling_weight) if isinstance(e, EnvironmentMap) else -1.0 for e in self.m_emitters], dtype=torch.float32, device=d)
        if Ne == 1 and not isinstance(self.m_emitters[0], EnvironmentMap):
            rad = self.m_emitters[0].radiance.t.reshape(1, 3)
        elif Ne:
            rad = torch.stack([torch.zeros(3, device=d) if isinstance(e, EnvironmentMap) else e.radiance.t.reshape(3) for e in self.m_emitters])
        else:
            rad = torch.zeros(1, 3, device=d)
        area, ef, epmf, ecmf, fpmf, fcmf = tables_native.emitter_tables(tri_info, foff_t, mesh_emitter_t, ei_t if Ne else None, rad.detach().contiguous().float(),
                                                                        env_w, n_face_words)
        for i, m in enumerate(self.m_meshes):
            m._area_dev = area[i]
        for i, e in enumerate(self.m_emitters):
            e.m_ready = True
            e._weight_dev = epmf[i]                        # normalised (scene.cpp:190-193 multiplies every weight by 1 / sum)
        coff = 0
        for i, e in enumerate(self.m_emitters):
            if isinstance(e, EnvironmentMap):
                continue
            n = e.m_mesh.num_faces
            fd = DiscreteDistribution(); fd.init_device(fpmf[coff:coff + n], fcmf[coff:coff + n], ef[i, 5])
            e.m_mesh._face_distrb = fd
            coff += n
        tb["mesh_emitter"] = mesh_emitter_t
        if Ne:
            tb.update(emitter_f=ef, emitter_i=ei_t, emitter_rad=rad, emitter_cmf=ecmf, emitter_pmf=epmf, emitter_sum=1.0)
        else:
            z = torch.zeros(1, device=d)
            tb.update(emitter_f=torch.zeros(1, _abi.EMITTER_F_STRIDE, device=d), emitter_i=ei_t, emitter_rad=rad, emitter_cmf=z, emitter_pmf=z, emitter_sum=0.0)
        tb.update(face_cmf=fcmf, face_pmf=fpmf, num_emitters=Ne)
        if self.m_emitter_env is not None:
            env = self.m_emitter_env
            cd = env._cell_distrb
            tb.update(env_emitter=self.m_emitters.index(env), env_tex=env_tex, env_reso=list(env._cell_reso),
                      env_f=env.record(), env_cmf=cd.m_cmf, env_pmf=cd.m_pmf, env_sum=cd.m_sum)
        else:
            tb.update(env_emitter=-1, env_tex=[0, 0, 0], env_reso=[0, 0], env_f=None, env_cmf=None, env_pmf=None, env_sum=0.0)

        # ---- secondary edges: kept records first, normalised length distribution, adjacent faces alongside
        if sec is not None:
            se, faces, _pos, pmf, cmf, hdr = tables_native.compact_edges(sec[0], sec[1], 3, 3, aux=tp["edges_i32"][:, 2:4], aux_cols=2)
            self._sec_edge_faces = faces
            tb.update(sec_edge=se, sec_cmf=cmf, sec_pmf=pmf, sec_sum=1.0, num_sec_edges=int(se.shape[0]), sec_edge_faces=faces, sec_header=hdr)
        else:
            tb.update(sec_edge=None, sec_cmf=None, sec_pmf=None, sec_sum=0.0, num_sec_edges=0, sec_edge_faces=None)
        self._version += 1
        tb["version"] = self._version
        tb["geo_version"] = self._version
        self._tables = tb
        self._edge_counts = None
        self._configured = True
        self._bvh_version = None
        self._static_cache = (key, alive, dict(tb)) if key is not None else None
        if o.log_level > 0:
            self.log("AABB: [lower = %s, upper = %s]" % (self.m_lower.tolist(), self.m_upper.tolist()))
            n_sec, n_prim = self._kept_edge_counts()
            if o.sppe > 0:
                self.log("(%s) primary edges initialized." % ", ".join(str(n) for n in n_prim))
            if o.sppse > 0:
                self.log("%d secondary edges initialized." % n_sec)
            self.log("Configured in %g seconds." % (time.perf_counter() - t_start))

    def _kept_edge_counts(self):
        """(kept secondary edges, [kept primary edges per sensor]) of a native configure(): ONE read of the device headers, on first use."""
        if getattr(self, "_edge_counts", None) is None:
            hs = [self._tables["sec_header"][:1]] if self._tables.get("sec_header") is not None else []
            hs += [st["prim_header"][:1] for st in self._sensor_tables if st.get("prim_header") is not None]
            vals = torch.cat(hs).view(torch.int32).tolist() if hs else []
            k = 0
            n_sec = 0
            if self._tables.get("sec_header") is not None:
                n_sec = int(vals[0]); k = 1
            n_prim = []
            for st in self._sensor_tables:
                if st.get("prim_header") is not None:
                    n_prim.append(int(vals[k])); k += 1
                else:
                    n_prim.append(0)
            self._edge_counts = (n_sec, n_prim)
        return self._edge_counts

    def _resolve_aabb(self):
        pend = getattr(self, "_aabb_pending", None)
        if pend is not None:
            allv, cams = pend
            lo = allv.min(dim=0)[0]
            hi = torch.clamp(allv.max(dim=0)[0], min=float(np.finfo(np.float32).tiny))
            for cp in cams:
                lo, hi = torch.minimum(lo, cp), torch.maximum(hi, cp)
            self._m_lower, self._m_upper, self._aabb_pending = lo, hi, None

    @property
    def m_lower(self):
        self._resolve_aabb()
        return getattr(self, "_m_lower", None)

    @m_lower.setter
    def m_lower(self, v):
        self._resolve_aabb()
        self._m_lower = v

    @property
    def m_upper(self):
        self._resolve_aabb()
        return getattr(self, "_m_upper", None)

    @m_upper.setter
    def m_upper(self, v):
        self._resolve_aabb()
        self._m_upper = v

    def is_ready(self):
        return self._configured and all(m.m_ready for m in self.m_meshes)

    def sample_boundary_segment_direct(self, sample3, active=True):
        """Scene::sample_boundary_segment_direct (scene.cpp:456-492), host/torch mirror for API parity
        (the kernels run their own copy per sample, csrc/psdr_device.h secondary_edge_sample)."""
        tb = self._tables
        psdr_assert(tb is not None and tb["num_sec_edges"] > 0, "Scene has no secondary edges")
        psdr_assert(len(self.m_emitters) == 1, "host mirror supports a single emitter")
        s = (sample3.t if isinstance(sample3, ek.ArrayBase) else torch.as_tensor(sample3, device=_dev())).to(torch.float32).detach()
        x = s[:, 0] * tb["sec_sum"]
        E = tb["num_sec_edges"]
        idx = torch.searchsorted(tb["sec_cmf"], x.contiguous(), right=False).clamp(max=E - 1)
        prev = torch.where(idx > 0, tb["sec_cmf"][(idx - 1).clamp(min=0)], torch.zeros_like(x))
        pm = tb["sec_pmf"][idx]
        s1 = torch.where(pm > 0, (x - prev) / pm, x - prev).clamp(0.0, 1.0) if E > 1 else s[:, 0]
        pdf0 = (pm / tb["sec_sum"]) if E > 1 else torch.ones_like(x)
        row = tb["sec_edge"][idx]
        r = BoundarySegSampleDirect()
        p0 = row[:, 3:6] * s1.unsqueeze(-1) + row[:, 0:3]
        r.p0 = Vector3fD._wrap(p0)
        e1 = row[:, 3:6].detach()
        e1len = torch.sqrt((e1 * e1).sum(-1))
        r.edge = Vector3fC._wrap(e1 / e1len.unsqueeze(-1))
        r.edge2 = Vector3fC._wrap(row[:, 12:15].detach() - row[:, 0:3].detach())
        pdf0 = pdf0 / e1len
        ps = self.m_emitters[0].m_mesh.sample_position(ek.cuda.Vector2f._wrap(s[:, 1:3]))
        r.p2, r.n = ps.p, ps.n
        e = r.p2.t - p0.detach()
        d2 = (e * e).sum(-1)
        e = e / torch.sqrt(d2).unsqueeze(-1)
        cos_t = -(r.n.t * e).sum(-1)
        d0, d1 = (row[:, 6:9].detach() * e).sum(-1), (row[:, 9:12].detach() * e).sum(-1)
        sg = lambda v: (v > EdgeEpsilon).to(torch.int32) - (v < -EdgeEpsilon).to(torch.int32)
        is_b = row[:, 15].detach() != 0
        valid = (cos_t > Epsilon) & torch.where(is_b, sg(d0) != 0, sg(d0) * sg(d1) < 0)
        r.is_valid = valid
        r.pdf = FloatC._wrap(torch.where(valid, pdf0 * ps.pdf.t * d2 / cos_t, torch.zeros_like(d2)))
        return r

    def to_string(self):
        return "Scene[\n  # Sensors\n%s\n  # BSDFs\n%s\n  # Meshes\n%s\n]" % tuple(
            "\n".join("  " + x.to_string() for x in arr) for arr in (self.m_sensors, self.m_bsdfs, self.m_meshes))

    # -- descriptor for the C ABI ------------------------------------------------
    def tables(self, sensor_id=0, capacity=False):
        """Flat dict of tensors for one sensor (scene tables + that sensor's camera/primary edges).  After a native configure() the edge tables
        are stored at the capacity of their candidate lists with the kept rows first and the count on the device: capacity=True hands them out
        as they stand (the render calls: no read-back), the default cuts them to the kept rows -- the tables the reference holds
        (scene.cpp:219-244, perspective.cpp:96-111); that reads the counts once per configure().
        NORMALISED FORM of a native configure(): `sec_pmf` / `prim_pmf` are divided by their sum and `sec_cmf` / `prim_cmf` run to 1 (so `sec_sum = prim_sum = 1.0`,
        where the reference's tables hold lengths and their total); the true summed length stays on the device in `sec_header[1]` / `prim_header[1]` (a float;
        `[0]` = the number of kept rows as integer bits).  A table in which NO edge is kept is all zero rows with pmf 0: the kernels treat pmf 0 -- and, for a capacity-one
        table, length 0 -- as an invalid draw (csrc/psdr_device.h primary_edge_sample; tests/test_edge_cases_gpu.py)."""
        psdr_assert(self._configured, "Input scene must be configured!")
        psdr_assert(0 <= sensor_id < self.num_sensors, "Invalid sensor id!")
        t = dict(self._tables)
        t.update(self._sensor_tables[sensor_id])
        t["width"], t["height"] = self.opts.width, self.opts.height
        t["num_meshes"], t["num_bsdfs"] = len(self.m_meshes), len(self.m_bsdfs)
        if t.get("device_counts") and not capacity:
            n_sec, n_prim = self._kept_edge_counts()
            if t.get("sec_header") is not None:
                if n_sec > 0:
                    t.update(sec_edge=t["sec_edge"][:n_sec], sec_cmf=t["sec_cmf"][:n_sec], sec_pmf=t["sec_pmf"][:n_sec],
                             sec_edge_faces=t["sec_edge_faces"][:n_sec], num_sec_edges=n_sec)
                else:
                    t.update(sec_edge=None, sec_cmf=None, sec_pmf=None, sec_sum=0.0, num_sec_edges=0, sec_edge_faces=None)
            if t.get("prim_header") is not None:
                n = n_prim[sensor_id]
                if n > 0:
                    t.update(prim_edge=t["prim_edge"][:n], prim_cmf=t["prim_cmf"][:n], prim_pmf=t["prim_pmf"][:n], num_prim_edges=n)
                    if t.get("prim_edge_z") is not None:
                        t["prim_edge_z"] = t["prim_edge_z"][:n]
                else:
                    t.update(prim_edge=None, prim_cmf=None, prim_pmf=None, prim_sum=0.0, num_prim_edges=0, prim_edge_z=None)
        return t


def make_desc(tb, guide=None, device=None):
    """SceneDesc (+ list of tensors kept alive) from a tables dict.  Pointers are device pointers
    if the tensors live on the GPU, host pointers otherwise (oracle)."""
    keep = []

    def p(x, dtype=torch.float32):
        if x is None:
            return None
        x = x.detach()
        if device is not None:
            x = x.to(device)
        x = x.to(dtype).contiguous()
        keep.append(x)
        return x.data_ptr()
    d = _abi.SceneDesc()
    d.width, d.height = tb["width"], tb["height"]
    d.num_tris, d.num_meshes, d.num_bsdfs, d.num_emitters = tb["num_tris"], tb["num_meshes"], tb["num_bsdfs"], tb["num_emitters"]
    d.num_sec_edges, d.num_prim_edges = tb["num_sec_edges"], tb["num_prim_edges"]
    d.num_texels = int(tb["texels"].numel())
    d.tri_info, d.tri_uv = p(tb["tri_info"]), p(tb["tri_uv"])
    i32 = torch.int32
    d.tri_mesh, d.mesh_bsdf, d.mesh_emitter = p(tb["tri_mesh"], i32), p(tb["mesh_bsdf"], i32), p(tb["mesh_emitter"], i32)
    d.bsdf_rec, d.texels = p(tb["bsdf_rec"], i32), p(tb["texels"])
    d.emitter_f, d.emitter_i = p(tb["emitter_f"]), p(tb["emitter_i"], i32)
    d.face_cmf, d.face_pmf = p(tb["face_cmf"]), p(tb["face_pmf"])
    d.emitter_cmf, d.emitter_pmf, d.emitter_sum = p(tb["emitter_cmf"]), p(tb["emitter_pmf"]), tb["emitter_sum"]
    d.cam = p(tb["cam"])
    d.sec_edge, d.sec_cmf, d.sec_pmf, d.sec_sum = p(tb["sec_edge"]), p(tb["sec_cmf"]), p(tb["sec_pmf"]), tb["sec_sum"]
    d.sec_edge_faces = p(tb.get("sec_edge_faces"), torch.int32)
    d.prim_edge_z = p(tb.get("prim_edge_z"))
    d.prim_edge, d.prim_cmf, d.prim_pmf, d.prim_sum = p(tb["prim_edge"]), p(tb["prim_cmf"]), p(tb["prim_pmf"]), tb["prim_sum"]
    d.material_mask = int(tb.get("material_mask", 0))
    d.env_emitter = int(tb.get("env_emitter", -1))
    if d.env_emitter >= 0:
        for i in range(3):
            d.env_tex[i] = int(tb["env_tex"][i])
        d.env_reso[0], d.env_reso[1] = int(tb["env_reso"][0]), int(tb["env_reso"][1])
        d.env_f, d.env_cmf, d.env_pmf, d.env_sum = p(tb["env_f"]), p(tb["env_cmf"]), p(tb["env_pmf"]), float(tb["env_sum"])
    if guide is not None:
        reso, cmf, pmf, s = guide
        d.guide_reso[0], d.guide_reso[1], d.guide_reso[2] = int(reso[0]), int(reso[1]), int(reso[2])
        d.num_guide_cells = int(reso[0]) * int(reso[1]) * int(reso[2])
        d.guide_cmf, d.guide_pmf, d.guide_sum = p(cmf), p(pmf), float(s)
    return d, keep
