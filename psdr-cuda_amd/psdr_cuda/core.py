"""Small host-side value types of the psdr_cuda surface (reference src/psdr.cpp:48-178)."""
import numpy as np
import torch

import enoki as ek
from enoki.cuda import Float32 as FloatC, Vector2f as Vector2fC, Vector3f as Vector3fC, Int32 as IntC  # noqa: F401
from enoki.cuda_autodiff import Float32 as FloatD, Vector2f as Vector2fD, Vector3f as Vector3fD  # noqa: F401
from enoki.cuda_autodiff import Matrix4f as Matrix4fD  # noqa: F401


class Exception_(RuntimeError):
    """psdr::Exception (include/misc/Exception.h:86-132) -> Python RuntimeError."""


def psdr_assert(cond, msg="Assertion failed"):
    if not cond:
        raise RuntimeError(msg)


class Object:
    """reference include/psdr/object.h:5-23"""
    _type_name = "Object"

    def __init__(self):
        self.id = ""

    def type_name(self):
        return self._type_name

    def log(self, msg):
        print("[%s] %s" % (self._type_name, msg))

    def to_string(self):
        return self._type_name

    def __repr__(self):
        return self.to_string()


class RenderOption:
    """reference include/psdr/types.h:171-182, src/psdr.cpp:53-72"""

    def __init__(self, width=128, height=128, spp=1, sppe=None, sppse=None):
        self.width, self.height, self.spp = int(width), int(height), int(spp)
        self.sppe = int(spp if sppe is None else sppe)
        self.sppse = int(self.sppe if sppse is None else sppse)   # reference leaves it uninitialised in the 3/4-arg ctor
        self.log_level = 1
        # the reference's compile-time variant PSDR_PRIMARY_EDGE_VIS_CHECK (include/psdr/macros.h:13) as a run-time option:
        # a primary-edge sample counts only if the edge point itself is visible from the camera
        self.primary_edge_vis_check = False

    def __repr__(self):
        return "[width: %d, height: %d, spp: %d, sppe: %d, sppse: %d, log_level: %d]" % (
            self.width, self.height, self.spp, self.sppe, self.sppse, self.log_level)


def _to_tensor(x, cols=None):
    if isinstance(x, ek.ArrayBase):
        t = x.t
    elif isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(np.asarray(x, dtype=np.float32), device=ek.default_device())
    return t


# ------------------------------------------------------------------ rays, frames, records
def _make_ray(name, V3, F):
    class _Ray:
        """Ray<ad> (reference include/psdr/core/ray.h:8-29; src/psdr.cpp:76-86): o, d, tmax, reversed()."""

        def __init__(self, o=None, d=None, tmax=None):
            self.o, self.d = o, d
            if tmax is None and o is not None:
                tmax = F._wrap(torch.full((ek.slices(o),), float("inf"), device=o.t.device))
            self.tmax = tmax

        def __call__(self, t):
            return V3._wrap(self.d.t * (t.t if isinstance(t, ek.ArrayBase) else t).reshape(-1, 1) + self.o.t)

        def reversed(self):
            return type(self)(self.o, V3._wrap(-self.d.t), self.tmax)
    _Ray.__name__ = _Ray.__qualname__ = name
    return _Ray


def _make_frame(name, V3):
    class _Frame:
        """Frame_ (reference include/psdr/core/frame.h:9-60; src/psdr.cpp:88-100): orthonormal basis around n
        by Duff et al. (coordinate_system), to_local / to_world."""

        def __init__(self, v=None):
            self.s = self.t = self.n = None
            if v is not None:
                n = v.t if isinstance(v, ek.ArrayBase) else torch.as_tensor(v, dtype=torch.float32).reshape(-1, 3)
                nx, ny, nz = n[:, 0], n[:, 1], n[:, 2]
                sign = torch.where(nz >= 0, torch.ones_like(nz), -torch.ones_like(nz))
                a = -1.0 / (sign + nz)
                b = nx * ny * a
                self.s = V3._wrap(torch.stack([sign * nx * nx * a + 1.0, sign * b, -sign * nx], dim=-1))
                self.t = V3._wrap(torch.stack([b, sign + ny * ny * a, -ny], dim=-1))
                self.n = V3._wrap(n)

        def to_local(self, v):
            return V3._wrap(torch.stack([(v.t * self.s.t).sum(-1), (v.t * self.t.t).sum(-1), (v.t * self.n.t).sum(-1)], dim=-1))

        def to_world(self, v):
            return V3._wrap(self.s.t * v.t[:, 0:1] + self.t.t * v.t[:, 1:2] + self.n.t * v.t[:, 2:3])
    _Frame.__name__ = _Frame.__qualname__ = name
    return _Frame


RayC, RayD = _make_ray("RayC", Vector3fC, FloatC), _make_ray("RayD", Vector3fD, FloatD)
FrameC, FrameD = _make_frame("FrameC", Vector3fC), _make_frame("FrameD", Vector3fD)


class SampleRecordC:
    """SampleRecord_ (reference include/psdr/core/records.h:10-17): pdf, is_valid"""
    pdf = is_valid = None


class SampleRecordD(SampleRecordC):
    pass


class PositionSampleC(SampleRecordC):
    """PositionSample_ (records.h:20-32): pdf, is_valid, p, n, J"""
    p = n = J = None


class PositionSampleD(SampleRecordD):
    p = n = J = None


class _Bitmap(Object):
    """Bitmap<channels> (reference include/psdr/core/bitmap.h:10-37, src/core/bitmap.cpp:9-89).
    `data` is an enoki-shim array: Float32 (1 channel) or Vector3f (3 channels) with w*h slices."""
    channels = 1
    _type_name = "Bitmap"

    def __init__(self, *args):
        super().__init__()
        self.resolution = (1, 1)
        if len(args) == 0:
            self.fill(0.0)
        elif len(args) == 1 and isinstance(args[0], str):
            self.load_openexr(args[0])
        elif len(args) == 1:
            self.fill(args[0])
        elif len(args) == 3:
            w, h, data = args
            self.resolution = (int(w), int(h))
            self.data = data
            psdr_assert(w * h == ek.slices(self._data))
        else:
            raise TypeError("Bitmap: unsupported constructor arguments")

    def fill(self, value):
        self.resolution = (1, 1)
        if self.channels == 1:
            self._data = FloatD(float(np.asarray(value).reshape(-1)[0]))
        else:
            v = np.asarray(value, dtype=np.float32).reshape(-1)
            if v.size == 1:
                v = np.repeat(v, 3)
            self._data = Vector3fD([float(v[0]), float(v[1]), float(v[2])])

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, value):
        cls = FloatD if self.channels == 1 else Vector3fD
        self._data = value if isinstance(value, cls) else cls(value)

    def load_openexr(self, file_name):
        from .exr import load_exr_rgba
        rgba, (w, h) = load_exr_rgba(file_name)
        self.resolution = (w, h)
        flat = rgba.reshape(-1, 4)
        if self.channels == 1:
            self._data = FloatD(flat[:, 0].copy())
        else:
            self._data = Vector3fD(torch.as_tensor(flat[:, :3].copy(), device=ek.default_device()))

    def tensor(self):
        """[w*h, channels] float32 tensor (keeps the autograd graph)."""
        t = self._data.t
        t = t.reshape(-1, 1) if self.channels == 1 else t.reshape(-1, 3)
        w, h = self.resolution
        if t.shape[0] != w * h:
            raise RuntimeError("Bitmap: invalid data size!")
        if (w, h) != (1, 1) and (w < 2 or h < 2):
            raise RuntimeError("Bitmap: invalid resolution!")
        return t

    def eval(self, uv, flip_v=True):
        """Bilinear lookup, reference src/core/bitmap.cpp:41-89 (host/torch implementation for API parity)."""
        t = self.tensor()
        w, h = self.resolution
        uvt = uv.t if isinstance(uv, ek.ArrayBase) else torch.as_tensor(uv)
        if (w, h) == (1, 1):
            out = t.expand(uvt.shape[0], -1)
        else:
            u, v = uvt[:, 0], uvt[:, 1]
            if flip_v:
                v = -v
            u = u - torch.floor(u); v = v - torch.floor(v)
            u = u * (w - 1); v = v * (h - 1)
            px, py = torch.floor(u).long(), torch.floor(v).long()
            w1x, w1y = (u - px).unsqueeze(-1), (v - py).unsqueeze(-1)
            px = torch.clamp(px, max=w - 2); py = torch.clamp(py, max=h - 2)
            idx = py * w + px
            v0 = (1 - w1x) * t[idx] + w1x * t[idx + 1]
            v1 = (1 - w1x) * t[idx + w] + w1x * t[idx + w + 1]
            out = (1 - w1y) * v0 + w1y * v1
        return FloatD._wrap(out[:, 0]) if self.channels == 1 else Vector3fD._wrap(out)


class Bitmap1fD(_Bitmap):
    channels = 1
    _type_name = "Bitmap1fD"


class Bitmap3fD(_Bitmap):
    channels = 3
    _type_name = "Bitmap3fD"


class DiscreteDistribution:
    """reference include/psdr/core/pmf.h:8-25, src/core/pmf.cpp:7-50 (host mirror; the kernels do the
    per-sample binary search on the cmf/pmf tables this class produces)."""

    def __init__(self):
        self.m_size = 0
        self._sum, self._sum_dev = 0.0, None
        self.m_pmf = None
        self.m_cmf = None

    # the sum may still be on the device (init_device): read when somebody asks
    @property
    def m_sum(self):
        if self._sum_dev is not None:
            self._sum, self._sum_dev = float(self._sum_dev.item()), None
        return self._sum

    @m_sum.setter
    def m_sum(self, v):
        self._sum, self._sum_dev = float(v), None

    def init_device(self, pmf, cmf, total):
        """tables built by a kernel (csrc/psdr_tables.hip); total: a host float, or a one-element device tensor that is read on first use"""
        self.m_size = int(pmf.shape[0])
        self.m_pmf, self.m_cmf = pmf, cmf
        if isinstance(total, torch.Tensor):
            self._sum_dev = total
        else:
            self.m_sum = total

    def init(self, pmf, total=None):
        """total: the sum of pmf when the caller already holds it on the host (Scene.configure reads every size and sum of a
        configure back in one batch): no device-to-host read here."""
        t = pmf.t.detach() if isinstance(pmf, ek.ArrayBase) else torch.as_tensor(pmf).detach()
        t = t.to(torch.float32).reshape(-1)
        self.m_size = int(t.shape[0])
        self.m_pmf = t.contiguous()
        if total is not None:
            self.m_sum = float(total)
        else:
            self.m_sum = float(t.sum().item()) if self.m_size else 0.0
        self.m_cmf = torch.cumsum(t, dim=0).contiguous()      # enoki::psum = inclusive prefix sum

    @property
    def sum(self):
        return FloatC([self.m_sum])

    def pmf(self):
        return FloatC(self.m_pmf)

    def sample(self, samples):
        s = samples.t if isinstance(samples, ek.ArrayBase) else torch.as_tensor(samples)
        if self.m_size == 1:
            return IntC(torch.zeros_like(s, dtype=torch.int32)), FloatC(torch.ones_like(s))
        u = s * self.m_sum
        idx = torch.searchsorted(self.m_cmf, u.contiguous(), right=False).clamp(max=self.m_size - 1)
        return IntC(idx.to(torch.int32)), FloatC(self.m_pmf[idx] / self.m_sum)


class HyperCubeDistribution3f:
    """reference include/psdr/core/cube_distrb.h:9-27, src/core/cube_distrb.cpp:8-62"""
    ndim = 3

    def __init__(self):
        self.m_resolution = None
        self.m_num_cells = 0
        self.m_distrb = DiscreteDistribution()
        self.m_ready = False

    def set_resolution(self, reso):
        reso = [int(r) for r in reso][: self.ndim]
        if self.m_resolution != reso:
            n = 1
            for r in reso:
                n *= r
            psdr_assert(n < 2 ** 31 - 1)
            self.m_resolution = reso
            self.m_num_cells = n
            self.m_ready = False

    @property
    def cells(self):
        idx = torch.arange(self.m_num_cells, device=ek.default_device())
        out = []
        for i in range(self.ndim):
            d = 1
            for r in self.m_resolution[i + 1:]:
                d *= r
            out.append((idx // d) % self.m_resolution[i])
        return torch.stack(out, dim=-1).to(torch.int32)

    def set_mass(self, pmf):
        t = pmf.t if isinstance(pmf, ek.ArrayBase) else torch.as_tensor(pmf)
        psdr_assert(t.numel() == self.m_num_cells)
        self.m_distrb.init(t)
        self.m_ready = True

    def pdf(self, p):
        psdr_assert(self.m_ready)
        pt = p.t if isinstance(p, ek.ArrayBase) else torch.as_tensor(p)
        reso = torch.tensor(self.m_resolution, device=pt.device)
        ip = torch.floor(pt * reso).long()
        valid = ((ip >= 0) & (ip < reso)).all(dim=-1)
        idx = ip[:, 0]
        for i in range(1, self.ndim):
            idx = idx * self.m_resolution[i] + ip[:, i]
        idx = idx.clamp(0, self.m_num_cells - 1)
        val = self.m_distrb.m_pmf[idx] / self.m_distrb.m_sum * self.m_num_cells
        return FloatC(torch.where(valid, val, torch.zeros_like(val)))


class HyperCubeDistribution2f(HyperCubeDistribution3f):
    ndim = 2
