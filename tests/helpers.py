"""Shared helpers for the test-suite: scene set-up, oracle / hostcheck / GPU render front-ends."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

import enoki as ek
import psdr_cuda
from enoki._array import _jvp_wrt
from enoki.cuda_autodiff import Float32 as FloatD, Vector3f as Vector3fD, Matrix4f as Matrix4fD
from psdr_cuda import _abi
from psdr_cuda.fixtures import scene_path

# developer switch of the TEST / TOOL drivers (the package itself reads no environment variable): PSDR_HIP_LIB=<path> runs the tests and tools on
# another build of the library (tools/build_variant_lib.sh: A/B runs inside one gpurun call)
if os.environ.get("PSDR_HIP_LIB") and _abi._hip is None:
    _abi.use_library(os.environ["PSDR_HIP_LIB"])
from psdr_cuda.scene import make_desc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AD_KEYS = list(_abi.TANGENT_FIELDS)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def same_rays(a, b):
    """Ray counters of two launch forms of the same samples.  Since round 6 a light sample the BSDF's cosine tests zero is not traced; whether cos theta is > 0 is decided
    on a vertex each form reconstructs with its own fp32 arithmetic (the fused kernel from the ray, a wavefront stage from its stream record, the adjoint sweep from its path
    record), so a handful of samples per million sit on the other side of zero: equal to 1e-4, not to the ray."""
    return abs(int(a) - int(b)) <= 1e-4 * max(int(a), int(b), 1)


def isolated_pixels_unbiased(a, ref, bad, label="", bias_bound=1e-3):
    """Tree-scene tests compare two fp32 evaluations of the same estimator on the same random numbers and EXCLUDE the isolated pixels where one sample resolved an
    epsilon-sized tie the other way (`bad`: boolean mask over pixels).  Excluding is only sound if those pixels are what that name says -- this is the net under the
    exclusion: their SIGNED error must sum to nothing,
        |sum_bad (a - ref)|  <=  3 sqrt(sum_bad (a - ref)^2) + 1e-6 sum|ref|       three sigma of a zero-mean flip model (independent flips of random sign)
        |sum_bad (a - ref)|  <=  bias_bound * sum|ref|                              and negligible against the image
    per colour channel.  A kernel bug confined to the excluded pixels (a rare branch that always loses or always gains energy) shows as a one-signed sum: with n
    excluded pixels of similar error e it is n e against 3 sqrt(n) e.  Returns the worst ratio of the second bound (printed by the callers)."""
    a, ref = np.asarray(a, np.float64).reshape(-1, 3), np.asarray(ref, np.float64).reshape(-1, 3)
    bad = np.asarray(bad).reshape(-1)
    tot = np.abs(ref).sum(axis=0) + 1e-30
    if not bad.any():
        return 0.0
    e = (a - ref)[bad]
    s, sig = e.sum(axis=0), np.sqrt((e * e).sum(axis=0))
    worst = float((np.abs(s) / tot).max())
    assert (np.abs(s) <= 3.0 * sig + 1e-6 * tot).all(), "%s: the %d excluded pixels are biased: signed sum %s against 3 sigma %s" % (label, int(bad.sum()), s, 3 * sig)
    assert worst <= bias_bound, "%s: the %d excluded pixels carry %.2e of the image's energy in one direction" % (label, int(bad.sum()), worst)
    return worst


def load_scene(name, res=32, spp=8, sppe=0, sppse=0, translate=None, device=None):
    """Scene fixture at a small resolution.  translate = (mesh_id, direction): returns a scalar
    parameter P (FloatD, requires grad) that translates that mesh by direction * P."""
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(name), False)
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    P = None
    if translate is not None:
        P = FloatD(0.)
        ek.set_requires_gradient(P)
        sc.param_map["Mesh[%d]" % translate[0]].set_transform(Matrix4fD.translate(Vector3fD(list(translate[1])) * P))
    sc.configure()
    return sc, P


def tangents_wrt(tb, P):
    return dict(zip(AD_KEYS, _jvp_wrt([tb.get(k) for k in AD_KEYS], P.t)))


# ---------------------------------------------------------------- scratch / LDS / VGPR poison (tests/poison/poison.hip)
_poison = None


def poison_gpu(pattern, what=1):
    """Fill the queue's scratch arena (what & 1), the LDS of every CU (& 2), the VGPRs later waves inherit (& 4) with a bit pattern."""
    global _poison
    if _poison is None:
        d = os.path.join(ROOT, "tests", "poison")
        so, src = os.path.join(d, "libpoison.so"), os.path.join(d, "poison.hip")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-fPIC", "-shared", src, "-o", so])
        _poison = C.CDLL(so)
    rc = _poison.poison_gpu(C.c_uint32(pattern), int(what))
    assert rc == 0, "poison_gpu failed (%d)" % rc


# ---------------------------------------------------------------- hostcheck (product code on the CPU)
_hostcheck = None


def hostcheck_lib():
    global _hostcheck
    if _hostcheck is None:
        d = os.path.join(ROOT, "tests", "hostcheck")
        so = os.path.join(d, "libhostcheck.so")
        src = os.path.join(d, "hostcheck.cpp")
        hdrs = [os.path.join(ROOT, "psdr-cuda_amd", "csrc", f) for f in ("psdr_math.h", "psdr_device.h", "psdr_bvh_build.h")]
        if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                                   src, "-o", so])
        _hostcheck = C.CDLL(so)
    return _hostcheck


def host_render(tb, opts, mode=0, tangents=None, guide=None, nthreads=None):
    H = hostcheck_lib()
    tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
    if guide is not None:
        guide = (guide[0], guide[1].cpu(), guide[2].cpu(), guide[3])
    desc, keep = make_desc(tbc, guide, device="cpu")
    n = tb["width"] * tb["height"] * 3
    img, dimg = np.zeros(n, np.float32), np.zeros(n, np.float32)
    tan = _abi.Tangents()
    for k, t in (tangents or {}).items():
        if t is not None:
            t = t.detach().cpu().float().contiguous()
            keep.append(t)
            setattr(tan, "d_" + k, t.data_ptr())
    rc = H.hostcheck_render(C.byref(desc), C.byref(opts), mode, C.byref(tan), C.c_void_p(img.ctypes.data),
                            C.c_void_p(dimg.ctypes.data), nthreads or os.cpu_count())
    assert rc == 0
    return (img.reshape(-1, 3), dimg.reshape(-1, 3)) if mode else img.reshape(-1, 3)


def _grad_buffers(tb, want):
    bufs, g = {}, _abi.Grads()
    for name in want:
        t = tb.get(name)
        if t is None:
            continue
        bufs[name] = np.zeros(tuple(t.shape), dtype=np.float32)
        setattr(g, "g_" + name, bufs[name].ctypes.data)
    return bufs, g


def host_render_rev(tb, opts, adj, want=AD_KEYS, guide=None):
    """Reverse mode of the product code on the host: returns (img, {table: gradient})."""
    H = hostcheck_lib()
    tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
    if guide is not None:
        guide = (guide[0], guide[1].cpu(), guide[2].cpu(), guide[3])
    desc, keep = make_desc(tbc, guide, device="cpu")
    bufs, g = _grad_buffers(tbc, want)
    adj = np.ascontiguousarray(adj, dtype=np.float32).reshape(-1)
    img = np.zeros(adj.shape[0], np.float32)
    rc = H.hostcheck_render_rev(C.byref(desc), C.byref(opts), C.c_void_p(adj.ctypes.data), C.c_void_p(img.ctypes.data), C.byref(g))
    assert rc == 0
    return img.reshape(-1, 3), bufs


def random_tangents(tb, names, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n in names:
        t = tb.get(n)
        if t is not None:
            out[n] = (torch.rand(t.shape, generator=g) - 0.5).float()
    return out


def dot_tables(grads, tangents):
    return float(sum((np.asarray(grads[n], dtype=np.float64) * tangents[n].detach().cpu().numpy().astype(np.float64)).sum()
                     for n in tangents if n in grads))


# ---------------------------------------------------------------- GPU through the C ABI
class GpuScene:
    """Thin ctypes driver of libpsdr_hip.so (exactly what a foreign-language binding would do)."""

    def __init__(self, tb, guide=None, options=None):
        self.lib = _abi.load_hip()
        self.h = C.c_void_p()
        _abi.check(self.lib, self.lib.psdr_scene_create(C.byref(self.h)))
        # developer options of the handle (psdr_scene_set_option): the `options` dict of this call, then PSDR_OPTIONS="name=value,..." of
        # the environment (tools: A/B runs without touching the scripts) -- read HERE, in the test driver, never by the library
        opts = dict(options or {})
        for kv in filter(None, os.environ.get("PSDR_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            opts.setdefault(k.strip(), float(v))
        for k, v in opts.items():
            self.set_option(k, v)
        self.tb = {k: (v.detach().cuda() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
        self.set_guide(guide)
        _abi.check(self.lib, self.lib.psdr_bvh_build(self.h, None))

    def set_option(self, name, value):
        _abi.check(self.lib, self.lib.psdr_scene_set_option(self.h, name.encode(), float(value)))

    def set_guide(self, guide):
        if guide is not None:
            guide = (guide[0], torch.as_tensor(guide[1]).cuda(), torch.as_tensor(guide[2]).cuda(), guide[3])
        self.desc, self.keep = make_desc(self.tb, guide)
        _abi.check(self.lib, self.lib.psdr_scene_set_tables(self.h, C.byref(self.desc)))

    def close(self):
        if self.h:
            self.lib.psdr_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def n(self):
        return self.tb["width"] * self.tb["height"] * 3

    def render_c(self, opts):
        img = torch.empty(self.n, dtype=torch.float32, device="cuda")
        _abi.check(self.lib, self.lib.psdr_render_c(self.h, C.byref(opts), img.data_ptr(), None))
        torch.cuda.synchronize()
        return img.cpu().numpy().reshape(-1, 3)

    def render_d_fwd(self, opts, tangent_sets):
        K = len(tangent_sets)
        img = torch.empty(self.n, dtype=torch.float32, device="cuda")
        dimg = torch.empty(K * self.n, dtype=torch.float32, device="cuda")
        tarr = (_abi.Tangents * K)()
        keep = []
        for k, ts in enumerate(tangent_sets):
            for name, t in ts.items():
                if t is not None:
                    t = t.detach().cuda().float().contiguous()
                    keep.append(t)
                    setattr(tarr[k], "d_" + name, t.data_ptr())
        _abi.check(self.lib, self.lib.psdr_render_d_fwd(self.h, C.byref(opts), K, tarr, img.data_ptr(), dimg.data_ptr(), None))
        torch.cuda.synchronize()
        return img.cpu().numpy().reshape(-1, 3), dimg.cpu().numpy().reshape(K, -1, 3)

    def render_d_rev(self, opts, adj, want=AD_KEYS, with_image=True):
        adj_t = torch.as_tensor(np.ascontiguousarray(adj, dtype=np.float32).reshape(-1)).cuda()
        img = torch.empty(self.n, dtype=torch.float32, device="cuda") if with_image else None
        g = _abi.Grads()
        bufs = {}
        for name in want:
            t = self.tb.get(name)
            if t is None:
                continue
            bufs[name] = torch.zeros(tuple(t.shape), dtype=torch.float32, device="cuda")
            setattr(g, "g_" + name, bufs[name].data_ptr())
        _abi.check(self.lib, self.lib.psdr_render_d_rev(self.h, C.byref(opts), adj_t.data_ptr(), img.data_ptr() if with_image else None,
                                                        C.byref(g), None))
        torch.cuda.synchronize()
        return (img.cpu().numpy().reshape(-1, 3) if with_image else None), {k: v.cpu().numpy() for k, v in bufs.items()}

    def trace(self, o, d, tmax=None):
        o = torch.as_tensor(np.ascontiguousarray(o, dtype=np.float32)).cuda()
        d = torch.as_tensor(np.ascontiguousarray(d, dtype=np.float32)).cuda()
        m = o.shape[0]
        cols = [o[:, i].contiguous() for i in range(3)] + [d[:, i].contiguous() for i in range(3)]
        tm = torch.full((m,), float("inf"), device="cuda") if tmax is None else torch.as_tensor(tmax, dtype=torch.float32).cuda()
        shape = torch.empty(m, dtype=torch.int32, device="cuda"); tri = torch.empty_like(shape)
        u = torch.empty(m, dtype=torch.float32, device="cuda"); v = torch.empty_like(u)
        _abi.check(self.lib, self.lib.psdr_trace(self.h, m, *[c.data_ptr() for c in cols], tm.data_ptr(), shape.data_ptr(),
                                                 tri.data_ptr(), u.data_ptr(), v.data_ptr(), None))
        torch.cuda.synchronize()
        return shape.cpu().numpy(), tri.cpu().numpy(), u.cpu().numpy(), v.cpu().numpy()

    def guide_build(self, opts, reso, nrounds):
        cells = int(reso[0]) * int(reso[1]) * int(reso[2])
        mass = torch.zeros(cells, dtype=torch.float32, device="cuda")
        r = (C.c_int32 * 4)(*[int(x) for x in reso])
        _abi.check(self.lib, self.lib.psdr_guide_build(self.h, C.byref(opts), r, int(nrounds), mass.data_ptr(), None))
        torch.cuda.synchronize()
        return mass.cpu().numpy()

    def counters(self):
        c = (C.c_uint64 * 4)()
        _abi.check(self.lib, self.lib.psdr_get_counters(self.h, c))
        return tuple(int(x) for x in c)


def camera_rays(tb, n, seed=0):
    """n random primary rays of the scene's camera (numpy)."""
    rng = np.random.default_rng(seed)
    cam = tb["cam"].detach().cpu().numpy().astype(np.float64)
    s2c, tw = cam[0:16].reshape(4, 4), cam[16:32].reshape(4, 4)
    s = rng.random((n, 2))
    v = np.concatenate([s, np.zeros((n, 1)), np.ones((n, 1))], axis=1) @ s2c.T
    d = v[:, :3] / v[:, 3:4]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dw = d @ tw[:3, :3].T
    o = np.broadcast_to(tw[:3, 3], (n, 3))
    return o.astype(np.float32), dw.astype(np.float32)
