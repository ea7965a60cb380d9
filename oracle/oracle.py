"""Python front-end of the CPU oracle (oracle/psdr_oracle.cpp).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "psdr-cuda_amd"))

from psdr_cuda import _abi  # noqa: E402
from psdr_cuda.scene import make_desc  # noqa: E402


ORACLE_LIB_PATH = os.path.join(_HERE, "libpsdr_oracle.so")
_oracle = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load_oracle():
    """Load oracle/libpsdr_oracle.so (tests / smoke / cpu_baseline only; the product package has no
    reference to it)."""
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_LIB_PATH):
        raise RuntimeError("oracle not built: run `make -C oracle`")
    L = C.CDLL(ORACLE_LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    SceneDesc, RenderOpts, Tangents = _abi.SceneDesc, _abi.RenderOpts, _abi.Tangents
    L.psdr_oracle_last_error.restype = C.c_char_p
    L.psdr_oracle_trace.argtypes = [C.POINTER(SceneDesc), i32] + [vp] * 7 + [vp] * 4
    L.psdr_oracle_render.argtypes = [C.POINTER(SceneDesc), C.POINTER(RenderOpts), i32, C.POINTER(Tangents), vp, vp,
                                     i32, i32]
    L.psdr_oracle_guide_build.argtypes = [C.POINTER(SceneDesc), C.POINTER(RenderOpts), C.POINTER(i32), i32, vp, i32]
    L.psdr_oracle_rng.argtypes = [C.c_uint64, C.c_uint64, i32, vp]
    L.psdr_oracle_pcg32_raw.argtypes = [C.c_uint64, C.c_uint64, i32, vp]
    L.psdr_oracle_sample_reuse.argtypes = [vp, vp, C.c_float, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.psdr_oracle_sample_reuse.restype = i32
    L.psdr_oracle_draws_per_camera_sample.argtypes = [C.POINTER(RenderOpts)]
    L.psdr_oracle_draws_per_camera_sample.restype = i32
    L.psdr_oracle_set_reference_form.argtypes = [i32]
    L.psdr_oracle_set_reference_form.restype = None
    _oracle = L
    return L


def lib():
    if not os.path.exists(ORACLE_LIB_PATH):
        build()
    return load_oracle()


def _cpu_tables(tb):
    return {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}


def render(tb, opts, mode=0, tangents=None, guide=None, precision=0, nthreads=None, reference_form=False):
    """mode 0: renderC -> img [H*W,3]; mode 1: renderD forward -> (img, dimg).
    tangents: dict name -> tensor (names of _abi.TANGENT_FIELDS).
    reference_form: evaluate the reference's literal expressions (p = ray(t) for the solid-angle hit, edge rays that
    may re-hit the faces adjacent to their edge) instead of the fp32-robust forms the product uses; identical in
    exact arithmetic (psdr_oracle.cpp g_reference_form)."""
    L = lib()
    L.psdr_oracle_set_reference_form(1 if reference_form else 0)
    tb = _cpu_tables(tb)
    if guide is not None:
        guide = (guide[0], guide[1].cpu(), guide[2].cpu(), guide[3])
    desc, keep = make_desc(tb, guide, device="cpu")
    n = tb["width"] * tb["height"] * 3
    img = np.zeros(n, dtype=np.float32)
    dimg = np.zeros(n, dtype=np.float32)
    tan = _abi.Tangents()
    if tangents:
        for name, t in tangents.items():
            if t is not None:
                t = t.detach().cpu().to(torch.float32).contiguous()
                keep.append(t)
                setattr(tan, "d_" + name, t.data_ptr())
    nthreads = nthreads or os.cpu_count() or 1
    rc = L.psdr_oracle_render(C.byref(desc), C.byref(opts), mode, C.byref(tan), img.ctypes.data,
                              dimg.ctypes.data if mode else None, precision, nthreads)
    if rc:
        raise RuntimeError(L.psdr_oracle_last_error().decode())
    return (img.reshape(-1, 3), dimg.reshape(-1, 3)) if mode else img.reshape(-1, 3)


def trace(tb, o, d, tmax=None):
    L = lib()
    desc, keep = make_desc(_cpu_tables(tb), None, device="cpu")
    o = np.ascontiguousarray(o, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32)
    m = o.shape[0]
    tmax = np.full(m, np.inf, dtype=np.float32) if tmax is None else np.ascontiguousarray(tmax, dtype=np.float32)
    cols = [np.ascontiguousarray(o[:, i]) for i in range(3)] + [np.ascontiguousarray(d[:, i]) for i in range(3)]
    shape = np.zeros(m, np.int32); tri = np.zeros(m, np.int32); u = np.zeros(m, np.float32); v = np.zeros(m, np.float32)
    L.psdr_oracle_trace(C.byref(desc), m, *[c.ctypes.data for c in cols], tmax.ctypes.data, shape.ctypes.data,
                        tri.ctypes.data, u.ctypes.data, v.ctypes.data)
    return shape, tri, u, v


def guide_build(tb, opts, reso, nrounds, nthreads=None):
    L = lib()
    desc, keep = make_desc(_cpu_tables(tb), None, device="cpu")
    cells = int(reso[0]) * int(reso[1]) * int(reso[2])
    mass = np.zeros(cells, dtype=np.float32)
    r = (C.c_int32 * 4)(*[int(x) for x in reso])
    L.psdr_oracle_guide_build(C.byref(desc), C.byref(opts), r, int(nrounds), mass.ctypes.data, nthreads or os.cpu_count())
    return mass


def rng(slot, offset, n):
    out = np.zeros(n, dtype=np.float32)
    lib().psdr_oracle_rng(int(slot), int(offset), n, out.ctypes.data)
    return out
