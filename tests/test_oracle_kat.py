"""Known-answer tests that pin the oracle's primitives (the reference holds no golden
vectors; PCG32 is the one third-party algorithm with a PUBLISHED test vector)."""
import ctypes as C

import numpy as np

import oracle
from psdr_cuda import _abi


def test_pcg32_published_vector():
    # pcg-c-basic demo, pcg32_srandom(42, 54): first six 32-bit outputs
    out = np.zeros(6, dtype=np.uint32)
    oracle.lib().psdr_oracle_pcg32_raw(42, 54, 6, out.ctypes.data)
    assert [hex(x) for x in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_stream_floats_in_unit_interval_and_streams_differ():
    a, b = oracle.rng(0, 0, 4096), oracle.rng(1, 0, 4096)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 0.02 and abs(np.corrcoef(a, b)[0, 1]) < 0.05


def test_jump_ahead_equals_stepping():
    full = oracle.rng(12345, 0, 300)
    for off in (1, 7, 64, 255):
        assert np.array_equal(oracle.rng(12345, off, 300 - off), full[off:])


def test_sample_reuse_matches_definition():
    rng = np.random.default_rng(0)
    pmf = rng.random(37).astype(np.float32) + 0.01
    pmf[5] = 0.0
    cmf = np.cumsum(pmf, dtype=np.float32)
    s = float(pmf.sum(dtype=np.float32))
    L = oracle.lib()
    for u0 in rng.random(500).astype(np.float32):
        u, p = C.c_float(float(u0)), C.c_float(0)
        idx = L.psdr_oracle_sample_reuse(cmf.ctypes.data, pmf.ctypes.data, s, 37, C.byref(u), C.byref(p))
        t = np.float32(u0) * np.float32(s)
        ref = min(int(np.searchsorted(cmf, t, side="left")), 36)      # first i with cmf[i] >= t
        assert idx == ref
        assert 0.0 <= u.value <= 1.0 and abs(p.value - pmf[idx] / s) < 1e-6
    # a single-entry distribution returns (0, 1) and leaves the sample untouched (pmf.cpp:31-33)
    u, p = C.c_float(0.3), C.c_float(0)
    one = np.ones(1, dtype=np.float32)
    assert L.psdr_oracle_sample_reuse(one.ctypes.data, one.ctypes.data, 1.0, 1, C.byref(u), C.byref(p)) == 0
    assert abs(u.value - 0.3) < 1e-7 and p.value == 1.0


def test_draw_counts():
    L = oracle.lib()
    for kw, n in ((dict(bsdf_samples=1, light_samples=1), 7), (dict(bsdf_samples=2, light_samples=2), 12),
                  (dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3), 17), (dict(integrator=_abi.INTEGRATOR_FIELD), 2)):
        o = _abi.make_opts(**kw)
        assert L.psdr_oracle_draws_per_camera_sample(C.byref(o)) == n == _abi.draws_per_slot(o)[0]
