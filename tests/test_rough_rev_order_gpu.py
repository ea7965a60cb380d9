"""The order-dependent camera-pose gradient of round 3, root-caused in round 4 (DESIGN.md "the order-dependent gradient"):

  * mechanism: a register-allocator defect of the ROCm 7.2 LLVM on gfx950 -- in a kernel that spills SGPRs and VGPRs, the FIRST spill store of a
    value that lived in registers until then can be placed in the prologue of a reconvergence block, behind an SGPR spill (v_writelane) and in
    FRONT of the `s_or_b64 exec` that re-enables the lanes of the other side of the branch.  The masked lanes never store; the reload runs under
    the full mask and hands them what the scratch arena held: zero in a fresh process, stale data after kernels with larger scratch frames;
  * what was hit: seven per-lane accumulators of the camera-matrix gradient (DeviceSink::cam[0..6]) in k_camera_rev<10, true, PATH> when the
    register accumulators (RegPrivSink) AND the DPP totals were both enabled in the rough-conductor instance;
  * the guard: tools/check_spill_exec.py scans every kernel of the built library for that shape (run by build(), tests/test_spill_exec_guard.py).

This test (1) fills the scratch arena with NaN / 123.0 and checks the product's rough-conductor reverse kernels against forward mode, (2) runs
the same on the test-only build of the library in which flag set 10 is compiled the way that exposed the defect (tests/poison/
libpsdr_hip_defect.so, made by build()): whenever that build computes a gradient that moves with the scratch contents, the scanner must have
flagged the kernel -- every observed failure is explained by the shape the guard looks for."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["texels", "emitter_rad", "tri_info", "cam_to_world"]


def run_case(pattern):
    """cbox_rough, PathTracer(3), reverse mode with every gradient after the scratch arena was filled with `pattern`: returns the gradients and
    the dot-product identity against forward mode per table."""
    from helpers import GpuScene, dot_tables, poison_gpu
    from psdr_cuda import _abi
    import test_reverse_mode as trm
    tb, o, adj = trm._setup("cbox_rough", dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3), 0, 0, res=32, spp=8)
    g = GpuScene(tb)
    poison_gpu(pattern, 1)
    _, grads = g.render_d_rev(o, adj, want=NAMES)
    out = {}
    for n in NAMES:
        tan = trm._tangents(tb, n)
        _, dimg = g.render_d_fwd(o, [tan])
        lhs = float((adj.astype(np.float64) * dimg[0]).sum())
        out[n] = (lhs, dot_tables(grads, tan), float(np.abs(adj.astype(np.float64) * dimg[0]).sum()))
    return out, {n: np.asarray(grads[n], np.float64).ravel() for n in NAMES}


def main():
    for p in ("psdr-cuda_amd", "tests", "oracle"):
        sys.path.insert(0, os.path.join(ROOT, p))
    res = {}
    for pattern in (0x0, 0x7fc00000, 0x42f60000):
        ident, grads = run_case(pattern)
        res["%08x" % pattern] = grads
        for n, (lhs, rhs, scale) in ident.items():
            print("pattern %08x %-13s forward %+.6e reverse %+.6e  err/scale %.1e" % (pattern, n, lhs, rhs, abs(lhs - rhs) / max(scale, 1e-30)))
    np.savez(sys.argv[1], **{"%s_%s" % (k, n): v for k, g in res.items() for n, v in g.items()})


def _run(lib, path):
    env = dict(os.environ)
    if lib:
        env["PSDR_HIP_LIB"] = lib
    else:
        env.pop("PSDR_HIP_LIB", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return np.load(path), r.stdout


def _moved(got):
    """tables whose gradient depends on what the scratch arena held"""
    out = []
    for n in NAMES:
        clean = got["00000000_%s" % n]
        for pat in ("7fc00000", "42f60000"):
            g = got["%s_%s" % (pat, n)]
            if not np.allclose(g, clean, rtol=1e-3, atol=1e-3 * max(np.abs(clean).max(), 1e-30), equal_nan=False):
                out.append((n, pat))
    return out


def test_product_gradients_do_not_depend_on_the_scratch_contents(tmp_path):
    got, log = _run(None, str(tmp_path / "product.npz"))
    print(log)
    assert _moved(got) == []
    # and they are right: the dot-product identity against forward mode, whatever the arena held
    for line in log.splitlines():
        if line.startswith("pattern"):
            assert float(line.split("err/scale")[1]) < 1e-3, line


def test_defect_build_every_failure_is_explained_by_the_spill_shape(tmp_path):
    lib = os.path.join(ROOT, "tests", "poison", "libpsdr_hip_defect.so")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_defect_lib()          # rebuilt from the product's cached objects when stale; a box without them (the objects do not travel) uses the shipped file
    assert os.path.exists(lib), "__graft_entry__.build_defect_lib() (also run by build()) makes tests/poison/libpsdr_hip_defect.so"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_spill_exec as cse
    hits = cse.check_library(lib, verbose=True)
    flagged = [h for h in hits if "k_camera_rev" in h[1] and "ILi10ELb1ELi1ELi0" in h[1]]          # k_camera_rev<10, true, PATH, 0>
    got, log = _run(lib, str(tmp_path / "defect.npz"))
    print(log)
    moved = _moved(got)
    print("defect build: scanner flagged %d kernel(s) (%d the all-gradients rough PathTracer kernel); tables that move with the scratch pattern: %s" % (len(hits), len(flagged), moved))
    if moved:
        assert flagged, "a gradient of the defect build depends on the scratch contents but the scanner found no first-spill in front of an exec restore"
        assert {n for n, _ in moved} == {"cam_to_world"}                # the seven spilled accumulators are DeviceSink::cam[0..6]
    # (a compiler or source change may move the register allocation of this translation unit off the defect: then nothing moves and nothing has to be flagged)
    assert bool(flagged) == bool(moved)


if __name__ == "__main__":
    main()
