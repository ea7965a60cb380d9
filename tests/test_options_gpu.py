"""psdr_scene_set_option: the developer switches of a handle (the library reads no environment variable)."""
import numpy as np
import pytest

from helpers import GpuScene, load_scene, rel_l2
from psdr_cuda import _abi

pytestmark = pytest.mark.gpu


def test_unknown_option_fails_and_known_ones_switch_strategies():
    sc, _ = load_scene("cbox_bunny", res=64, spp=16)
    tb = sc.tables(0)
    g = GpuScene(tb)
    assert g.lib.psdr_scene_set_option(g.h, b"no_such_option", 1.0) != 0
    assert b"unknown option" in g.lib.psdr_last_error()
    o = _abi.make_opts(integrator=_abi.INTEGRATOR_PATH, max_depth=3, spp=16, flags=_abi.FLAG_WAVEFRONT)
    stats = _abi.scene_stats(g.h)
    assert stats.get("n_blas", 0) == 1                                  # the two-level tree of a room with one object
    a = g.render_c(o)
    for opts in ({"wf_traced": 0}, {"wf_traced": 0, "wf_binned": 0}, {"two_level": 0}):
        g2 = GpuScene(tb, options=opts)
        if "two_level" in opts:
            assert _abi.scene_stats(g2.h).get("n_blas", 0) == 0
        b = g2.render_c(o)
        bad = np.abs(a - b).max(1) > 1e-5 * (1.0 + np.abs(a).max(1))
        assert bad.mean() < 2e-3 and rel_l2(b[~bad], a[~bad]) < 1e-4, (opts, bad.mean())


def test_library_reads_no_environment_variable():
    """No PSDR_* name survives in the binary (17 getenv knobs lived in the hot host path)."""
    import subprocess
    out = subprocess.run(["strings", "-a", _abi.HIP_LIB_PATH], capture_output=True, text=True).stdout
    names = [l for l in out.splitlines() if l.startswith("PSDR_") and l.replace("_", "").isalnum() and l.isupper()]
    assert names == [], names
