"""The C-ABI shared library loads without a GPU and exports every symbol include/psdr_hip.h declares."""
import ctypes as C
import os
import re

from psdr_cuda import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "psdr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psdr_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = _abi.load_hip()
    names = header_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "libpsdr_hip.so does not export " + n
    assert sorted(_abi.HIP_SYMBOLS) == names


def test_struct_mirrors_match_the_header():
    lib = _abi.load_hip()
    sizes = (C.c_int32 * 4)()
    assert lib.psdr_abi_struct_sizes(sizes) == 0
    assert tuple(sizes) == (C.sizeof(_abi.SceneDesc), C.sizeof(_abi.RenderOpts), C.sizeof(_abi.Tangents), C.sizeof(_abi.Grads))
    import oracle
    orc = oracle.load_oracle()
    osz = (C.c_int32 * 4)()
    orc.psdr_oracle_struct_sizes(osz)
    assert tuple(osz) == tuple(sizes)


def test_version_and_error_strings_without_gpu():
    lib = _abi.load_hip()
    assert b"gfx950" in lib.psdr_version()
    # null arguments are rejected with a message, no crash, no GPU needed
    assert lib.psdr_scene_set_tables(None, None) != 0
    assert b"null" in lib.psdr_last_error()
    assert lib.psdr_render_c(None, None, None, None) != 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "psdr-cuda_amd", "psdr_cuda")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            txt = open(os.path.join(pkg, f)).read()
            assert not any(k in txt for k in ("load_oracle", "import oracle", "libpsdr_oracle", "ORACLE_LIB")), f
    csrc = os.path.join(ROOT, "psdr-cuda_amd", "csrc")
    for f in os.listdir(csrc):
        assert "oracle" not in open(os.path.join(csrc, f)).read().lower().replace("test-only", ""), f
