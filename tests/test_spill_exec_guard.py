"""The build-time guard against the compiler defect behind the order-dependent gradient (tools/check_spill_exec.py): the scanner
on the two shapes it has to tell apart, and the product library itself (no kernel may spill a slot for the first time in front of an exec restore)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_spill_exec as cse  # noqa: E402

DEFECT = """
_Z6kernelv:
	v_mov_b32_e32 v88, 0
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_f32_e32 v88, v88, v1
.LBB0_2:
	v_writelane_b32 v255, s34, 21
	scratch_store_dword off, v88, off offset:196 ; 4-byte Folded Spill
	v_writelane_b32 v255, s35, 22
	s_or_b64 exec, exec, s[4:5]
	v_mov_b32_e32 v56, -1
	scratch_load_dword v88, off, off offset:196 ; 4-byte Folded Reload
	s_endpgm
"""
# the value is defined inside the region (both arms of an inner uniform branch) and stored there under the region's mask; the other lanes stored
# the slot where they defined it (the store in front of the branch)
BENIGN = """
_Z6kernelv:
	v_mov_b32_e32 v13, 0
	scratch_store_dword off, v13, off offset:236 ; 4-byte Folded Spill
	s_and_saveexec_b64 s[14:15], vcc
	s_cbranch_execz .LBB0_4
; %bb.1:
	s_cbranch_vccnz .LBB0_3
; %bb.2:
	global_load_dword v13, v[90:91], off
	s_branch .LBB0_4
.LBB0_3:
	v_mov_b32_e32 v13, 0
.LBB0_4:
	s_waitcnt vmcnt(0)
	scratch_store_dword off, v13, off offset:236 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[14:15]
	s_endpgm
"""


def test_scanner_flags_a_first_spill_in_front_of_the_exec_restore():
    hits = cse.scan_asm(DEFECT)
    assert len(hits) == 1 and hits[0][0] == "_Z6kernelv" and len(hits[0][2]) == 1


def test_scanner_accepts_a_slot_that_has_another_store_site():
    assert cse.scan_asm(BENIGN) == []


def test_product_library_is_free_of_the_defect():
    lib = os.path.join(ROOT, "psdr-cuda_amd", "lib", "libpsdr_hip.so")
    cos = cse.code_objects(lib)
    assert len(cos) >= 10                               # eight flag sets + the host unit + the table chain
    assert cse.check_library(lib, verbose=True) == []


def test_guard_fails_closed_when_it_cannot_look(tmp_path, monkeypatch):
    """a library without a readable code object, a failing objdump or an unparsable disassembly is an ERROR, never "0 hits"."""
    import pytest
    # (1) no gfx code object in the file (e.g. a compressed bundle this reader does not unpack)
    empty = tmp_path / "nothing.o"
    empty.write_bytes(b"\x7fELF" + b"\0" * 64 + b"CCOB" + b"\0" * 64)
    with pytest.raises(cse.GuardError, match="code object"):
        cse.check_library(str(empty), verbose=False)
    # (2) fewer code objects than the caller knows the library must hold
    lib = os.path.join(ROOT, "psdr-cuda_amd", "lib", "libpsdr_hip.so")
    with pytest.raises(cse.GuardError):
        cse.check_library(lib, verbose=False, min_code_objects=10 ** 6)
    # (3) objdump fails
    co = cse.code_objects(lib)[0]
    bad = tmp_path / "objdump"
    bad.write_text("#!/bin/sh\nexit 3\n"); bad.chmod(0o755)
    monkeypatch.setattr(cse, "find_objdump", lambda: str(bad))
    with pytest.raises(cse.GuardError, match="failed"):
        cse._scan_code_object(co)
    # (4) objdump succeeds but prints nothing the scanner understands
    mute = tmp_path / "objdump2"
    mute.write_text("#!/bin/sh\necho file format elf64-amdgpu\n"); mute.chmod(0o755)
    monkeypatch.setattr(cse, "find_objdump", lambda: str(mute))
    with pytest.raises(cse.GuardError, match="nothing parsed"):
        cse._scan_code_object(co)


def test_objdump_is_found_through_the_toolchain_not_a_fixed_path(monkeypatch):
    p = cse.find_objdump()
    assert os.path.isfile(p) and os.access(p, os.X_OK)
    monkeypatch.setenv("ROCM_PATH", "/nonexistent")          # falls through to hipcc's tree / the default root
    assert os.path.isfile(cse.find_objdump())
