"""Static instruction / scratch counts of the headline kernels, pinned.

The hot kernels sit on spill cliffs and share their device headers with every other kernel: twice in round 5 a refactor that looked neutral changed them (intersect()
returning a second hit record by value: 21 scratch instructions in the C2 renderC kernel, +2.4 % instructions, 4x the counter traffic, -5 % on the headline;
`*next = its1` on a dual record: 300 B of scratch traffic per path vertex).  This test disassembles the kernels of flag sets 8 (no tree: BASELINE configs 1 / 2) and 4
(two-level tree: configs 3 / 4) from the BUILT library and compares instruction and scratch-instruction counts with tests/golden/isa_counts.json: +-1 % instructions,
no new scratch instruction.  A deliberate change of a kernel regenerates the table:  python tests/test_isa_guard.py --write  (and says so in the commit).
CPU test: reads psdr-cuda_amd/lib/obj/variant<N>.o (what build() leaves here) or, without the objects, the code objects inside libpsdr_hip.so.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
TABLE = os.path.join(ROOT, "tests", "golden", "isa_counts.json")
# kernel (demangled, without namespaces) -> what it is
KERNELS = {
    "k_camera<float, float, 1, 8, true>": "C2 renderC (PathTracer, scene without a tree)",
    "k_camera_logd<1, 8, true>": "C2 renderD forward K = 1, log-derivative kernel (the headline's dominant kernel)",
    "k_camera_logd<3, 8, true>": "C2 renderD forward K = 3",
    "k_camera_rev<8, false, 1, 0>": "C2 reverse, texel gradient",
    "k_camera_rev<8, true, 1, 2>": "C2 reverse, adjoint kernel on kept records (all gradients)",
    "k_wf_bounce<float, 4, true, false>": "C4 traced wavefront, bounce stage",
    "k_wf_camera<float, 4, true, false>": "C4 traced wavefront, camera stage",
    "k_camera_rev<4, true, 1, 2>": "C4 PathTracer reverse, adjoint kernel",
    "k_primary_edge_rev<4, 0>": "C3 / C4 primary-edge term, reverse",
    "k_primary_edge<1, 4, 0>": "C3 primary-edge term, forward K = 1",
}


def disassemble():
    import check_spill_exec as cse
    objdump = cse.find_objdump()
    blobs = []
    for v in (8, 4):
        o = os.path.join(ROOT, "psdr-cuda_amd", "lib", "obj", "variant%d.o" % v)
        if os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(os.path.join(ROOT, "psdr-cuda_amd", "lib", "libpsdr_hip.so")) - 3600:
            blobs += cse.code_objects(o)
    if not blobs:
        blobs = cse.code_objects(os.path.join(ROOT, "psdr-cuda_amd", "lib", "libpsdr_hip.so"))
    assert blobs, "no gfx950 code object found"
    import tempfile
    out = {}
    for b in blobs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(b); f.flush()
            txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for l in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
            if m:
                cur = m.group(1)
                out.setdefault(cur, [0, 0])
                continue
            if cur and re.match(r"^\s+[a-z]", l):
                ins = l.strip().split()[0]
                if ins.startswith("s_nop") or ins.startswith("s_code_end"):
                    continue
                out[cur][0] += 1
                out[cur][1] += 1 if ins.startswith("scratch_") else 0
    names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.split("\n")
    res = {}
    for mangled, dem in zip(out, names):
        short = dem.replace("(anonymous namespace)::", "").replace("psdr::", "").replace("void ", "").split("(")[0]
        if short in KERNELS and out[mangled][0] > 100:
            res[short] = {"instructions": out[mangled][0], "scratch": out[mangled][1]}
    return res


def test_headline_kernels_keep_their_instruction_and_scratch_counts():
    now = disassemble()
    want = json.load(open(TABLE))
    missing = sorted(set(KERNELS) - set(now))
    assert not missing, "kernels not found in the library: %s" % missing
    bad = []
    for k in KERNELS:
        a, b = now[k], want[k]
        if abs(a["instructions"] - b["instructions"]) > 0.01 * b["instructions"] or a["scratch"] > b["scratch"]:
            bad.append("%s (%s): %d instructions / %d scratch, table %d / %d" % (k, KERNELS[k], a["instructions"], a["scratch"], b["instructions"], b["scratch"]))
    assert not bad, "kernels moved (a deliberate change regenerates the table: python tests/test_isa_guard.py --write):\n  " + "\n  ".join(bad)


if __name__ == "__main__":
    if "--write" in sys.argv:
        t = disassemble()
        json.dump(t, open(TABLE, "w"), indent=1, sort_keys=True)
    for k, v in sorted(disassemble().items()):
        print("%-44s %6d instructions %4d scratch   %s" % (k, v["instructions"], v["scratch"], KERNELS[k]))
