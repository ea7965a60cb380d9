#!/usr/bin/env python
"""Static instruction / scratch counts of the kernels of one flag set (device assembly of csrc/psdr_variant.hip): the check to run after ANY change of the
shared device headers -- the headline kernels sit on spill cliffs, and a refactor that looks neutral (round 5: intersect() returning a second hit record by
value) put 21 scratch instructions into the C2 renderC kernel (+2.4 % instructions, 4x the counter traffic, -5 % on the headline).
usage: isa_counts.py <flag set> [name filter] [--against other.s]   (compiles to /tmp/isa_<flags>.s; with --against prints both)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero "
         "-fno-slp-vectorize -freciprocal-math -fapprox-func").split()


def counts(path, flt):
    t = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+): ", t, re.M):
        end = t.find(".Lfunc_end", m.end())
        lines = [l.strip() for l in t[m.end():end].split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        out[m.group(1)] = (len(lines), sum(l.startswith("v_") for l in lines), sum("scratch_" in l for l in lines), sum(l.startswith("ds_") for l in lines))
    names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.split("\n")
    return {n.replace("(anonymous namespace)::", "").replace("psdr::", "").replace("void ", "").split("(")[0]: v for n, v in zip(names, out.values()) if flt in n}


if __name__ == "__main__":
    fl = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "k_camera<"
    dst = "/tmp/isa_%s.s" % fl
    subprocess.check_call(["hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-DPSDR_VARIANT_FLAGS=" + fl, "-S", "--cuda-device-only",
                           os.path.join(ROOT, "psdr-cuda_amd", "csrc", "psdr_variant.hip"), "-o", dst], stderr=subprocess.DEVNULL)
    now = counts(dst, flt)
    ref = counts(sys.argv[sys.argv.index("--against") + 1], flt) if "--against" in sys.argv else {}
    for n, v in sorted(now.items()):
        r = ref.get(n)
        print("%-64s instr %6d valu %6d scratch %4d lds %4d%s" % (n[:64], v[0], v[1], v[2], v[3], "" if r is None else ("   | other: %6d %6d %4d %4d%s" % (r + ("  <-- differs" if r != v else "",)))))
