#!/usr/bin/env python
"""Condense gpurun_out/prof/<tag>/ into profiles/<tag>_summary.txt (our kernels only)."""
import csv, glob, os, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof", tag)
out = [("# rocprofv3 summary '%s' (bench.py; kernel-trace --stats run = 5 timed steps + 2 warmup; each PMC pass = separate 1-step run)" % tag)]
ks = os.path.join(src, "kernel_stats.csv")
if os.path.exists(ks):
    out.append("\n## kernel-trace --stats (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev)")
    for r in csv.reader(open(ks)):
        if r and (any(k in r[0] for k in ("k_camera", "k_trace", "k_primary", "k_secondary", "k_guide", "k_wf_", "k_wfg_", "k_direct_probe", "k_se_probe", "k_tangent_live")) or r[0] == "Name"):
            name = r[0].replace("void (anonymous namespace)::", "").split("(LaunchCtx")[0].split("((anonymous")[0].split("(psdr::SceneView")[0]
            out.append("%-40s %s" % (name, " ".join(r[1:])))
out.append("\n## PMC (per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts 64 B per 128 B request -> double it for wide reads, MI355X_MICROARCH.md)")
for f in sorted(glob.glob(os.path.join(src, "pmc_*.txt"))):
    for line in open(f):
        if "k_" in line and "namespace" in line:
            parts = line.split()
            name = line.split("(anonymous namespace)::")[1].split("((anonymous")[0].split("(")[0] if "(anonymous namespace)::" in line else parts[0]
            i = [k for k, p in enumerate(parts) if p.startswith("dispatches=")][0]
            out.append("%-28s %-22s %s %s" % (name, parts[i - 1], parts[i], parts[i + 2]))
for f in sorted(glob.glob(os.path.join(src, "*bench.log"))):
    for line in open(f):
        if line.startswith("{"):
            out.append("\n## bench line of the kernel-trace run\n" + line.strip())
# per-kernel HBM traffic per launch: (2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes.  FETCH_SIZE is doubled
# because gfx950's counter tallies 64 B per 128 B wide request (MI355X_MICROARCH.md, HBM section).
import json, re
traffic = {}
for f in ("pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt"):
    fp = os.path.join(src, f)
    if not os.path.exists(fp):
        continue
    for line in open(fp):
        m = re.search(r"(k_[a-z_]+<[^(]*>|k_[a-z_]+)\(.*?(FETCH_SIZE|WRITE_SIZE)\s+dispatches=(\d+)\s+total=([0-9.e+]+)\s+per_dispatch=([0-9.e+]+)", line)
        if m:
            traffic.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(5))
# VALU wave-instructions per launch (SQ_INSTS_VALU): bench.py turns them into the VALU issue fraction
valu = {}
fp = os.path.join(src, "pmc_SQ_WAVES.txt")
if os.path.exists(fp):
    for line in open(fp):
        m = re.search(r"(k_[a-z_]+<[^(]*>|k_[a-z_]+)\(.*?SQ_INSTS_VALU\s+dispatches=(\d+)\s+total=([0-9.e+]+)\s+per_dispatch=([0-9.e+]+)", line)
        if m:
            valu[m.group(1).strip()] = float(m.group(4))
tj = {k: {"fetch_kib": v.get("FETCH_SIZE"), "write_kib": v.get("WRITE_SIZE"),
          "hbm_bytes_per_launch": (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0,
          "valu_wave_insts_per_launch": valu.get(k)} for k, v in traffic.items()}
os.makedirs("profiles", exist_ok=True)
json.dump({"tag": tag, "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate 1-step passes of bench.py; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": tj},
          open(os.path.join("profiles", tag + "_traffic.json"), "w"), indent=1)
json.dump({"tag": tag, "traffic": tag + "_traffic.json", "summary": tag + "_summary.txt"}, open(os.path.join("profiles", "latest.json"), "w"))
open(os.path.join("profiles", tag + "_summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
