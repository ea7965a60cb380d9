#!/usr/bin/env python
"""Condense the rocprofv3 passes of ONE tools/prof_case.py workload (tools/r03_tree_prof.sh) into one row per kernel of this library.
Every pass runs the workload three times; counters and durations are SUMMED over all launches of a kernel in a pass and divided by the
three calls: `per call` = what one render call spends in that kernel (all its stages / chunks together).  VALU issue fraction against
the SIMD-32 peak (1228.8 G wave-instructions/s), wait fractions of the wave cycles, L2 hit rate, HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE)
KiB (MI355X_MICROARCH.md)."""
import collections, csv, glob, sys
name, prefix = sys.argv[1], sys.argv[2]
calls = 3.0
dur, cnt, nl = collections.defaultdict(list), collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(int)
def ours(k):
    return ("(anonymous namespace)::k_" in k or k.startswith("k_")) and "refit" not in k and "lbvh" not in k and "rocprim" not in k and "at::native" not in k
for d in glob.glob(prefix + "*"):
    if d.endswith(".log"):
        continue
    per = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                per[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in per.items():
        dur[k].append(v)
    seen = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                cnt[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[r["Kernel_Name"]][r["Counter_Name"]] += 1
    for k, c in seen.items():
        nl[k] = max(nl[k], max(c.values()))
print("## %s" % name)
for k, c in cnt.items():
    ds = sorted(dur.get(k, [0.0]))
    ns = ds[len(ds) // 2] / calls                      # per call, median over the passes
    short = k.replace("void (anonymous namespace)::", "").replace("psdr::", "").split("(")[0][:56] or "k_trace"
    g = lambda n: c.get(n, float("nan")) / calls
    valu, wc = g("SQ_INSTS_VALU"), g("SQ_WAVE_CYCLES")
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    hbm = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
    print("%-58s launches/call %4.1f  %8.3f ms/call | VALU %.4g winst, issue frac %.3f | SALU %.3g LDS %.3g SMEM %.3g VMEM_RD %.3g VMEM_WR %.3g | "
          "WAIT_ANY %.3f WAIT_INST_ANY %.3f ACTIVE_VALU %.3f of wave cycles | L2 hit %.3f | HBM %.4g B/call = %.1f GB/s (%.4f of 8 TB/s)"
          % (short, nl[k] / calls, ns / 1e6, valu, valu / (ns * 1e-9) / 1228.8e9 if ns else 0, g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_VMEM_RD"),
             g("SQ_INSTS_VMEM_WR"), g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, hit / (hit + miss), hbm,
             hbm / (ns * 1e-9) / 1e9 if ns else 0, hbm / (ns * 1e-9) / 8e12 if ns else 0)
          + (" | LDS_IDX_ACTIVE %.3g BANK_CONFLICT %.3g cycles/call, busy cycles %.4g" % (g("SQ_LDS_IDX_ACTIVE"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_BUSY_CYCLES")) if "SQ_LDS_IDX_ACTIVE" in c else ""))
