# usage (through gpurun): bash tools/prof_case_pmc.sh <scene> <direct|path> <c|fwd|rev> [res spp depth]  -> VALU / wait counters of our kernels
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rm -rf /tmp/pc_$N
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d /tmp/pc_$N -o p -- python $R/tools/prof_case.py "$@" > /tmp/pc_$N.log 2>&1
  python - /tmp/pc_$N <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = collections.defaultdict(list)
for f in kt:
    for r in csv.DictReader(open(f)):
        if "k_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:50]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in cc:
    for r in csv.DictReader(open(f)):
        if "k_" in r["Kernel_Name"] and "refit" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    ns = dur.get(k, [0])[-1]
    print(k, "last launch %.3f ms" % (ns / 1e6))
    for n, v in c.items():
        extra = ""
        if n == "SQ_INSTS_VALU" and ns:
            extra = "  -> VALU issue %.2f" % (v[-1] * 4 / (ns * 1e-9 * 2.4e9 * 1024))
        print("   %-22s %.4g%s" % (n, v[-1], extra))
PY
done
