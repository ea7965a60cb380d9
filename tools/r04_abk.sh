#!/bin/bash
# A/B of library variants by KERNEL time (rocprofv3 --kernel-trace --stats): r04_abk.sh <tag> "<names>" "<cases>" [mode]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
MODE=${4:-wavefront}
for c in $3; do
  for v in base $2; do
    if [ $v = base ]; then L=""; else L=$R/variants/lib_$v.so; fi
    rm -rf /tmp/abk_$v
    PSDR_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -o p -- python $R/tools/wf_case.py $c $MODE 5 > /tmp/abk_$v.log 2>&1
    f=$(find /tmp/abk_$v -name "*kernel_stats.csv" | head -1)
    python - "$f" "$v $c" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace)::k_" in r["Name"] and "refit" not in r["Name"] and "bvh4" not in r["Name"] and "gather_top" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
print("%-16s total %7.2f ms/call | " % (sys.argv[2], tot) + " | ".join("%s %.1f us x%d" % (r["Name"].replace("void (anonymous namespace)::","").split("(")[0][:40], float(r["AverageNs"]) / 1e3, int(r["Calls"]) // 6) for r in rows))
PY
  done
done | tee -a $O/abk.txt
