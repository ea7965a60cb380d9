#!/bin/bash
# A/B of handle options by KERNEL time (rocprofv3 --kernel-trace --stats): r04_abk_opt.sh <tag> "<cases>" "<options A>" "<options B>" ...  ("-" = defaults)
# PSDR_HIP_LIB (optional) picks the library
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CASES=$2; shift; shift
for c in $CASES; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then opt=""; else opt="$v"; fi
    rm -rf /tmp/abko
    PSDR_OPTIONS=$opt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abko -o p -- python $R/tools/wf_case.py $c wavefront 5 > /tmp/abko.log 2>&1
    f=$(find /tmp/abko -name "*kernel_stats.csv" | head -1)
    python - "$f" "$c [$opt]" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace)::k_" in r["Name"] and "refit" not in r["Name"] and "bvh4" not in r["Name"] and "gather_top" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
print("%-28s total %7.2f ms/call | " % (sys.argv[2], tot) + " | ".join("%s %.1f us x%d" % (r["Name"].replace("void (anonymous namespace)::","").split("(")[0][:40], float(r["AverageNs"]) / 1e3, int(r["Calls"]) // 6) for r in rows[:4]))
PY
  done
done | tee -a $O/abk.txt
