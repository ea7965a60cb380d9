#!/bin/bash
# round 4, session 5: chunk size of the traced wavefront (psdr_scene_set_option chunk_log2) by kernel time AND wall time per call (launch gaps count at small chunks)
# usage (through gpurun): tools/r04_chunk_sweep.sh <tag> "<cases>" "<chunk_log2 values>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in $2; do
  for k in $3; do
    if [ "$k" = "-" ]; then opt=""; else opt="chunk_log2=$k"; fi
    w=$(PSDR_OPTIONS=$opt timeout 300 python $R/tools/wf_case.py $c ${MODE:-wavefront} 7 2>&1 | tail -1)
    rm -rf /tmp/cks
    PSDR_OPTIONS=$opt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cks -o p -- python $R/tools/wf_case.py $c ${MODE:-wavefront} 5 > /tmp/cks.log 2>&1
    f=$(find /tmp/cks -name "*kernel_stats.csv" | head -1)
    python - "$f" "$c [$opt]" "$w" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace)::k_" in r["Name"] and "refit" not in r["Name"] and "bvh4" not in r["Name"] and "gather_top" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
print("%-22s kernels %7.2f ms/call | " % (sys.argv[2], tot) + " | ".join("%s %.2f ms x%d" % (r["Name"].replace("void (anonymous namespace)::","").split("(")[0].split("<")[0], float(r["TotalDurationNs"]) / 6e6, int(r["Calls"]) // 6) for r in rows[:3]) + " | wall: " + sys.argv[3])
PY
  done
done | tee -a $O/sweep.txt
