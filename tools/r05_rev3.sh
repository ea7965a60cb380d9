#!/bin/bash
# experiment: the reverse camera kernels compiled for three waves per SIMD (variants/lib_rev3.so: no register accumulators) against the product's two
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05rev3}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {  # name lib options case
  rm -rf /tmp/vr
  PSDR_HIP_LIB=$2 PSDR_OPTIONS="$3" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vr -o p -- python $R/tools/wf_case.py $4 default 5 > /tmp/vr.log 2>&1
  f=$(find /tmp/vr -name "*kernel_stats.csv" | head -1)
  python - "$f" "$4 $1 [$3]" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace)::k_" in r["Name"] and "refit" not in r["Name"] and "bvh4" not in r["Name"] and "gather_top" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
print("%-60s total %7.2f ms/call | " % (sys.argv[2], tot) + " | ".join("%s %.1f us x%d" % (r["Name"].replace("void (anonymous namespace)::","").split("(")[0][:34], float(r["AverageNs"]) / 1e3, int(r["Calls"]) // 6) for r in rows if "rev" in r["Name"]))
PY
}
for c in c2ra c4pr; do
  run base "" "" $c
  run base "" "sink_private=0,rev_sorted=0" $c
  run rev3 $R/variants/lib_rev3.so "sink_private=0,rev_sorted=0" $c
  run rev3 $R/variants/lib_rev3.so "rev_sorted=0" $c
  if [ $c = c2ra ]; then run rev3-split $R/variants/lib_rev3.so "sink_private=0,rev_sorted=0,rev_split=1" $c; run base-split "" "rev_split=1" $c; fi
done 2>&1 | tee -a $O/abk.txt
