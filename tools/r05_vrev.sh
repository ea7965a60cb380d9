#!/bin/bash
# round 5: per-vertex adjoint launches (k_vertex_rev) against the one adjoint kernel -- parity tests + kernel times of the reverse workloads
# usage (through gpurun): tools/r05_vrev.sh <tag> ["cases"]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05v}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_reverse_mode.py tests/test_edge_cases_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -15 $O/tests.log
cd /tmp; export TMPDIR=/tmp
for c in ${2:-c2ra c4pr c5pr}; do
  for opt in "rev_vertex=0" "rev_vertex=1"; do
    extra=""; if [ $c = c2ra ]; then extra=",rev_split=1"; fi
    if [ $c = c2ra ] && [ "$opt" = "rev_vertex=0" ]; then extra=""; fi      # the round-4 default on C2: one fused kernel
    rm -rf /tmp/vr
    PSDR_OPTIONS="$opt$extra" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vr -o p -- python $R/tools/wf_case.py $c default 5 > /tmp/vr.log 2>&1
    tail -1 /tmp/vr.log
    f=$(find /tmp/vr -name "*kernel_stats.csv" | head -1)
    python - "$f" "$c $opt$extra" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace)::k_" in r["Name"] and "refit" not in r["Name"] and "bvh4" not in r["Name"] and "gather_top" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
print("%-28s total %7.2f ms/call | " % (sys.argv[2], tot) + " | ".join("%s %.1f us x%d" % (r["Name"].replace("void (anonymous namespace)::","").split("(")[0][:40], float(r["AverageNs"]) / 1e3, int(r["Calls"]) // 6) for r in rows))
PY
  done
done 2>&1 | tee -a $O/abk.txt
