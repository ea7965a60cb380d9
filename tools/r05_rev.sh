#!/bin/bash
# round 5: reverse kernels with deferred, sorted row adds (rev_sorted) against scattering on the spot -- reverse-mode tests + kernel times
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05r}; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -k "reverse or projection or inverse or rough_rev or full_size or split or config5 or tables_native or partial or dot_product or vertex or ad_vs_fd or edge_cases" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -6 $O/tests.log
cd /tmp; export TMPDIR=/tmp
for c in ${2:-c2ra c4pr c5pr c3r}; do
  for opt in "rev_sorted=0" "rev_sorted=1"; do
    rm -rf /tmp/vr
    PSDR_OPTIONS="$opt" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vr -o p -- python $R/tools/wf_case.py $c default 5 > /tmp/vr.log 2>&1
    f=$(find /tmp/vr -name "*kernel_stats.csv" | head -1)
    python - "$f" "$c $opt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous namespace)::k_" in r["Name"] and "refit" not in r["Name"] and "bvh4" not in r["Name"] and "gather_top" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
print("%-22s total %7.2f ms/call | " % (sys.argv[2], tot) + " | ".join("%s %.1f us x%d" % (r["Name"].replace("void (anonymous namespace)::","").split("(")[0][:40], float(r["AverageNs"]) / 1e3, int(r["Calls"]) // 6) for r in rows if float(r["TotalDurationNs"]) > 3e5))
PY
  done
done 2>&1 | tee -a $O/abk.txt
