#!/bin/bash
# kernel-trace stats of one wf_case: r04_prof.sh <tag> <case> <mode>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/wf_case.py $2 $3 3 > $O/prof.log 2>&1
tail -1 $O/prof.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-70s calls %5s  avg %10.1f us  total %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O/prof -name "*.db" -delete
