#!/bin/bash
# Timeline of one bench step (developer tool, via gpurun): every kernel between two renderC launches with its duration and the idle gap before it.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/btl
rocprofv3 --kernel-trace --output-format csv -d /tmp/btl -o t -- python $R/bench.py --steps 6 --warmup 3 --no-pmc --no-cpu-baseline > /tmp/btl.log 2>&1
F=$(find /tmp/btl -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_camera<float, float, 1" in n]
# a step in the timed region: between the 5th and 6th renderC launch
a, b = idx[5], idx[6]
prev_end = int(rows[a - 1]["End_Timestamp"])
tot_k = tot_gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void at::native::", "at::")[:70]
    print("gap %7.1f us  run %8.1f us  %s" % (gap / 1e3, (e - s) / 1e3, n))
    tot_k += e - s; tot_gap += max(gap, 0); prev_end = max(prev_end, e)
print("step: kernels %.3f ms, gaps %.3f ms, %d launches" % (tot_k / 1e6, tot_gap / 1e6, b - a))
PY
