#!/bin/bash
# Timeline of one surface_reverse_all_step of bench.py (configure with vertex / radiance / camera / albedo gradients + renderD + enoki.backward on the headline scene):
# every kernel with its duration and the idle gap before it (developer tool, via gpurun)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05tl}; mkdir -p $O
cd /tmp; rm -rf /tmp/rtl
cat > /tmp/rtl_drv.py <<PY
import sys, types
sys.path.insert(0, "$R")
import torch, bench
args = types.SimpleNamespace(scene="cbox", res=512, spp=64, max_depth=3)
w = bench.Workload(args, 1)
for _ in range(8):
    w.surface_reverse_all_step()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/rtl -o t -- python /tmp/rtl_drv.py > /tmp/rtl.log 2>&1 || tail -5 /tmp/rtl.log
F=$(find /tmp/rtl -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY' | tee $O/revall_timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_camera_rev<8, true, 1, 1>" in n]
a, b = idx[5], idx[6]
prev_end = int(rows[a - 1]["End_Timestamp"])
tot_k = tot_gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void at::native::", "at::")[:90]
    print("gap %7.1f us  run %8.1f us  %s" % (gap / 1e3, (e - s) / 1e3, n))
    tot_k += e - s; tot_gap += max(gap, 0); prev_end = max(prev_end, e)
print("step (from one recording primal render to the next): kernels %.3f ms, gaps %.3f ms, %d launches" % (tot_k / 1e6, tot_gap / 1e6, b - a))
PY
