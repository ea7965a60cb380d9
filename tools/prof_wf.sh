#!/bin/bash
# kernel-trace stats of the wavefront / fused PathTracer on one tree scene (developer tool, via gpurun)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WHICH=${1:-c4}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pwf -o kt -- python $R/tools/wf_binned_probe.py $WHICH > /tmp/pwf.log 2>&1
F=$(find /tmp/pwf -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and (r[0] == "Name" or "k_" in r[0]):
        name = r[0].replace("void (anonymous namespace)::", "").split("(")[0][:60]
        print("%-62s %s" % (name, " ".join(r[1:5])))
PY
grep -v amdgpu /tmp/pwf.log | tail -4
