#!/bin/bash
# round 4: per-dispatch durations of the traced-wavefront kernels on the C4 shard (through gpurun): profiles/r04_stage_dispatch_times.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt1; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o p -- python $R/tools/wf_case.py c4 default 3 > /tmp/kt1.log 2>&1
f=$(find /tmp/kt1 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "k_wf_" in r["Kernel_Name"]]
for r in sel[-16:]:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    print("%-44s %8.1f us  grid %s" % (n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
PY
