#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r03q}; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["kernel_only"], {k:d["roofline"][k] for k in ("frac","valu_wave_insts_per_launch","wait_any_frac","algorithmic_floor_frac")}, d.get("grad_rel_l2"))
PY
# the tree workloads too (a change aimed at C2 once doubled C4's renderC unnoticed)
timeout 900 python tools/perf_cases.py c4 c5 c3 2>&1 | grep "renderC\|rev\|fwd" > $O/perf_tree.txt; cat $O/perf_tree.txt
