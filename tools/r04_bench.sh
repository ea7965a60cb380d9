#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04bench}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_rccl_single_rank_gpu.py -x -q -s > $O/rccl.log 2>&1; echo "rc=$?" >> $O/rccl.log; tail -6 $O/rccl.log
(time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err) 2>&1 | grep real; tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_only"]["render_c_ms"], d["kernel_only"]["render_d_fwd_k1_ms"], d["roofline"]["frac"], d["cpu_baseline"], d["grad_rel_l2"]["rel_l2"])
print(json.dumps(d["tree_scenes"], indent=1))
PY
