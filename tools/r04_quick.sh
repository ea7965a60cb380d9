#!/bin/bash
# round 4: GPU tests + bench (with its tree_scenes block) + kernel times of the tree workloads in ONE call -- run it after every kernel change
# usage (through gpurun): tools/r04_quick.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04q}; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["kernel_only"], {k:d["roofline"][k] for k in ("frac","valu_wave_insts_per_launch","wait_any_frac")}, d.get("grad_rel_l2",{}).get("rel_l2"))
for k,v in (d.get("tree_scenes") or {}).items():
    if isinstance(v, dict) and "ms" in v: print(k, v["ms"], "ms", v["Grays_per_s"], "Grays/s", (v.get("dominant_kernel") or {}).get("name"), (v.get("dominant_kernel") or {}).get("valu_issue_frac"))
PY
bash tools/r04_abk.sh ${1:-r04q} "" "c4 c5 c3b"
