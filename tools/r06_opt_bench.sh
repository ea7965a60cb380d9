#!/bin/bash
# round 6: the headline's kernel-only block under several handle options in ONE call.  usage: tools/r06_opt_bench.sh <tag> "<name=value or ->" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
for rep in 1 2; do i=0
for opt in "$@"; do i=$((i+1))
  if [ "$opt" = "-" ]; then arg=""; else arg="--native-option $opt"; fi
  python bench.py --no-tree-scenes --no-c4-strong --no-cpu-baseline --no-pmc --steps 20 --warmup 5 $arg > $O/$i.$rep.json 2> $O/$i.$rep.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open("$O/$i.$rep.json").read().strip().splitlines()[-1]); k=d["kernel_only"]
print("%-20s rep $rep value %.0f  renderC %.4f  renderD k1 %.4f k3 %.4f  rev %.3f rev_all %.3f" % ("$opt", d["value"], k["render_c_ms"], k["render_d_fwd_k1_ms"], k["render_d_fwd_k3_ms"], k["render_d_rev_ms"], k["render_d_rev_all_ms"]))
PY
done; done
