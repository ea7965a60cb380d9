#!/bin/bash
# A/B of library variants on the headline kernels (bench.py kernel_only, interleaved): r04_abbench.sh "<names>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2 3; do
  for v in base $1; do
    if [ $v = base ]; then L=""; else L=$R/variants/lib_$v.so; fi
    timeout 600 python bench.py ${L:+--hip-lib $L} --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-tree-scenes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_only']
print('$v', 'value %.0f' % d['value'], 'c %.4f k1 %.4f k3 %.4f rev %.4f rev_all %.4f' % (k['render_c_ms'], k['render_d_fwd_k1_ms'], k['render_d_fwd_k3_ms'], k['render_d_rev_ms'], k['render_d_rev_all_ms']))"
  done
done
