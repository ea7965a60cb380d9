#!/bin/bash
# bench.py kernel times under an environment: usage tools/r03_b.sh "ENV=1 ENV2=2" [more env sets ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for E in "$@"; do
  echo "== $E"
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d['kernel_only'].items() if k.endswith('_ms')}, d.get('grad_rel_l2'))"
done
