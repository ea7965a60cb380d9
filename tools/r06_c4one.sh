#!/bin/bash
# round 6: c4_strong / c4_strong_one_integrator of several builds in one call.  usage: tools/r06_c4one.sh <tag> <lib or product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
for lib in "$@"; do
  if [ "$lib" = product ]; then arg=""; else arg="--hip-lib $R/variants/lib_$lib.so"; fi
  python bench.py --no-tree-scenes --no-cpu-baseline --no-pmc --steps 5 --warmup 2 $arg > $O/$lib.json 2> $O/$lib.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open("$O/$lib.json").read().strip().splitlines()[-1])
print("%-8s c4_strong %.1f  one_integrator %.1f" % ("$lib", d["c4_strong"]["ms_per_step"], d["c4_strong_one_integrator"]["ms_per_step"]))
PY
done
