#!/bin/bash
# round 4: A/B of handle options (PSDR_OPTIONS="name=value,...") on the C2 kernels inside one box
# usage (through gpurun): tools/r04_opt_ab.sh <tag> "<options A>" "<options B>" ...   ("-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04ab}; mkdir -p $O; cd $R; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then opt=""; else opt="$v"; fi
    echo "== options '$opt' (rep $rep)"; PSDR_OPTIONS=$opt timeout 600 python tools/perf_cases.py c2 2>&1 | grep "^C2" | grep -v "path6\|wavefront"
  done
done | tee $O/perf_ab.txt
