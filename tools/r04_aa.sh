#!/bin/bash
# round 4: A/B of library variants (variants/lib_<name>.so, tools/build_variant_lib.sh) on the C2 kernels inside one box, plus the GPU suite
# usage (through gpurun): tools/r04_aa.sh <tag> "<variant names>" [notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04aa}; mkdir -p $O; cd $R
if [ "$3" != notests ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
  tail -4 $O/gputests.log
fi
for rep in 1 2; do
  for v in base $2; do
    if [ $v = base ]; then L=""; else L=$R/variants/lib_$v.so; fi
    echo "== $v (rep $rep)"; PSDR_HIP_LIB=$L timeout 600 python tools/perf_cases.py c2 2>&1 | grep "^C2" | grep -v "path6\|wavefront"
  done
done | tee $O/perf_ab.txt
