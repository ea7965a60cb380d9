#!/bin/bash
# round 4: the product library against variants/lib_old.so (the library before a change) inside one box: kernel times of the tree workloads
# (forward c4 / c5 / c3b, PathTracer reverse c4pr / c5pr), perf_cases wall clock of the C2 lines, then the GPU suite on the product library
# usage (through gpurun): tools/r04_oldnew.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04on}; mkdir -p $O; cd $R
bash tools/r04_abk.sh ${1:-r04on} "old" "c4 c5 c3b c4pr c5pr" default
for rep in 1 2; do
  for v in base old; do
    if [ $v = base ]; then L=""; else L=$R/variants/lib_$v.so; fi
    echo "== $v (rep $rep)"; PSDR_HIP_LIB=$L timeout 600 python tools/perf_cases.py c2 2>&1 | grep "^C2" | grep -v "path6\|wavefront\|fused"
  done
done | tee $O/perf_ab.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
