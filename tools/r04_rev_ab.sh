#!/bin/bash
# round 4: A/B of variant libraries on the C2 reverse kernels inside one box: the reverse-mode GPU tests with each library, then perf_cases c2 (rev lines) interleaved
# usage (through gpurun): [CASES="c2 c4 c5"] tools/r04_rev_ab.sh <tag> "<variant names>" [notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04rev}; mkdir -p $O; cd $R
if [ "$3" != notests ]; then
  for v in $2; do
    PSDR_HIP_LIB=$R/variants/lib_$v.so timeout 1200 python -m pytest tests -m gpu -x -q -k "reverse or projection or inverse or rough_rev or full_size or split or config5 or tables_native" > $O/gputests_$v.log 2>&1; echo "pytest rc=$?" >> $O/gputests_$v.log
    echo "== tests $v"; tail -3 $O/gputests_$v.log
  done
fi
for rep in 1 2; do
  for v in base $2; do
    if [ $v = base ]; then L=""; else L=$R/variants/lib_$v.so; fi
    echo "== $v (rep $rep)"; PSDR_HIP_LIB=$L timeout 600 python tools/perf_cases.py ${CASES:-c2} 2>&1 | grep "^C[0-9]" | grep "rev\|K=1 geo"
  done
done | tee $O/perf_ab.txt
