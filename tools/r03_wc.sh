#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
for L in base wc3; do
  echo "== $L" >> $O/wc.txt
  if [ $L = base ]; then unset PSDR_HIP_LIB; else export PSDR_HIP_LIB=$R/variants/lib_$L.so; fi
  timeout 600 python tools/perf_cases.py c4 c3 2>&1 | grep "renderC\|rev" >> $O/wc.txt
  PSDR_TWO_LEVEL=0 timeout 600 python tools/perf_cases.py c4 skipmain 2>&1 | grep "renderC" | sed 's/^/one-tree /' >> $O/wc.txt
done
cat $O/wc.txt
