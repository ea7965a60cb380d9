#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
for L in base wf3 wf5 wf6; do
  echo "== $L" >> $O/wfwaves.txt
  if [ $L = base ]; then unset PSDR_HIP_LIB; else export PSDR_HIP_LIB=$R/variants/lib_$L.so; fi
  timeout 600 python tools/perf_cases.py c4 c5 skipmain 2>&1 | grep "C4 shard path3\|C5 path3\|C4 shard direct11 *renderC\|C5 direct11 *renderC" >> $O/wfwaves.txt
done
cat $O/wfwaves.txt
