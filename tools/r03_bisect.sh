#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for L in "$@"; do
  echo "== $L"
  if [ $L = base ]; then unset PSDR_HIP_LIB; else export PSDR_HIP_LIB=$R/variants/lib_$L.so; fi
  timeout 600 python tools/perf_cases.py c4 c3 2>&1 | grep "renderC\|rev\|fwd"
done
