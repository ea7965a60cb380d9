#!/bin/bash
# perf_cases under several builds: usage tools/r03_bisect.sh "<cases>" <lib name | base> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CASES=$1; shift
for L in "$@"; do
  echo "== $L"
  if [ $L = base ]; then unset PSDR_HIP_LIB; else export PSDR_HIP_LIB=$R/variants/lib_$L.so; fi
  timeout 600 python tools/perf_cases.py $CASES 2>&1 | grep "renderC\|rev\|fwd"
done
