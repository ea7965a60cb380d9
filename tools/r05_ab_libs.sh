#!/bin/bash
# A/B of variant libraries (tools/build_variant_lib.sh) on one tools/wf_case.py workload: r05_ab_libs.sh <tag> <case> <mode> <lib names ...>  ("product" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
CASE=$2; MODE=$3; shift 3
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "product" ]; then unset PSDR_HIP_LIB; else export PSDR_HIP_LIB=$R/variants/lib_$L.so; fi
  timeout 600 python tools/wf_case.py $CASE $MODE 5 2>&1 | tail -n 1 | sed "s/^/$L /" | tee -a $O/ab_$CASE.txt
done; done
