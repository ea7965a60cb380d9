#!/bin/bash
# A/B of library variants (variants/lib_<name>.so) on the wf_case workloads, interleaved: r04_ab.sh <tag> "<names>" "<cases>" [mode]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
MODE=${4:-wavefront}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wavefront_and_fused or trace_matches" 2>&1 | tail -2
for rep in 1 2; do
for c in $3; do
  for v in base $2; do
    if [ $v = base ]; then L=""; else L=$R/variants/lib_$v.so; fi
    PSDR_HIP_LIB=$L timeout 300 python tools/wf_case.py $c $MODE 5 2>&1 | tail -1 | sed "s/^/$v /"
  done
done
done | tee $O/ab.txt
