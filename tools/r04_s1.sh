#!/bin/bash
# round 4, session 1: the traced wavefront (dense trace kernel) against the class-binned streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04s1}; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wavefront or trace_matches" > $O/parity.log 2>&1; echo "pytest rc=$?" >> $O/parity.log; tail -5 $O/parity.log
for c in c4 c5 c3b; do
  for m in fused wavefront default; do
    PSDR_OPTIONS=wf_traced=0 timeout 300 python tools/wf_case.py $c $m 3 2>&1 | tail -1 | sed 's/^/traced=0 /'
    timeout 300 python tools/wf_case.py $c $m 3 2>&1 | tail -1 | sed 's/^/traced=1 /'
  done
done | tee $O/cases.txt
for c in c4 c5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $R/tools/wf_case.py $c wavefront 3 > $O/prof_$c.log 2>&1)
  f=$(ls $O/prof_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
done
