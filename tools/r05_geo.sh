#!/bin/bash
# round 5: geometry duals through the traced wavefront -- parity tests + timing (perf_cases c4 / c5 print fused vs wavefront rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05g}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_config5_interior.py -m gpu -x -q -s -k "geometry_duals or config5_path_tracer" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -v "^$" $O/tests.log | tail -30
timeout 900 python tools/perf_cases.py c4 c5 skipmain 2>&1 | grep -i "fwd\|renderC" | tee $O/perf.txt
