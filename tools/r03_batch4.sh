#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r03h}; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout 900 python tools/perf_cases.py c4 c5 c3 open > $O/perf.txt 2>&1
cat $O/perf.txt
