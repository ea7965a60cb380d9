#!/bin/bash
# full GPU suite + the tree perf cases: r04_full.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04full}; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -15 $O/gputests.log
