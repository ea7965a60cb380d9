#!/bin/bash
# round 5: the whole GPU suite + the forward-geometry / guiding timings in ONE call
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05q}; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -5 $O/gputests.log
timeout 900 python tools/perf_cases.py c4 c5 skipmain 2>&1 | grep -i "fwd\|renderC\|rev" | tee $O/perf.txt
timeout 600 python tools/perf_cases.py c2 2>&1 | grep -i "geo\|rev all\|renderC " | tee -a $O/perf.txt
for opt in "probe=1" "probe=0"; do PSDR_OPTIONS=$opt timeout 600 python tools/guide_case.py 32 3 2>&1 | tail -1 | sed "s/^/$opt /" | tee -a $O/perf.txt; done
