#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r03j}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q tests/test_device_bvh_gpu.py tests/test_edge_cases_gpu.py tests/test_gpu_parity.py > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
timeout 900 python tools/perf_cases.py c4 c5 skipmain > $O/perf.txt 2>&1
cat $O/perf.txt
cd /tmp; export TMPDIR=/tmp
for C in "cbox_bunny direct trace 1024 4" "interior direct trace 512 16"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$$ -o kt -- python $R/tools/prof_case.py $C > /dev/null 2>&1
  grep "k_trace" $(find /tmp/kt_$$ -name "*kernel_stats.csv") | cut -d, -f1-4 | cut -c1-60,200-; rm -rf /tmp/kt_$$
done
