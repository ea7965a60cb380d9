#!/bin/bash
# round 3, first GPU batch: micro-benchmark, gpu tests, bench (N=1, self-launched N=2 on one GPU), kernel-trace + PMC profile
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R
tools/micro/bin/valu_rate > gpurun_out/r03/valu_rate.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_n1.json 2> gpurun_out/r03/bench_n1.err
PSDR_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/r03/bench_n2_onegpu.json 2> gpurun_out/r03/bench_n2.err
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 > gpurun_out/r03/bench_c4.json 2> gpurun_out/r03/bench_c4.err
timeout 900 tools/profile_bench.sh r03a > gpurun_out/r03/profile.log 2>&1
tail -3 gpurun_out/r03/gputests.log; cat gpurun_out/r03/valu_rate.txt; cut -c1-1500 gpurun_out/r03/bench_n1.json; cut -c1-600 gpurun_out/r03/bench_n2_onegpu.json; cut -c1-800 gpurun_out/r03/bench_c4.json; tail -5 gpurun_out/r03/bench_c4.err
