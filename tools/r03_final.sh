#!/bin/bash
# round 3 final measurements: bench (N=1, c4), kernel-trace + PMC profile of the bench, tree-kernel PMC rows
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_final; mkdir -p $O; cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 1200 python bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
PSDR_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > $O/bench_n2_onegpu.json 2> $O/bench_n2.err
timeout 900 tools/profile_bench.sh r03_final > $O/profile.log 2>&1
timeout 600 python tools/perf_cases.py c2 c3 c4 c5 open > $O/perf_cases.txt 2>&1
timeout 300 python tools/iter_sync_probe.py > $O/iter_probe.txt 2>&1
timeout 300 python tools/iter_breakdown.py >> $O/iter_probe.txt 2>&1
cut -c1-2500 $O/bench_n1.json; echo; cut -c1-3000 $O/bench_c4.json; echo; cat $O/perf_cases.txt; tail -5 $O/iter_probe.txt
