#!/bin/bash
# Round-2 evidence batch, second half of the round (GPU box): rocprofv3 kernel stats + PMC passes of bench.py, perf cases, probes.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r02b
bash tools/profile_bench.sh r02b_final > gpurun_out/r02b/profile_bench.log 2>&1
python tools/perf_cases.py c2 c3 c4 c5 open 2>&1 | grep -v amdgpu > gpurun_out/r02b/perf_cases.txt
python tools/rev_cases.py 2>&1 | grep -v amdgpu >> gpurun_out/r02b/perf_cases.txt
python tools/rev_split_probe.py 2>&1 | grep -v amdgpu >> gpurun_out/r02b/perf_cases.txt
python tools/wf_binned_probe.py c3 c4 c5 2>&1 | grep -v amdgpu >> gpurun_out/r02b/perf_cases.txt
python tools/bvh_build_probe.py 2>&1 | grep -v amdgpu >> gpurun_out/r02b/perf_cases.txt
python tools/iter_breakdown.py 2>&1 | grep -v amdgpu >> gpurun_out/r02b/perf_cases.txt
python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu | grep -v "^   [0-9 ][0-9]  hip" > gpurun_out/r02b/gputests_final.log
tail -3 gpurun_out/r02b/gputests_final.log
(time python bench.py) > gpurun_out/r02b/bench_final.log 2>&1
tail -5 gpurun_out/r02b/bench_final.log | cut -c1-900
