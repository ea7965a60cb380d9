#!/bin/bash
# Round-2 evidence batch (GPU box): rocprofv3 kernel stats + PMC passes of bench.py, perf cases, projection probes.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r02
bash tools/profile_bench.sh r02_final > gpurun_out/r02/profile_bench.log 2>&1
python tools/perf_cases.py c2 c3 c4 c5 open 2>&1 | grep -v amdgpu > gpurun_out/r02/perf_cases.txt
python tools/rev_cases.py 2>&1 | grep -v amdgpu >> gpurun_out/r02/perf_cases.txt
python tools/wf_binned_probe.py c3 c4 c5 2>&1 | grep -v amdgpu >> gpurun_out/r02/perf_cases.txt
python tools/iter_breakdown.py 2>&1 | grep -v amdgpu >> gpurun_out/r02/perf_cases.txt
python tools/proj_probe.py c3_cbox_bunny 384 64 32 0,6,15,20,24,30 2>&1 | grep -v amdgpu > gpurun_out/r02/proj_probe_cbox_bunny.txt
python tools/proj_probe.py c3_bunny_light 384 64 32 0,6,15,20,24,30 2>&1 | grep -v amdgpu > gpurun_out/r02/proj_probe_bunny_light.txt
python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu | grep -v "^   [0-9 ][0-9]  hip" > gpurun_out/r02/gputests_final.log
tail -3 gpurun_out/r02/gputests_final.log
(time python bench.py) > gpurun_out/r02/bench_final.log 2>&1
tail -5 gpurun_out/r02/bench_final.log | cut -c1-600
