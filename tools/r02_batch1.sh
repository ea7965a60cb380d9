cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -8
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r02/bench_a.log 2>&1; tail -4 gpurun_out/r02/bench_a.log
python tools/proj_probe.py c3_cbox_bunny 384 64 32 0,6,15,20,24,30 2>&1 | grep -v amdgpu > gpurun_out/r02/proj_probe_cbox_bunny.txt
python tools/proj_probe.py c3_bunny_light 384 64 32 0,6,15,20,24,30 2>&1 | grep -v amdgpu > gpurun_out/r02/proj_probe_bunny_light.txt
tail -9 gpurun_out/r02/proj_probe_cbox_bunny.txt
