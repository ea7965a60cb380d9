#!/bin/bash
# Runs on the GPU box (its 256 host cores): regenerates tests/golden/proj_*.npz with the fp64 oracle at full size,
# copies them to gpurun_out/ (from where they are committed), then runs the GPU projection tests against them.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/golden
python tests/golden/make_projections.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/golden/make_projections.log
cp tests/golden/proj_*.npz gpurun_out/golden/
python -m pytest tests/test_projections_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40
