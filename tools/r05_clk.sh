#!/bin/bash
# round 5: (1) where a wave of the traced bounce stage spends its wall time (variants/lib_clk.so = -DPSDR_STAGE_CLOCKS, tools/build_variant_lib.sh);
# (2) run-to-run spread of the PathTracer vertex gradient in ONE process (the tolerance of tests/test_multi_gpu_exec_gpu.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05clk}; mkdir -p $O; cd $R
PSDR_HIP_LIB=$R/variants/lib_clk.so timeout 600 python tools/wf_case.py c4 wavefront 2 2>&1 | grep -v Warning | tail -4 | tee $O/clk.txt
python tools/wf_case.py c4 wavefront 2 2>&1 | tail -1 | tee -a $O/clk.txt
PSDR_HIP_LIB=$R/variants/lib_clk.so timeout 600 python tools/wf_case.py c4pr default 2 2>&1 | grep -v Warning | tail -4 | tee -a $O/clk.txt
python tools/wf_case.py c4pr default 2 2>&1 | tail -1 | tee -a $O/clk.txt
timeout 900 python - <<'PY' 2>&1 | tail -8 | tee $O/noise.txt
import os, sys
sys.path.insert(0, "psdr-cuda_amd"); sys.path.insert(0, "tests")
import numpy as np
from test_multi_gpu_exec_gpu import run_sequence
from helpers import rel_l2
a = run_sequence(); b = run_sequence()
for k in ("pt_g_vert", "pt_fwd_grad", "g_vert", "pt_rev_img"):
    d = np.abs(a[k] - b[k]); i = np.unravel_index(np.argmax(d), d.shape)
    print(k, "rel_l2 between two runs of one process %.3e" % rel_l2(a[k], b[k]), "max |diff| %.3e at %s value %.4e, |g|max %.3e" % (d[i], i, b[k][i], np.abs(b[k]).max()))
PY
