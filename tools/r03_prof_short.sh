#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03_short}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
run_case() {
  NAME=$1; ENVV=$2; shift 2
  for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $PASS | cut -d' ' -f1)
    rm -rf /tmp/pc_${NAME}_$N
    env $ENVV rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d /tmp/pc_${NAME}_$N -o p -- python $R/tools/prof_case.py "$@" > /tmp/pc_${NAME}_$N.log 2>&1
  done
  python $R/tools/summarize_pmc_case.py $NAME /tmp/pc_${NAME}_ >> $O/tree_kernels.txt 2>&1
}
run_case c4_wavefront PSDR_PROF_WAVEFRONT=1 cbox_bunny path c 1024 32 3
run_case c4_fused PSDR_PROF_WAVEFRONT=0 cbox_bunny path c 1024 32 3
run_case c5_fused PSDR_PROF_WAVEFRONT=0 interior path c 512 16 3
run_case c4_trace X=1 cbox_bunny direct trace 1024 4
run_case c5_trace X=1 interior direct trace 512 16
cat $O/tree_kernels.txt | cut -c1-420
