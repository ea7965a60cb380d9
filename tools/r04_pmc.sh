#!/bin/bash
# PMC rows of one tools/wf_case.py workload (one rocprofv3 pass per counter group + --kernel-trace): r04_pmc.sh <tag> <case> <mode>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
NAME=$2_$3
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rm -rf /tmp/pc_${NAME}_$N
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d /tmp/pc_${NAME}_$N -o p -- python $R/tools/wf_case.py $2 $3 2 > /tmp/pc_${NAME}_$N.log 2>&1
done
python $R/tools/summarize_pmc_case.py $NAME /tmp/pc_${NAME}_ | cut -c1-700 | tee -a $O/pmc_rows.txt
