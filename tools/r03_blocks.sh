#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
for B in 10 16 24 32 40 64; do
  echo "== PSDR_BLOCKS_PER_CU=$B" >> $O/blocks.txt
  PSDR_BLOCKS_PER_CU=$B timeout 300 python tools/perf_cases.py c2 2>&1 | grep "path3     renderC\|direct11  renderC\|path3     renderD fwd K=3\|path3     renderD rev" >> $O/blocks.txt
done
cat $O/blocks.txt
