#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
tools/micro/bin/gather_rate > $O/gather_rate.txt 2>&1
cat $O/gather_rate.txt
tools/r03_tree_prof.sh r03g > /dev/null 2>&1
grep -v "^$" $O/tree_kernels.txt | cut -c1-420
