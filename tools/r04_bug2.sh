#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export PSDR_HIP_LIB=$R/variants/lib_${1:-dppall}.so
for h in none bvh big; do python tools/rough_rev_repro.py $h 2>&1 | grep "history\|cam grad\|Error"; done
for h in none bvh; do for pat in 0 7fc00000 42f60000; do for w in 1 2 4 7; do python tools/rough_rev_repro.py $h $pat $w 2>&1 | grep "history\|Error"; done; done; done
