#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of bench.py, and writes
# small text summaries into gpurun_out/prof/ (copy what you want judged into profiles/).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
export TMPDIR=/tmp
mkdir -p $OUT /tmp/prof_$TAG
cd /tmp
ARGS="--no-cpu-baseline --no-pmc $*"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -o kt -- python $R/bench.py --steps 5 --warmup 2 $ARGS > $OUT/kt_bench.log 2>&1
find /tmp/prof_$TAG/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --output-format csv -d /tmp/prof_$TAG/pmc_$N -o p -- python $R/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_$N.log 2>&1
  F=$(find /tmp/prof_$TAG/pmc_$N -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then
    python - "$F" > $OUT/pmc_$N.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "?")[:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    for c, v in agg[k].items():
        n = cnt[(k, c)]
        print("%-90s %-22s dispatches=%d total=%.6g per_dispatch=%.6g" % (k, c, n, v, v / n))
PY
  fi
done
echo "== kernel stats"; cat $OUT/kernel_stats.csv 2>/dev/null | cut -c1-220 | head -12
echo "== pmc"; cat $OUT/pmc_*.txt | grep -v "at::native\|elementwise\|fill" | head -60
tail -1 $OUT/kt_bench.log | cut -c1-400
du -sh $OUT
