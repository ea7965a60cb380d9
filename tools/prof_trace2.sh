# PMC counters of k_trace (incoherent bounce rays) on one scene: where a tree walk spends its time (developer tool, via gpurun)
# usage: tools/prof_trace2.sh <scene>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
SCENE=${1:-cbox_bunny}
cd /tmp
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  N=$(echo $PASS | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $PASS --output-format csv -d /tmp/pt_$N -o p -- python $R/tools/trace_rate.py $SCENE > /tmp/pt_$N.log 2>&1
  F=$(find /tmp/pt_$N -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "pass $N: no output"; tail -3 /tmp/pt_$N.log; continue; fi
  python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "k_trace" in r.get("Kernel_Name", ""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("%-30s n=%d camera=%.6g bounce=%.6g" % (c, len(v), v[0], v[-1]))
PY
done
grep Grays /tmp/pt_SQ_WAVES.log
