# instruction-cache counters of the reverse camera kernel (developer tool, via gpurun)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for PASS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  N=$(echo $PASS | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $PASS --output-format csv -d /tmp/pi_$N -o p -- python $R/tools/prof_case.py cbox path rev 512 64 3 > /tmp/pi_$N.log 2>&1
  F=$(find /tmp/pi_$N -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "pass $N: no output"; tail -2 /tmp/pi_$N.log; continue; fi
  python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "k_camera_rev" in r.get("Kernel_Name", ""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("%-24s n=%d avg=%.6g" % (c, len(v), sum(v) / len(v)))
PY
done
