# PMC counters of k_trace on cbox_bunny (developer tool, run through gpurun)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --output-format csv -d /tmp/pt_$N -o p -- python $R/tools/trace_rate.py cbox_bunny > /tmp/pt_$N.log 2>&1
  F=$(find /tmp/pt_$N -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "k_trace" in r.get("Kernel_Name", ""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("%-24s n=%d last=%.6g" % (c, len(v), v[-1]))
PY
done
