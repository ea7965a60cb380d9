export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_rev; mkdir -p $OUT
cd /tmp
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INSTS_GDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PASS | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $PASS --output-format csv -d /tmp/pr_$N -o p -- python $R/tools/prof_case.py cbox path rev 512 64 3 > $OUT/log_$N.txt 2>&1
  F=$(find /tmp/pr_$N -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY' >> $OUT/pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "?")
    if "k_camera_rev" not in k: continue
    agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k[:60], r["Counter_Name"])] += 1
for k in agg:
    for c, v in agg[k].items():
        print("%-60s %-22s n=%d per_dispatch=%.6g" % (k, c, cnt[(k, c)], v / cnt[(k, c)]))
PY
done
cat $OUT/pmc.txt
