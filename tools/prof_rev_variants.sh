# PMC counters of the C2 PathTracer(3) all-gradients reverse kernel for several builds of the library (developer tool):
#   bash tools/prof_rev_variants.sh <lib.so> [<lib.so> ...]      ("default" = the in-tree library)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_rev_variants; mkdir -p $OUT
cd /tmp
for LIB in "$@"; do
  TAG=$(basename $LIB .so)
  if [ "$LIB" != "default" ]; then export PSDR_HIP_LIB=$LIB; else unset PSDR_HIP_LIB; fi
  for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
    N=$(echo $PASS | cut -d' ' -f1)
    rm -rf /tmp/prv_$N
    timeout 150 rocprofv3 --pmc $PASS --output-format csv -d /tmp/prv_$N -o p -- python $R/tools/prof_case.py cbox path rev 512 64 3 > $OUT/log_${TAG}_$N.txt 2>&1
    F=$(find /tmp/prv_$N -name "*counter_collection.csv" | head -1)
    python - "$F" "$TAG" <<'PY' >> $OUT/pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    if "k_camera_rev" not in r.get("Kernel_Name", "?"): continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(sys.argv[2], " ".join("%s=%.4g" % (c, v / cnt[c]) for c, v in sorted(agg.items())))
PY
  done
done
cat $OUT/pmc.txt
