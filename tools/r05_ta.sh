#!/bin/bash
# round 5: is the bounce stage bound by the texture-address / L1 pipeline (gathers of whole rows, one cache line per lane and instruction)?
# r05_ta.sh <tag> <case> <mode>: TA / TCP / TD counters of one tools/wf_case.py workload, one rocprofv3 pass per group
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
NAME=$2_$3
for PASS in "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rm -rf /tmp/ta_${NAME}_$N
  timeout 600 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d /tmp/ta_${NAME}_$N -o p -- python $R/tools/wf_case.py $2 $3 1 > /tmp/ta_${NAME}_$N.log 2>&1 || tail -3 /tmp/ta_${NAME}_$N.log
done
python - <<PY | tee -a $O/ta_rows.txt
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); nl = collections.defaultdict(int)
for d in glob.glob("/tmp/ta_${NAME}_*"):
    if d.endswith(".log"): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_" in k and "native" not in k:
                cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    first = not dur
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        if not first: break
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_" in k and "native" not in k:
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); nl[k] += 1
print("## ${NAME}: counters summed over all launches of the workload's calls (wf_case.py runs the call twice)")
for k, c in cnt.items():
    short = k.replace("void (anonymous namespace)::", "").replace("psdr::", "").split("(")[0][:60] or "k_trace"
    print("%-60s launches %d total %.3f ms | " % (short, nl[k], dur[k] * 1e-6) + " ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY
