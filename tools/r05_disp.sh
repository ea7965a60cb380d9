#!/bin/bash
# per-dispatch durations + a few counters of selected launches of one wf_case workload: r05_disp.sh <tag> <case> [PSDR_OPTIONS] [kernel name substrings, "|"-separated]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
export PSDR_OPTIONS=$3
for PASS in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rm -rf /tmp/dp_$N
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d /tmp/dp_$N -o p -- python $R/tools/wf_case.py $2 default 1 > /tmp/dp_$N.log 2>&1
done
python - "$2 $3" "${4:-k_vertex_rev|k_camera_rev}" <<'PY' | tee -a $O/disp.txt
import csv, glob, sys, collections
print("##", sys.argv[1])
rows = {}
for d in glob.glob("/tmp/dp_*"):
    if d.endswith(".log"): continue
    kt = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kt[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    order = sorted(kt.items(), key=lambda kv: kv[1][1])
    idx = {did: i for i, (did, _) in enumerate(order)}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            did = r["Dispatch_Id"]
            if did not in kt: continue
            n = kt[did][0]
            if not any(w in n for w in sys.argv[2].split("|")): continue
            e = rows.setdefault(idx[did], {"name": n.replace("void (anonymous namespace)::", "").split("(")[0], "us": []})
            e["us"].append(kt[did][2] / 1e3)
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for i in sorted(rows):
    e = rows[i]
    us = sorted(e["us"])[len(e["us"]) // 2]
    g = lambda k: e.get(k, float("nan"))
    cyc = us * 1e-6 * 2.4e9
    print("%3d %-28s %8.1f us | VALU %.3g issue %.3f | LDS inst %.3g idx_active/CU %.2f of kernel cycles | WAIT_ANY %.2f WAIT_INST %.2f | VMEM rd %.3g wr %.3g | HBM %.2f GB (%.0f GB/s)" % (
        i, e["name"], us, g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU") / (us * 1e-6) / 1228.8e9, g("SQ_INSTS_LDS"), g("SQ_LDS_IDX_ACTIVE") / 256 / cyc,
        g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"),
        (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9, (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9 / (us * 1e-6)))
PY
