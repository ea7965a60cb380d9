#!/bin/bash
# round 3, batch 2: tiny-scene variants (LDS tables, no tree code) A/B against the general instances; full gpu test suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
tools/micro/bin/valu_rate > $O/valu_rate.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_tiny.json 2> $O/bench_tiny.err
PSDR_TINY_VARIANTS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_general.json 2> $O/bench_general.err
timeout 600 python tools/perf_cases.py c2 > $O/perf_c2_tiny.txt 2>&1
PSDR_TINY_VARIANTS=0 timeout 600 python tools/perf_cases.py c2 > $O/perf_c2_general.txt 2>&1
tail -3 $O/gputests.log; cat $O/valu_rate.txt
python - <<PY
import json
for n in ("tiny","general"):
    try:
        d=json.loads(open("$O/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["kernel_only"], {k:d["roofline"][k] for k in ("frac","valu_wave_insts_per_launch","wait_any_frac","wait_inst_any_frac","valu_active_frac_of_wave_cycles","algorithmic_floor_frac")}, d.get("grad_rel_l2"))
    except Exception as e: print(n, "ERR", e)
PY
paste -d'\n' $O/perf_c2_tiny.txt $O/perf_c2_general.txt | grep -v "^$" | cut -c1-120
