#!/bin/bash
# round 6: chosen GPU tests (or the whole suite: TESTS=all) + the default bench line in ONE call.  usage (through gpurun): tools/r06_quick.sh <tag> ["<pytest args>"]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06q}; mkdir -p $O; cd $R
T=${2:-tests -m gpu}
timeout 2400 python -m pytest $T -x -q -s > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
grep -v "^$" $O/gputests.log | tail -${TAILN:-12} | cut -c1-400
if [ "${NOBENCH:-0}" != 1 ]; then
timeout 1200 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: d["kernel_only"][k] for k in d["kernel_only"] if k.endswith("_ms")})
r=d["roofline"]; print({k: r.get(k) for k in ("kernel_ms","frac","algorithmic_floor_frac","floor_time_frac","traffic","wait_any_frac","valu_wave_insts_per_launch")}, (d.get("grad_rel_l2") or {}).get("rel_l2"))
for k,v in (d.get("tree_scenes") or {}).items():
    if isinstance(v, dict) and "ms" in v: print(k, v["ms"], "ms", (v.get("dominant_kernel") or {}).get("name"), (v.get("dominant_kernel") or {}).get("ms_under_profiler"), (v.get("dominant_kernel") or {}).get("valu_issue_frac"))
for k in ("c4_strong","c4_strong_one_integrator"):
    s=d.get(k) or {}; print(k, s.get("ms_per_step"), s.get("error"))
PY
fi
