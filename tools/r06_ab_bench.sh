#!/bin/bash
# round 6: the headline's kernel-only block + counters for several builds of the library inside ONE call.  usage: tools/r06_ab_bench.sh <tag> <lib or "product"> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = product ]; then arg=""; else arg="--hip-lib $R/variants/lib_$lib.so"; fi
  python bench.py --no-tree-scenes --no-c4-strong --no-cpu-baseline --steps 20 --warmup 5 $arg > $O/$lib.$rep.json 2> $O/$lib.$rep.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open("$O/$lib.$rep.json").read().strip().splitlines()[-1])
k=d["kernel_only"]; r=d["roofline"]
print("%-10s rep $rep value %.0f  renderC %.4f  renderD k1 %.4f k3 %.4f  rev %.3f rev_all %.3f | dom VALU %.1fM traffic %.0f MB; renderC VALU %.1fM traffic %.0f MB" % ("$lib", d["value"], k["render_c_ms"], k["render_d_fwd_k1_ms"], k["render_d_fwd_k3_ms"], k["render_d_rev_ms"], k["render_d_rev_all_ms"],
      (r["valu_wave_insts_per_launch"] or 0)/1e6, (r["traffic"] or 0)/1e6, (r["other_kernels"].get("c",{}).get("valu_wave_insts") or 0)/1e6, (r["other_kernels"].get("c",{}).get("hbm_bytes") or 0)/1e6))
PY
done; done
