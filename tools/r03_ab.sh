#!/bin/bash
# A/B of two builds on the tree workloads: usage tools/r03_ab.sh <outdir-tag> <variant lib name>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O; cd $R
for L in base $2 base $2; do
  echo "== $L" >> $O/ab.txt
  if [ $L = base ]; then unset PSDR_HIP_LIB; else export PSDR_HIP_LIB=$R/variants/lib_$L.so; fi
  timeout 600 python tools/perf_cases.py c4 c5 c3 open 2>&1 | grep "renderC\|rev\|fwd" >> $O/ab.txt
  PSDR_OPTIONS=two_level=0 timeout 600 python tools/perf_cases.py c4 skipmain 2>&1 | grep "renderC" | sed 's/^/one-tree /' >> $O/ab.txt
done
python - <<PY
import re, collections
rows = collections.OrderedDict(); cur = None
for l in open("$O/ab.txt"):
    if l.startswith("=="): cur = l.split()[1]; continue
    m = re.match(r"(.*?)\s+([0-9.]+) ms", l)
    if m: rows.setdefault(m.group(1).strip(), collections.defaultdict(list))[cur].append(float(m.group(2)))
for k, v in rows.items():
    a, b = min(v["base"]), min(v["$2"])
    print("%-62s base %8.2f  $2 %8.2f   base/other %.3f" % (k, a, b, a / b))
PY
