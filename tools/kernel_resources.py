#!/usr/bin/env python
"""Print VGPR / SGPR / scratch / occupancy of every kernel from a `-Rpass-analysis=kernel-resource-usage` log.
usage: hipcc <flags> -Rpass-analysis=kernel-resource-usage psdr_hip.hip -o /tmp/x.so 2> res.txt; kernel_resources.py res.txt"""
import re
import subprocess
import sys

KEYS = ["VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "SGPRs:", "LDS Size", "VGPRs Spill", "SGPRs Spill"]
cur, d = None, {}
for l in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1)
        d[cur] = {}
    for k in KEYS:
        m = re.search(re.escape(k) + r"[^\d]*(\d+)", l)
        if m and cur and "remark" in l:
            d[cur][k.strip(":")] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.split("\n")
for n, v in zip(names, d.values()):
    n = n.replace("(anonymous namespace)::", "").replace("psdr::", "").split("(")[0][:64]
    print("%-66s" % n, " ".join("%s=%d" % kv for kv in v.items()))
