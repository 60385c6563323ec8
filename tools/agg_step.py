#!/usr/bin/env python
"""Aggregate a tools/profile_step.py table by kernel kind.  usage: agg_step.py file [rows...]"""
import re, sys
from collections import defaultdict
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+(\S+)\s+(.*?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", l)
    if m:
        rows.append((int(m[1]), m[2], m[3], float(m[4]), float(m[5]), float(m[6]), float(m[7]), int(m[8])))
agg = defaultdict(lambda: [0, 0, 0])
for r in rows:
    a = agg[r[1]]; a[0] += r[3]; a[1] += r[4]; a[2] += 1
print("  ".join(f"{k}={a[0]:.0f}us" for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])), " total=%.0fus" % sum(a[0] for a in agg.values()))
for i in map(int, sys.argv[2:]):
    r = rows[i]; print("   ", r[0], r[1], r[2][:44], r[3], "us", r[5], "TF/s", r[7], "GB/s")
