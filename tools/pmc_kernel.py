#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.  usage: pmc_kernel.py filter file.csv..."""
import collections, csv, sys
flt = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32} {sum(v) / len(v):16.0f}   (n={len(v)})")
