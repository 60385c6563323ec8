#!/usr/bin/env python
"""Per-launch durations (us) of the attention kernels of one traced run, in launch order, folded over the steps:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --streams 1 --legs c2 ... ; python tools/attn_launches.py DIR [launches per step]"""
import csv, glob, os, sys
d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "window_attention" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
names = [("a32" if "attention32" in r["Kernel_Name"] else "gather") for r in rows]
n = len(dur) // per
dur, names = dur[-per * (n - 1):], names[-per * (n - 1):]        # drop the first (warm-up) step
steps = len(dur) // per
avg = [sum(dur[i + s * per] for s in range(steps)) / steps for i in range(per)]
print(" ".join(f"{names[i][0]}{avg[i]:.1f}" for i in range(per)), "| sum %.1f us over %d steps" % (sum(avg), steps))
