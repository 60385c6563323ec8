#!/usr/bin/env python
"""Timeline statistics of the 4-lane replayed C2 steps from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --probe c2mix --probe-steps 24 --streams 4
    python tools/mix_timeline.py DIR [steps]
Prints, for the traced steps (behind the probe's 60 ms idle gap): per queue the busy time and the gaps between consecutive kernels,
the time-weighted histogram of concurrently running kernels, and the per-family launch durations."""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for x in csv.DictReader(open(f)):
    rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"], x.get("Queue_Id", "?"), x.get("Stream_Id", "?")))
rows.sort()
cut, last_end = 0, rows[0][1]
for i, (st, en, *_) in enumerate(rows):
    if st - last_end >= 40_000_000:
        cut = i
    last_end = max(last_end, en)
rows = [r for r in rows[cut:] if not any(k in r[2] for k in bench.ONE_TIME_KERNELS)]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = (t1 - t0) / 1e3
print(f"{len(rows)} launches over {span:.1f} us = {span / K:.1f} us per step ({K} steps), {len(rows) / K:.1f} launches per step")
# per queue
byq = collections.defaultdict(list)
for r in rows:
    byq[(r[3], r[4])].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(en - st for st, en, *_ in rs) / 1e3
    gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1)]
    gaps_pos = sorted(g for g in gaps)
    n = len(gaps_pos)
    print(f"queue {q}: {len(rs)} launches, busy {busy:.0f} us = {busy / span:.2f} of the span; gap between consecutive kernels: median "
          f"{gaps_pos[n // 2]:.1f} us, p10 {gaps_pos[n // 10]:.1f}, p90 {gaps_pos[9 * n // 10]:.1f}, mean {sum(gaps_pos) / n:.1f}, negative (overlap) {sum(1 for g in gaps if g < 0)}")
# concurrency histogram
ev = []
for st, en, *_ in rows:
    ev.append((st, 1)); ev.append((en, -1))
ev.sort()
hist, cur, prev = collections.Counter(), 0, t0
for t, dlt in ev:
    hist[cur] += t - prev
    prev, cur = t, cur + dlt
tot = sum(hist.values())
print("kernels running at once (share of the span): " + "  ".join(f"{k}: {v / tot:.3f}" for k, v in sorted(hist.items())))
print(f"average {sum(k * v for k, v in hist.items()) / tot:.2f}")
fam = collections.defaultdict(lambda: [0, 0.0])
for st, en, name, *_ in rows:
    e = fam[bench.family_of(name)]
    e[0] += 1; e[1] += (en - st) / 1e3
for k, (n, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:14s} {n / K:5.1f} launches/step  avg {us / n:7.1f} us  sum/step {us / K:7.1f} us")
