#!/usr/bin/env python
"""How the stream lanes of the bench share the chip: from a rocprofv3 --kernel-trace csv of a multi-stream run, per kernel family the
time it spends running under the 4-lane mix against its one-stream duration, the concurrency histogram (how many kernels are in flight),
and the wall time covered by 0 / 1 / 2 / ... concurrent kernels.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --legs c2 --no-cpu-baseline --no-pmc --steps 20 --min-timed-s 0
    python tools/lane_overlap.py DIR"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# keep the last 60 % of the trace (steady state of the timed region)
t_lo = rows[0][0] + 0.4 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
def fam(n):
    n = re.sub(r"^void ", "", n).replace("kvq::", "")
    m = re.match(r"([a-z0-9_]+)(<[^>]*>)?", n)
    base = m.group(1) if m else n
    if "tailmm" in base: return "tailmm"
    if "block_tail" in base: return "tail" + ("96" if "<Fp16, 3" in n or "Fp16, 3," in n else "192" if "Fp16, 6" in n else "")
    if "attention32" in base: return "attn32"
    if "gemm8p" in base: return "gemm8p"
    if "gemm_kernel" in base: return "gemm"
    return base
ev = []
for s, e, n, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = defaultdict(int); cur = 0; last = ev[0][0]
for t, dlt in ev:
    hist[cur] += t - last; last = t; cur += dlt
tot = sum(hist.values())
print("kernels in flight: share of wall time")
for k in sorted(hist): print(f"  {k}: {100 * hist[k] / tot:5.1f} %")
byf = defaultdict(list)
for s, e, n, q in rows: byf[fam(n)].append((e - s) / 1e3)
print(f"{'family':28} {'calls':>6} {'avg us':>9} {'sum ms':>9}")
for k, v in sorted(byf.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28} {len(v):6d} {sum(v) / len(v):9.1f} {sum(v) / 1e3:9.3f}")
print(f"wall {tot / 1e6:.3f} ms, sum of kernel durations {sum(sum(v) for v in byf.values()) / 1e3:.3f} ms, queues {len(set(r[3] for r in rows))}")
