#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``rocprofv3 --kernel-trace --stats``) into the per-kernel
table committed under profiles/: calls, total / average / min / max duration, share of GPU time.
    python tools/rocprof_summary.py gpurun_out/prof_r01a/r01a_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':90} {'calls':>6} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for n, c, t, a, mn, mx in rows:
        n = n.replace("void kvq::", "").replace("kvq::", "")
        print(f"{n[:90]:90} {c:6d} {t / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * t / total:6.2f}")
    print(f"{'TOTAL':90} {sum(r[1] for r in rows):6d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
