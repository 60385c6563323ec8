#!/usr/bin/env python
"""rocprofv3 --pmc SQ_* counter_collection.csv -> per-kernel per-dispatch averages (profiles/<tag>_pmc_sq.txt).
    python tools/pmc_sq_summary.py gpurun_out/r01n/pmc_sq/s_counter_collection.csv r01n > profiles/r01n_pmc_sq.txt"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"])).replace("kvq::", "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
tag = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"# rocprofv3 --pmc SQ_* pass (separate from the trace passes), `python bench.py --streams 1 --steps 2 --warmup 1`, {tag} build, "
      "per-dispatch averages.")
print("# SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 MFMA, 16 per 16x16x32); SQ_WAVE_CYCLES / SQ_WAIT_* count "
      "quad-cycles per wave.")
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
print(f"{'kernel':72} {'n':>4} " + " ".join(f"{c[3:][:12]:>12}" for c in cols) + "  wait/wave")
rows = []
for k, d in acc.items():
    if not k.startswith(("gemm_kernel", "window_attention", "layernorm", "block_tail", "patch_embed", "vqa_head", "conv_stem", "splitk",
                         "pool_nd", "mean_std", "fragment_gather", "pack_", "gemm8p", "fast_bottleneck", "select_frames")):
        continue
    avg = {c: (sum(d[c]) / len(d[c]) if d.get(c) else 0.0) for c in cols}
    n = len(d.get(cols[2], []))
    rows.append((avg["SQ_WAVE_CYCLES"] * n, k, n, avg))
for _, k, n, avg in sorted(rows, reverse=True):
    w = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"] if avg["SQ_WAVE_CYCLES"] else 0.0
    print(f"{k[:72]:72} {n:4d} " + " ".join(f"{avg[c]:12.0f}" for c in cols) + f"  {w:.2f}")
