#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/pmc_traffic.json: per kernel symbol the average
HBM bytes per launch, corrected as MI355X_MICROARCH.md §HBM prescribes (gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide coalesced stream -> x2; WRITE_SIZE as reported; both in KB).
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv"""
import collections
import csv
import json
import re
import sys


def sym(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("kvq::window_attention", "window_attention").replace("kvq::gemm_kernel", "gemm_kernel") \
               .replace("kvq::layernorm_rows_kernel", "layernorm_rows_kernel").replace("kvq::patch_im2col_kernel", "patch_im2col_kernel") \
               .replace("kvq::block_tail_kernel", "block_tail_kernel").replace("kvq::block_tail16_kernel", "block_tail16_kernel") \
               .replace("kvq::block_tailmm_kernel", "block_tailmm_kernel") \
               .replace("kvq::patch_embed_kernel", "patch_embed_kernel")


def agg(path, counter):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a[sym(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return a


f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {}
for k in f:
    if not k.startswith(("gemm_kernel", "window_attention", "layernorm", "patch_im2col", "block_tail", "patch_embed")):
        continue
    fb = 2.0 * 1024.0 * sum(f[k]) / len(f[k])
    wb = 1024.0 * sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    # the bench symbol drops the FULL template flag of the attention kernel and LN's template arguments
    key = re.sub(r"(window_attention_kernel<kvq::\w+, \w+, \w+), \w+>", r"\1>", k)
    key = re.sub(r"layernorm_rows_kernel<.*>", "layernorm_rows_kernel", key)
    key = re.sub(r"patch_im2col_kernel<.*>", "patch_im2col_kernel", key)
    e = out.setdefault(key, dict(fetch_bytes=0.0, write_bytes=0.0, launches=0))
    n = len(f[k])
    e["fetch_bytes"] = (e["fetch_bytes"] * e["launches"] + fb * n) / (e["launches"] + n)
    e["write_bytes"] = (e["write_bytes"] * e["launches"] + wb * n) / (e["launches"] + n)
    e["launches"] += n
json.dump({"build": (sys.argv[3] if len(sys.argv) > 3 else "r01"), "note": "avg HBM bytes per launch over one B=4 fp16 step mix; FETCH_SIZE x2 (gfx950 correction), "
                   "separate --pmc passes (" + (sys.argv[3] if len(sys.argv) > 3 else "r01") + " build, bench.py --streams 1)", "kernels": out}, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:600])
