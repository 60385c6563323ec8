import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "patch_embed_kernel" in k or "fragment_gather" in k:
        acc[(k[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in acc.items():
    print(f"{k}  {c}: mean {sum(v) / len(v):.0f} (counter units) over {len(v)} launches")
