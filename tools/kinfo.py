#!/usr/bin/env python3
"""Register / LDS / spill summary of every kernel in a hipcc -save-temps .s file.  usage: kinfo.py file.s [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:90]:90} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>3} spill {g('vgpr_spill_count'):>4} sgpr {g('sgpr_count'):>4} "
              f"lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}")
