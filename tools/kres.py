"""Per-kernel register / spill / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output:
    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o x.o 2> res.txt; python tools/kres.py res.txt [substring]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
pat = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", re.S)
for m in pat.finditer(t):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = name.replace("void kvq::", "").split("(")[0]
    if sub in name:
        print(f"{name[:100]:100s} vgpr {m.group(2):>4s} agpr {m.group(3):>4s} scratch {m.group(4):>5s} occ {m.group(5)}")
