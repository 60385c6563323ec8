#!/bin/bash
# HBM bytes of the sampler / patch-embedding launches of a C2 step, both sequencings (one counter per pass).
cd /tmp && export TMPDIR=/tmp
for m in "" "--two-launch-sampler"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; timeout 240 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pm -o c -- python /root/repo/bench.py --probe c2 --probe-steps 2 $m >/dev/null 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    echo "== mode '$m' $ctr"
    [ -n "$f" ] && python /root/repo/tools/pmc_embed_sum.py "$f"
  done
done
