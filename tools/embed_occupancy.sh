#!/bin/bash
# experiment: patch_embed_kernel<..., FRAG> duration against workgroups per CU (KVQ_EMBED_LDS_PAD bytes of extra LDS per workgroup)
cd /tmp && export TMPDIR=/tmp
for pad in 0 7168 16384 30720 61440; do
  rm -rf /tmp/pf; KVQ_EMBED_LDS_PAD=$pad timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t --output-format csv -- python /root/repo/bench.py --legs c2 --no-pmc --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 --min-timed-s 0 --streams 1 >/dev/null 2>&1
  f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
  echo "pad $pad: $(grep -i 'patch_embed' $f | cut -d, -f2-6 | tail -1)"
done
