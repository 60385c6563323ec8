#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --legs c2 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), d['config']['hipgraph'])"; }
: > $out/prio_ab.txt
for i in 1 2; do
  for v in "KVQ_LANE_PRIO=0" "KVQ_LANE_PRIO=-1,0,0,0" "KVQ_LANE_PRIO=-1,-1,0,0" "KVQ_LANE_PRIO=-1,-1,-1,-1"; do
    env $v $B 2>/dev/null | line "$v" >> $out/prio_ab.txt
  done
done
cat $out/prio_ab.txt
