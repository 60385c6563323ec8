#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --legs c2 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), d['steps'], d.get('repeats'), d['config']['hipgraph'])"; }
: > $out/graph_lanes_ab.txt
for i in 1 2; do
  for n in 4 5 6 8 3; do
    $B --streams $n 2>/dev/null | line "lanes$n" >> $out/graph_lanes_ab.txt
  done
done
GPU_MAX_HW_QUEUES=8 $B --streams 8 2>/dev/null | line "lanes8_q8" >> $out/graph_lanes_ab.txt
GPU_MAX_HW_QUEUES=8 $B --streams 6 2>/dev/null | line "lanes6_q8" >> $out/graph_lanes_ab.txt
cat $out/graph_lanes_ab.txt
