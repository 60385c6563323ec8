#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --legs c2 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), d['steps'], d.get('repeats'))"; }
for i in 1 2 3; do
  $B --no-sampler 2>/dev/null | line "eager_presampled" >> $out/graph_ab.txt
  $B --graph 1 2>/dev/null | line "graph_presampled" >> $out/graph_ab.txt
done
cat $out/graph_ab.txt
