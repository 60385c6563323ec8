#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --legs c5 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); c=d['c5']; print('$1', round(d['value'],2), 'c5', c.get('value'), c.get('ms_per_step'), c.get('hipgraph'), c.get('error'))"; }
: > $out/c5_graph_ab.txt
for i in 1 2 3; do
  $B --graph 0 2>$out/c5g.err | line "eager" >> $out/c5_graph_ab.txt
  $B 2>>$out/c5g.err | line "replay" >> $out/c5_graph_ab.txt
done
cat $out/c5_graph_ab.txt; tail -3 $out/c5g.err
