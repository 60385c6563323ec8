#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --legs c3 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); c=d['c3']; print('$1', round(d['value'],2), 'c3', c.get('value'), c.get('ms_per_step'), c.get('hipgraph'), c.get('finite'), c.get('error'))"; }
: > $out/c3_graph_ab.txt
for i in 1 2 3; do
  $B --graph 0 2>$out/c3g.err | line "eager" >> $out/c3_graph_ab.txt
  $B 2>>$out/c3g.err | line "replay" >> $out/c3_graph_ab.txt
done
cat $out/c3_graph_ab.txt; grep -v amdgpu.ids $out/c3g.err | tail -5
