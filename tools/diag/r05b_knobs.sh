#!/bin/bash
# the round's geometry decisions re-checked under graph replay (20-step blocks), same box
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --legs c2 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), d['config']['hipgraph'])"; }
: > $out/knobs_graph_ab.txt
for i in 1 2; do
  for v in "KVQ_NONE=0" "KVQ_ATTN_QSPLIT_MAX=4" "KVQ_TAILMM_768=0" "KVQ_TAILMM_HC=256" "KVQ_LATENCY=1" "KVQ_MERGE_MAXC=128"; do
    env $v $B 2>/dev/null | line "$v" >> $out/knobs_graph_ab.txt
  done
done
cat $out/knobs_graph_ab.txt
