#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gputest4.log 2>&1; echo "pytest rc $?" >> $out/gputest4.log; tail -6 $out/gputest4.log
python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 > $out/bench_graph_line.json 2> $out/bench_graph_err.txt; tail -3 $out/bench_graph_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b/bench_graph_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['hipgraph'])
for k in ('no_sampler','two_launch_sampler','bf16','bf16_init','batch8','c3','c5','ksvqe','ksvqe96'):
    v=d.get(k)
    if isinstance(v,dict): print(k, v.get('value'), v.get('error'), {kk:v[kk] for kk in v if 'dscore' in kk or 'equal' in kk})
PY
