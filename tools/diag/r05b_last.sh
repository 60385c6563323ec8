#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; tail -2 $out/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['hipgraph'], d['roofline']['frac'], d['roofline'].get('whole_step_traffic'), d['cpu_baseline']['value'])
PY
