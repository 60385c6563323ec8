#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_harness.py -m gpu -x -q -k "slot or fragment_source or graph or hipgraph" > $out/slot_tests.log 2>&1; echo "rc $?" >> $out/slot_tests.log; tail -12 $out/slot_tests.log
B="python bench.py --no-cpu-baseline --no-pmc --legs c2 --profile-steps 0 --steps 20 --warmup 5"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), d['steps'], d.get('repeats'), d['score_checksum'])"; }
: > $out/graph_slot_ab.txt
for i in 1 2 3; do
  $B 2>$out/graph_err.txt | line "eager" >> $out/graph_slot_ab.txt
  $B --graph 1 2>>$out/graph_err.txt | line "graph_slot" >> $out/graph_slot_ab.txt
done
cat $out/graph_slot_ab.txt; tail -5 $out/graph_err.txt
