#!/bin/bash
# round-5 re-entry: GPU tests at HEAD, then block-length and lane-stagger A/Bs of the C2 line
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1; echo "pytest rc $?" >> $out/gputest.log
B="python bench.py --no-cpu-baseline --no-pmc --legs c2 --profile-steps 0"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), d['steps'], d.get('repeats'))"; }
for i in 1 2; do
  $B --steps 20 --warmup 5 2>/dev/null | line "steps20" >> $out/steps_ab.txt
  $B --steps 60 --warmup 10 2>/dev/null | line "steps60" >> $out/steps_ab.txt
  $B --steps 200 --warmup 10 2>/dev/null | line "steps200" >> $out/steps_ab.txt
done
for i in 1 2; do
  for c in 0 20000 100000 500000; do
    KVQ_LANE_STAGGER_CYCLES=$c $B --steps 20 --warmup 5 2>/dev/null | line "stagger$c" >> $out/stagger_ab.txt
  done
done
cat $out/steps_ab.txt $out/stagger_ab.txt; tail -3 $out/gputest.log
