#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
bash tools/ab_bench.sh $out/lane_threads_ab.txt 4 "KVQ_LANE_THREADS=0" "KVQ_LANE_THREADS=1" --legs c2 --steps 20 --warmup 5 --profile-steps 0
