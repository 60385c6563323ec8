#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
bash tools/ab_bench.sh $out/qsplit_graph_ab.txt 5 "KVQ_ATTN_QSPLIT_MAX=1" "KVQ_ATTN_QSPLIT_MAX=4" --legs c2 --steps 20 --warmup 5 --profile-steps 0
bash tools/ab_bench.sh $out/qsplit_graph_ab60.txt 3 "KVQ_ATTN_QSPLIT_MAX=1" "KVQ_ATTN_QSPLIT_MAX=4" --legs c5 --steps 60 --warmup 10 --profile-steps 0
