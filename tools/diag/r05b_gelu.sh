#!/bin/bash
# packed-fp16 GELU in the fused tails: kernel + end-to-end parity tests, then same-box alternating A/B against the fp32 evaluation (KVQ_BUILD_TAG=pk0)
# (the pk0 variant = KVQ_BUILD_TAG=pk0 KVQ_EXTRA_HIPCC_FLAGS="-DKVQ_GELU_PK16=0" python kvq-challenge-cvpr-ntire2024_amd/_build.py, built while the product default was KVQ_GELU_PK16=1;
#  with today's default the roles are swapped: build the tag with -DKVQ_GELU_PK16=1 to repeat the A/B)
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -x -q > $out/gelu_tests.log 2>&1; echo "pytest rc $?" >> $out/gelu_tests.log
tail -4 $out/gelu_tests.log
bash tools/ab_bench.sh $out/gelu_pk16_ab.txt 3 "KVQ_BUILD_TAG=pk0" "KVQ_GELU_VARIANT=pk16" --legs c5 --steps 20 --warmup 5 --profile-steps 0
