#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py -m gpu -x -q -k "pre_pool or feature_taps or ksvqe" > $out/new_tests.log 2>&1; echo "rc $?" >> $out/new_tests.log; tail -5 $out/new_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gputest2.log 2>&1; echo "pytest rc $?" >> $out/gputest2.log; tail -4 $out/gputest2.log
