#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "adaptive" > $out/adaptive_tests.log 2>&1; echo "rc $?" >> $out/adaptive_tests.log; tail -15 $out/adaptive_tests.log
