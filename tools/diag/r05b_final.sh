#!/bin/bash
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gputest3.log 2>&1; echo "pytest rc $?" >> $out/gputest3.log; tail -4 $out/gputest3.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log; tail -3 $out/smoke.log
bash tools/collect_profiles.sh r05
