#!/bin/bash
# marginal cost of each launch family in the multi-lane C2 bench line: the step with the family's launches left out (KVQ_SKIP, csrc/plan.hip).
# KVQ_SKIP is honoured by DIAGNOSTIC builds only: build the variant first, on the build host (it travels with the gpurun snapshot):
#   KVQ_BUILD_TAG=diag KVQ_EXTRA_HIPCC_FLAGS=-DKVQ_DIAG python -c 'import kvq_amd; from kvq_amd import _build; _build.build()'
export KVQ_BUILD_TAG=diag
out=${1:-gpurun_out/skip_ablation.txt}
mkdir -p $(dirname $out); : > $out
for m in 0 1 2 4 8 16 32 64 128 0; do
  line=$(KVQ_SKIP=$m timeout -s KILL 150 python bench.py --legs c2 --no-cpu-baseline --no-pmc --profile-steps 0 2>/dev/null | tail -1)
  python -c "import json,sys; d=json.loads(sys.argv[2]); print('KVQ_SKIP=%-4s ms/step %.4f' % (sys.argv[1], d['ms_per_step']))" $m "$line" >> $out
done
cat $out
