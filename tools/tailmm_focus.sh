#!/bin/bash
# register / spill table of ONE form of the wide fused tail in seconds (study builds; nothing is linked):
#   tools/tailmm_focus.sh <CF*1000 + HC/128*10*10 + TT, e.g. 3102 = C 384, HC 128, 64 tokens> [extra -D flags...]
f=$1; shift
d=/tmp/tmf_$f; mkdir -p $d
( cd kvq-challenge-cvpr-ntire2024_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable \
    -fno-slp-vectorize -fno-honor-nans -DKVQ_TAILMM_FOCUS=$f "$@" -Rpass-analysis=kernel-resource-usage --save-temps=obj -c tailmm.hip -o $d/tailmm.o > $d/res.txt 2>&1 )
python tools/kres.py $d/res.txt "block_tailmm_kernel<kvq::Fp16"
