#!/bin/bash
# A tagged study build of the library that differs from the product build in csrc/tailmm.hip only, in seconds:
#   tools/tailmm_variant.sh <tag> [-DKVQ_TAILMM_FOCUS=3102] [-DKVQ_TAIL_TRACE] [other -D flags]
# every other object is taken from kvq-challenge-cvpr-ntire2024_amd/build/ (the product build must be fresh); the result is
# libkvq_hip_<tag>.so, loaded with KVQ_BUILD_TAG=<tag>.  With KVQ_TAILMM_FOCUS only ONE form of the kernel exists in the variant.
set -e
tag=$1; shift
pkg=kvq-challenge-cvpr-ntire2024_amd
mkdir -p $pkg/build_$tag
cp -p $pkg/build/*.o $pkg/build_$tag/; rm -f $pkg/build_$tag/tailmm.hip.p*.o      # the variant compiles tailmm.hip as ONE unit
echo "$@" > $pkg/build_$tag/flags.txt
( cd $pkg/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -fno-honor-nans \
    "$@" -Rpass-analysis=kernel-resource-usage -c tailmm.hip -o ../build_$tag/tailmm.hip.o 2> ../build_$tag/tailmm.res.txt )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $pkg/build_$tag/*.o -o $pkg/libkvq_hip_$tag.so
python tools/kres.py $pkg/build_$tag/tailmm.res.txt block_tailmm_kernel
