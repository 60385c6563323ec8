#!/bin/bash
# trunk geometries of C2 (4 clips): stage 0..3, un-shifted / shifted (depth-split) blocks.  usage: attn32_run.sh bin [bin ...]
cd "$(dirname "$0")"
for b in "$@"; do
  echo "== $b"
  timeout 60 ./$b 128 3 4 392 64 3 | head -3
  timeout 60 ./$b 32 6 4 392 32 3 16 1 | head -3
  # (append "0 8" to a geometry for the HBM-cold image the trunk sees: 8 rotating copies)
  for g in "128 3 4 392 64 20 -1" "128 3 4 392 128 20 64" "32 6 4 392 16 20 -1" "32 6 4 392 32 20 16" "8 12 4 392 4 20 -1" "8 12 4 392 8 20 4" "2 24 4 392 1 20 -1" "2 24 4 392 2 20 1"; do
    timeout 120 ./$b $g | tail -1
  done
done
