#!/bin/bash
# trunk geometries of C2 (4 clips): stage 0..3, un-shifted / shifted type counts.  usage: attn_run.sh bin [bin ...]
cd "$(dirname "$0")"
for b in "$@"; do
  echo "== $b"
  timeout 60 ./$b 128 3 4 392 64 3 | head -2
  for g in "128 3 4 392 64" "128 3 4 392 128" "32 6 4 392 16" "8 12 4 392 8" "2 24 4 392 2"; do
    timeout 120 ./$b $g 20 | tail -1
  done
done
