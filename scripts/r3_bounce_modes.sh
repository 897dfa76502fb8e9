#!/bin/bash
# round 3: offline bounce, delivery through the copy engine per plan window (ZL_BOUNCE_DIRECT=0) against stores from the render kernel itself (1: 16 bit, 2: both)
for m in 0 1 2; do for shape in "4096 32 96000 3750" "1024 8 48000 8192"; do
  echo "--- ZL_BOUNCE_DIRECT=$m  $shape"
  ZL_BOUNCE_DIRECT=$m python3 scripts/bounce_bench.py --one $shape 2>&1 | grep -v amdgpu.ids | tail -3
done; done
