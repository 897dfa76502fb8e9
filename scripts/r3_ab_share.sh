#!/bin/bash
# round 3: unit-step tap sharing (ZL_K2_UNIT_SHARE, libzlhip_share.so) against the shipped kernel, inside one gpurun call
set -o pipefail
out=gpurun_out/r3_ab_share.txt
: > $out
for a in "" "--loop-seconds 10" "--hermite" "--hermite --loop-seconds 10" "--frames 128" "--frames 128 --loop-seconds 10" \
         "--voices 4096 --buses 32 --fs 96000 --blocks-per-step 3750" "--voices 4096 --buses 32 --fs 96000 --blocks-per-step 3750 --loop-seconds 10" "--mono" "--mono --loop-seconds 10"; do
  bash scripts/ab_libs.sh "base share base share" "$a" >> $out 2>&1
done
cat $out
