#!/bin/bash
# round 3: rolling software pipeline in K2 (ZL_K2_ROLL; libzlhip.so) against the build before it (libzlhip_head.so) and the same sources
# without it (libzlhip_noroll.so), inside one gpurun call.  usage: scripts/r3_ab_roll.sh "<libs>"
set -o pipefail
libs=${1:-"head base noroll head base"}
out=gpurun_out/r3_ab_roll.txt
: > $out
n=0
for a in "--voices 64 --buses 8 --source-rate 44100 --notes 48,72" "--voices 64 --buses 8" "--voices 96 --buses 12" "" "--loop-seconds 10" \
         "--notes 48,72" "--notes 48,72 --loop-seconds 10" "--notes 48,72 --hermite" "--notes 48,72 --hermite --loop-seconds 10" "--hermite" "--hermite --loop-seconds 10" \
         "--frames 128" "--frames 128 --loop-seconds 10" "--voices 4096 --buses 32 --fs 96000 --loop-seconds 2 --blocks-per-step 3750"; do
  n=$((n+1)); [ -n "$ROLL_FIRST" ] && [ $n -gt $ROLL_FIRST ] && break       # ROLL_FIRST=<k>: only the first k shapes
  bash scripts/ab_libs.sh "$libs" "$a" >> $out 2>&1
done
cat $out
