#!/bin/bash
# A/B on ONE box: the previous kernels (libzlhip_prev.so: git stash; build_variant('prev', ...); git stash pop) against the current ones -- the
# per-bus epilogue of K2 (peak conversion in one instruction, the four level reductions as one interleaved DPP sequence, report peaks over DPP
# instead of ds_bpermute).  Alternating runs; prints value and K2's share of the nominal peak from the HIP events.
O=gpurun_out/${1:-r4_ablv}; mkdir -p $O
PREV=$PWD/libzl_amd/lib/libzlhip_prev.so
run() { # label, env, args, steps
  ( export $2; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-repeats --no-reuse-check --steps $4 --warmup 2 $3 2>$O/err.log ) | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']
print(f\"$1 | {d['value']:.4e} vs/s  {d['ms_per_step']:.4f} ms/step  K2 {r['frac']*100:.1f} % ({r['avg_launch_ms']:.4f} ms x{r['launches_per_step']})  check {all(c['bit_exact'] for c in d['output_check']['rows_vs_oracle'])}\")"
}
for rep in 1 2 3; do
for shape in "64v:--voices 64 --buses 8:48" "64r:--voices 64 --buses 8 --source-rate 44100 --notes 48,72:48" "96v:--voices 96 --buses 12:48" "96r:--voices 96 --buses 12 --source-rate 44100 --notes 48,72:48" "256p:--voices 256 --buses 32 --notes 48,72:24" "headline::6"; do
  name=${shape%%:*}; rest=${shape#*:}; args=${rest%:*}; steps=${rest##*:}
  run "prev $name" ZLHIP_LIBRARY=$PREV "$args" $steps | tee -a $O/ab.txt
  run "new  $name" ZL_DUMMY=1 "$args" $steps | tee -a $O/ab.txt
done; done
