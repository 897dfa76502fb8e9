#!/bin/bash
# A/B on ONE box: K2's split tail (ZL_K2_TAIL=0 / 1): the last blocks of a narrow-bus launch rendered by four short workgroups each.
O=gpurun_out/${1:-r4_abtail}; mkdir -p $O
run() { # label, env, args, steps
  ( export $2; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-repeats --no-reuse-check --steps $4 --warmup 2 $3 2>$O/err.log ) | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']
print(f\"$1 | {d['value']:.4e} vs/s  {d['ms_per_step']:.4f} ms/step  K2 {r['frac']*100:.1f} % ({r['avg_launch_ms']:.4f} ms x{r['launches_per_step']})  check {all(c['bit_exact'] for c in d['output_check']['rows_vs_oracle'])}\")"
}
for rep in 1 2 3; do
for shape in "64v:--voices 64 --buses 8:48" "64r:--voices 64 --buses 8 --source-rate 44100 --notes 48,72:48" "96v:--voices 96 --buses 12:48" "96r:--voices 96 --buses 12 --source-rate 44100 --notes 48,72:48"; do
  name=${shape%%:*}; rest=${shape#*:}; args=${rest%:*}; steps=${rest##*:}
  run "whole $name" ZL_K2_TAIL=0 "$args" $steps | tee -a $O/ab.txt
  run "split $name" ZL_K2_TAIL=1 "$args" $steps | tee -a $O/ab.txt
done; done
