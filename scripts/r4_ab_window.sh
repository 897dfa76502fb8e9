#!/bin/bash
# A/B on ONE box: plan windows of the fixed size (ZL_WINDOW_MUL=1, rounds 1-3) against the per-call choice of round 4 (one window per call
# when every playing voice is cheap to plan).  Alternating runs; prints value, K2 fractions (timed and HBM-only).
O=gpurun_out/${1:-r4_abw}; mkdir -p $O
run() { # label, env, args
  ( export $2; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-repeats --steps 8 --warmup 2 $3 2>$O/err.log ) | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; nr=r.get('no_reuse_variant') or {}
print(f\"$1 | {d['value']:.4e} vs/s  {d['ms_per_step']:.3f} ms/step  K2 {r['frac']*100:.1f} % x{r['launches_per_step']}  HBM-only {100*(nr.get('frac') or 0):.1f} % ({nr.get('loop_seconds')} s sources)  check {all(c['bit_exact'] for c in d['output_check']['rows_vs_oracle'])}\")"
}
for rep in 1 2; do
for shape in ":" "4096v96k:--voices 4096 --buses 32 --fs 96000 --blocks-per-step 3750" "128f:--frames 128" "64v:--voices 64 --buses 8" "96v:--voices 96 --buses 12" "herm1:--hermite"; do
  name=${shape%%:*}; args=${shape#*:}
  run "fixed  ${name:-headline}" ZL_WINDOW_MUL=1 "$args" | tee -a $O/ab.txt
  run "auto   ${name:-headline}" ZL_DUMMY=1 "$args" | tee -a $O/ab.txt
done; done
