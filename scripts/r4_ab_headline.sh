#!/bin/bash
# A/B on ONE box: the previous kernels (libzlhip_prev.so, see r4_ab_levels.sh) against the current ones on the headline shape, both legs of bench.py:
# the timed workload (2 s sources, cache-assisted) and the HBM-only leg (12 s sources, never re-read inside a launch).
O=gpurun_out/${1:-r4_abhead}; mkdir -p $O
PREV=$PWD/libzl_amd/lib/libzlhip_prev.so
run() { # label, env
  ( export $2; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-repeats --steps 12 --warmup 3 2>$O/err.log ) | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; nr=r.get('no_reuse_variant') or {}
print(f\"$1 | {d['value']:.4e} vs/s  {d['ms_per_step']:.4f} ms/step  K2 {r['frac']*100:.1f} % ({r['avg_launch_ms']:.4f} ms)  HBM-only {100*(nr.get('frac') or 0):.1f} % ({(nr.get('avg_launch_ms') or 0):.4f} ms x{nr.get('launches_per_step')})  check {all(c['bit_exact'] for c in d['output_check']['rows_vs_oracle'])}\")"
}
for rep in 1 2 3 4; do
  run "prev" ZLHIP_LIBRARY=$PREV | tee -a $O/ab.txt
  run "new " ZL_DUMMY=1 | tee -a $O/ab.txt
done
