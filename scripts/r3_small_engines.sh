#!/bin/bash
# round 3: small engines (BASELINE configs[1] and the reference's 12 x 8): occupancy of the render kernel (ZL_K2_LDS_PAD: 10240 = 5 workgroups per CU, 0 = 6)
B="--no-cpu-baseline --no-reuse-check --no-repeats --no-spot-check --steps 8 --warmup 2"
for pad in 10240 0; do for a in "--voices 64 --buses 8 --source-rate 44100 --notes 48,72" "--voices 64 --buses 8" "--voices 96 --buses 12" "--voices 96 --buses 12 --source-rate 44100 --notes 48,72" "--voices 256 --buses 8 --notes 48,72"; do
  ZL_K2_LDS_PAD=$pad python3 bench.py $B $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(f'pad %-6s %-62s {d[\"value\"]:.3e} vs/s  {d[\"ms_per_step\"]:.3f} ms/step  K2 {r[\"achieved\"]:.0f} GB/s ({r[\"frac\"]*100:.1f}%%)  {r[\"avg_launch_ms\"]:.3f} ms/launch x{r[\"launches_per_step\"]}' % ('$pad', '$a'))"
done; done
