#!/bin/bash
# HBM-only leg of bench.py on ONE box, alternating: 10 s sources with launches of at most 10 s (the default) against sources as long as
# a whole call with the headline's one launch per call, at 8192 and at 8000 blocks per call (is the XCDs' 2 MiB spacing the difference?).
O=gpurun_out/${1:-r4_abn}; mkdir -p $O
for rep in 1 2; do for cfg in "10:8192" "0:8192" "0:8000" "0:7168" "10:8000"; do
  nr=${cfg%%:*}; kb=${cfg#*:}
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-repeats --steps 4 --warmup 1 --no-reuse-seconds $nr --blocks-per-step $kb 2>$O/err.log | python3 -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=d['roofline']; nr=r['no_reuse_variant']
print(f\"blocks per call $kb, sources {nr['loop_seconds']:.0f} s, window {nr['plan_window_blocks']} blocks: HBM-only {100*nr['frac']:.1f} %  ({nr['avg_launch_ms']:.3f} ms per launch, {nr['launches']} launches); timed workload {100*r['frac']:.1f} %, value {d['value']:.4e}\")" | tee -a $O/ab.txt
done; done
