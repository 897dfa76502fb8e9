#!/bin/bash
# A/B of library builds on the same box: scripts/ab_bench.sh "<bench args>" lib1.so lib2.so ...   (alternating, 2 rounds)
set -o pipefail
ARGS=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    ZLHIP_LIBRARY=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reuse-check --steps 8 --warmup 2 $ARGS > gpurun_out/ab_tmp.json 2>gpurun_out/ab_err.log || { echo "FAILED $lib"; tail -3 gpurun_out/ab_err.log; continue; }
    python3 - "$lib" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json")); r=d["roofline"]
print(f'{sys.argv[1]:48s} value {d["value"]:.4e}  ms/step {d["ms_per_step"]:.3f}  K2 {r["achieved"]:.0f} GB/s  launch {r["avg_launch_ms"]*1e3:.1f} us  other {[round(x,3) for x in r["other_ms_per_step"].values()]}')
PY
  done
done
