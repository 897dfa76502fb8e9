#!/bin/bash
# Round-2 A/B sweep: the BASELINE config shapes the verdict names, each with 2 s loops (nominal, Infinity-Cache assisted)
# and with 10 s loops (no source byte is re-read inside a plan window: the HBM-honest figure).
# Usage: scripts/sweep_r2.sh <tag> [extra bench.py flags]
set -o pipefail
tag=${1:-r2}; shift
out=gpurun_out/sweep_$tag.jsonl
mkdir -p gpurun_out
: > $out
run() { echo "### $*" >> $out; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reuse-check --no-repeats --steps 6 --warmup 2 "$@" >> $out 2>gpurun_out/sweep_err.log || { echo "FAILED: $*" >> $out; tail -5 gpurun_out/sweep_err.log >> $out; return 1; }; }
for ls in 2 10; do
run --loop-seconds $ls "$@" &&
run --loop-seconds $ls --notes 48,72 "$@" &&
run --loop-seconds $ls --notes 48,72 --hermite "$@" &&
run --loop-seconds $ls --hermite "$@" &&
run --loop-seconds $ls --voices 64 --buses 8 --source-rate 44100 --notes 48,72 "$@" || exit 1
done
python3 - <<PY
import json
for l in open("$out"):
    if l.startswith("#") or l.startswith("FAILED"): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    r=d["roofline"]
    print(f'  value {d["value"]:.3e} vs/s  ms/step {d["ms_per_step"]:.3f}  K2 {r["achieved"]:.0f} GB/s ({r["frac"]*100:.1f}%)  K2 ms/launch {r["avg_launch_ms"]:.3f} x{r["launches_per_step"]}  B/vs {r["bytes_per_voice_sample"]:.2f} slow {r["slow_blocks"]}')
PY
