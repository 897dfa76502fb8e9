#!/bin/bash
# Per-GPU shares of BASELINE.json configs 2..5 through bench.py (no CPU leg).  Usage: scripts/config_sweep.sh <tag>
set -o pipefail
tag=${1:-r01}
out=gpurun_out/config_sweep_$tag.jsonl
mkdir -p gpurun_out
: > $out
run() { echo "### $*" >> $out; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reuse-check --no-repeats --steps 6 --warmup 2 "$@" >> $out 2>gpurun_out/config_sweep_err.log || { echo "FAILED: $*" >> $out; tail -5 gpurun_out/config_sweep_err.log >> $out; return 1; }; }
run --voices 64 --buses 8 --frames 256 &&
run --voices 64 --buses 8 --frames 256 --source-rate 44100 --notes 48,72 &&
run --voices 96 --buses 12 --frames 256 &&
run --voices 1024 --buses 8 --frames 128 &&
run --voices 1024 --buses 8 --notes 48,72 &&
run --voices 1024 --buses 8 --notes 48,72 --hermite &&
run --voices 1024 --buses 8 --hermite &&
run --voices 4096 --buses 32 --fs 96000 --loop-seconds 2 --blocks-per-step 3750 &&
# the HBM-only counterparts (10 s sources: nothing is re-read inside a plan window)
run --voices 1024 --buses 8 --loop-seconds 10 &&
run --voices 1024 --buses 8 --notes 48,72 --loop-seconds 10 &&
run --voices 1024 --buses 8 --notes 48,72 --hermite --loop-seconds 10 &&
run --voices 1024 --buses 8 --frames 128 --loop-seconds 10
python3 - <<PY
import json
for l in open("$out"):
    if l.startswith("#") or l.startswith("FAILED"): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    r=d["roofline"]
    print(f'  value {d["value"]:.3e} vs/s  ms/step {d["ms_per_step"]:.3f}  K2 {r["achieved"]:.0f} GB/s ({r["frac"]*100:.1f}%)  K2 ms/launch {r["avg_launch_ms"]:.3f} x{r["launches_per_step"]}  B/vs {r["bytes_per_voice_sample"]:.2f}  other {list(r["other_ms_per_step"].values())} slow {r["slow_blocks"]}')
PY
