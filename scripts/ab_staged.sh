#!/bin/bash
# A/B of the K2 variants: register gather (ZL_K2_STAGED=0) against LDS-staged source windows (=2: every mode), on the BASELINE
# shapes the round-1 verdict names, with 2 s (Infinity-Cache assisted) and 10 s (HBM only) sources.  Usage: scripts/ab_staged.sh <tag>
set -o pipefail
tag=${1:-ab}
out=gpurun_out/ab_staged_$tag.jsonl
: > $out
run() { echo "### ZL_K2_STAGED=$ST $*" >> $out; ZL_K2_STAGED=$ST timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reuse-check --no-repeats --steps 6 --warmup 2 "$@" >> $out 2>gpurun_out/ab_err.log || { echo "FAILED: $*" >> $out; tail -5 gpurun_out/ab_err.log >> $out; return 1; }; }
for ls in 2 10; do for ST in 0 2; do
run --loop-seconds $ls --notes 48,72 --hermite &&
run --loop-seconds $ls --hermite &&
run --loop-seconds $ls --notes 48,72 &&
run --loop-seconds $ls &&
run --loop-seconds $ls --voices 64 --buses 8 --source-rate 44100 --notes 48,72 || exit 1
done; done
python3 - <<PY
import json
for l in open("$out"):
    if l.startswith("#") or l.startswith("FAILED"): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    r=d["roofline"]; c=d.get("output_check") or {}
    ok = all(x["bit_exact"] for x in (c.get("rows_vs_oracle") or []))
    print(f'  value {d["value"]:.3e} vs/s  ms/step {d["ms_per_step"]:.3f}  K2 {r["achieved"]:.0f} GB/s ({r["frac"]*100:.1f}%)  K2 ms/launch {r["avg_launch_ms"]:.3f} x{r["launches_per_step"]}  B/vs {r["bytes_per_voice_sample"]:.2f} slow {r["slow_blocks"]} check {"ok" if ok else "FAIL"}')
PY
