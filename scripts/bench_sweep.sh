#!/bin/bash
# runs bench.py with several argument sets, one JSON line each, into gpurun_out/<tag>.jsonl
TAG=$1; shift
mkdir -p gpurun_out
: > gpurun_out/$TAG.jsonl
while [ $# -gt 0 ]; do
  echo "### $1" >> gpurun_out/$TAG.jsonl
  python3 bench.py $1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/$TAG.jsonl
  shift
done
python3 - <<PY
import json
for line in open("gpurun_out/$TAG.jsonl"):
    if line.startswith("###"):
        print(line.strip()); continue
    try:
        d = json.loads(line); r = d["roofline"]
        print(f"  value={d['value']:.3e} vs/s  ms/step={d['ms_per_step']:.3f}  k2={r['avg_launch_ms']:.3f} ms  {r['achieved']:.0f} GB/s ({100*r['frac']:.1f}%)  plan={list(r['other_ms_per_step'].values())[0]:.3f} other={list(r['other_ms_per_step'].values())[1]:.3f} ms/step launches={r['launches_per_step']}")
    except Exception as e:
        print("  parse error", e, line[:200])
PY
