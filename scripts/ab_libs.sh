#!/bin/bash
# bench.py over several builds of the library (libzl_amd/lib/libzlhip_<name>.so, scripts/: build_variant) x bench argument sets.
# usage: scripts/ab_libs.sh "<name> <name> ..." "<args 1>" "<args 2>" ...   ('base' = libzlhip.so; env ZL_K2_STAGED is passed through)
libs=$1; shift
for a in "$@"; do for l in $libs; do
  lib=libzl_amd/lib/libzlhip_$l.so; [ $l = base ] && lib=libzl_amd/lib/libzlhip.so
  ZLHIP_LIBRARY=$PWD/$lib timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-reuse-check --no-repeats --no-spot-check --steps 6 --warmup 2 $a 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(f'%-14s %-44s {d[\"value\"]:.3e} vs/s  K2 {r[\"achieved\"]:.0f} GB/s ({r[\"frac\"]*100:.1f}%%)  {r[\"avg_launch_ms\"]:.3f} ms/launch' % ('$l', '$a'))" || { echo "$l $a FAILED"; tail -3 gpurun_out/ab_err.log; }
done; done
