#!/bin/bash
# Round-3 evidence run on one MI355X box (gpurun).  Usage: scripts/profile_r3.sh <tag>
# = scripts/profile_r2.sh (default bench line, rocprofv3 kernel-trace + stats and PMC passes of the timed and of the HBM-only workload,
# pitched-Hermite counters, the config sweep WITH the output check against the oracle) + the round's own measurements.
tag=${1:-r3a}
O=gpurun_out/$tag
bash scripts/profile_r2.sh $tag
python3 scripts/realtime_latency.py --quick > $O/realtime.txt 2>&1
./scripts/probes/_build/rt_setter_latency 6000 > $O/rt_setter_latency.txt 2>&1
python3 scripts/bounce_bench.py 2>&1 | grep -v amdgpu.ids > $O/bounce.txt
# N > 1 rehearsal on the one GPU of the box: two ranks, gloo, bus-aligned partition (no data-path collective)
timeout -k 10 400 python3 bench.py --gpus 2 --steps 4 --warmup 1 \
    --dist-backend gloo --same-device --no-cpu-baseline --no-reuse-check --no-repeats > $O/two_ranks_one_gpu.json 2> $O/two_ranks_one_gpu.err
tail -c 600 $O/two_ranks_one_gpu.json
ls -la $O
