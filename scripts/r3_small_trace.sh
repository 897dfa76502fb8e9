#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3_small_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-reuse-check --no-repeats --no-spot-check --steps 8 --warmup 2 --voices 64 --buses 8 --source-rate 44100 --notes 48,72 > $out/log.txt 2>&1
python3 scripts/r3_timeline.py $out 1800
