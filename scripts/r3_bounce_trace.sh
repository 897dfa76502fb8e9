#!/bin/bash
# round 3: where a bounce's time goes -- kernel + memory-copy trace of the configs[4] share, and the per-row copy experiment
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3_bounce_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 scripts/bounce_bench.py --one 4096 32 96000 3750 > $out/log.txt 2>&1
cat $out/log.txt | tail -5
python3 scripts/r3_bounce_timeline.py $out > gpurun_out/r3_bounce_timeline.txt 2>&1; tail -60 gpurun_out/r3_bounce_timeline.txt
echo "--- plain"
python3 scripts/bounce_bench.py --one 4096 32 96000 3750 2>&1 | tail -4
echo "--- ZL_BOUNCE_STAMPS"
ZL_BOUNCE_STAMPS=1 python3 scripts/bounce_bench.py --one 4096 32 96000 3750 2>&1 | tail -12
