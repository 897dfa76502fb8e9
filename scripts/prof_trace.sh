#!/bin/bash
# kernel trace + stats only (fast); summary under gpurun_out/<tag>_trace.txt
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-trace}; shift || true
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/zltrace_$TAG; rm -rf $RAW; mkdir -p $RAW $REPO/gpurun_out
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 bench.py $ARGS > $RAW/trace.log 2>&1
python3 scripts/summarize_prof.py $RAW gpurun_out/${TAG}_trace.txt
grep -E "zl_k|Name" gpurun_out/${TAG}_trace.txt | cut -c1-200 | head -20
