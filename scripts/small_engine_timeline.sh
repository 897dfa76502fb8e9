#!/bin/bash
# Kernel timeline of consecutive pipelined calls of small engines (what sits between two calls' render kernels; is planning hidden?):
# rocprofv3 --kernel-trace of bench.py, condensed by summarize_prof.py (timeline of the last dispatches, gaps between K2 dispatches),
# and the host's view of the same calls (ZL_CALL_STAMPS=1).  Usage (GPU box): scripts/small_engine_timeline.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-small}; mkdir -p $O
for shape in "64r:--voices 64 --buses 8 --source-rate 44100 --notes 48,72" "96r:--voices 96 --buses 12 --source-rate 44100 --notes 48,72" "64u:--voices 64 --buses 8"; do
  name=${shape%%:*}; args=${shape#*:}
  rocprofv3 --kernel-trace --output-format csv -d $O/${name}_trace -- python3 bench.py --no-cpu-baseline --no-reuse-check --no-spot-check --no-repeats --steps 8 --warmup 2 $args > $O/${name}.log 2>&1
  python3 scripts/summarize_prof.py $O/${name}_trace $O/${name}_summary.txt; rm -rf $O/${name}_trace
  echo "=== $name: $args"; grep -A18 "timeline of the last" $O/${name}_summary.txt | head -17; grep -m1 "gaps between" $O/${name}_summary.txt
  ZL_CALL_STAMPS=1 python3 bench.py --no-cpu-baseline --no-reuse-check --no-spot-check --no-repeats --steps 8 --warmup 2 $args 2>&1 | grep "zlhip_render_batch #" | tail -6
done
