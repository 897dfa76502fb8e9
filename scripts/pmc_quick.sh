#!/bin/bash
# rocprofv3 PMC pass over bench.py for one library build: usage scripts/pmc_quick.sh <tag> <lib name|base> "<counters>" <bench args...>
tag=$1; l=$2; ctr=$3; shift 3
lib=libzl_amd/lib/libzlhip_$l.so; [ $l = base ] && lib=libzl_amd/lib/libzlhip.so
export ZLHIP_LIBRARY=$GRAFT_REPO_ROOT/$lib
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reuse-check --no-repeats --no-spot-check "$@" > $out/log.txt 2>&1
python3 scripts/summarize_prof.py $out $out/summary.txt
grep "zl_k2_render" $out/summary.txt
