#!/bin/bash
# Collects the rocprofv3 kernel trace + stats and PMC counters (separate passes) of bench.py on the
# GPU box; raw CSVs stay in /tmp, only the summary is written under gpurun_out/.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-prof}
shift || true
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/zlprof_$TAG
rm -rf $RAW; mkdir -p $RAW $REPO/gpurun_out
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 bench.py $ARGS > $RAW/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $RAW/pmc1 -- python3 bench.py $ARGS > $RAW/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $RAW/pmc2 -- python3 bench.py $ARGS > $RAW/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $RAW/pmc3 -- python3 bench.py $ARGS > $RAW/pmc3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $RAW/pmc4 -- python3 bench.py $ARGS > $RAW/pmc4.log 2>&1 || true
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $RAW/pmc5 -- python3 bench.py $ARGS > $RAW/pmc5.log 2>&1 || true
python3 scripts/summarize_prof.py $RAW gpurun_out/${TAG}_summary.txt
tail -1 $RAW/trace.log > gpurun_out/${TAG}_bench_line.json
wc -c gpurun_out/${TAG}_summary.txt
