#!/bin/bash
# End-of-round verification on the GPU box: tests, smoke, default bench line, profile, config sweep, probes, rehearsals of
# the N > 1 path.  Usage: scripts/final_check.sh <tag>   (writes gpurun_out/<tag>_*)
set -o pipefail
tag=${1:-final}
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/${tag}_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python3 bench.py > gpurun_out/${tag}_bench_line.json 2>gpurun_out/${tag}_bench.err; echo "bench rc=$?"
bash scripts/prof_counters.sh ${tag}p --steps 4 --warmup 1 --no-cpu-baseline --no-reuse-check > gpurun_out/${tag}_prof.log 2>&1; echo "prof rc=$?"
bash scripts/config_sweep.sh $tag > gpurun_out/${tag}_config_sweep.txt 2>&1; echo "sweep rc=$?"
timeout -k 10 300 python3 scripts/realtime_latency.py > gpurun_out/${tag}_realtime.txt 2>&1; echo "realtime rc=$?"
timeout -k 10 300 python3 scripts/command_storm.py > gpurun_out/${tag}_command_storm.txt 2>&1; echo "storm rc=$?"
timeout -k 10 400 python3 scripts/fanout_probe.py > gpurun_out/${tag}_fanout.txt 2>&1; echo "fanout rc=$?"
timeout -k 10 300 python3 bench.py --rehearse-collectives --steps 3 --warmup 1 --no-cpu-baseline --no-reuse-check > gpurun_out/${tag}_rehearse_rccl.json 2>gpurun_out/${tag}_rehearse_rccl.err; echo "rccl single-rank rehearsal rc=$?"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --dist-backend gloo --same-device --no-cpu-baseline --no-reuse-check > gpurun_out/${tag}_rehearse_gloo2.json 2>gpurun_out/${tag}_rehearse_gloo2.err; echo "2-rank gloo rehearsal rc=$?"
