#!/bin/bash
# Rehearsal of `bench.py --gpus N` on a one-GPU box: (a) the bare command becomes 2 ranks by itself (both on GPU 0, gloo: the
# exchange goes through host memory), headline + span_buses sub-record; (b) the RCCL calls of the span_buses legs with one rank.
O=gpurun_out/${1:-r4_launch}
mkdir -p $O
timeout -k 10 400 python3 bench.py --gpus 2 --steps 4 --warmup 1 --dist-backend gloo --same-device --no-reuse-check --no-repeats > $O/two_ranks_one_gpu.json 2> $O/two_ranks_one_gpu.err
echo "rc=$?"; tail -c 3000 $O/two_ranks_one_gpu.json
timeout -k 10 300 python3 bench.py --rehearse-collectives --steps 4 --warmup 1 --no-cpu-baseline --no-reuse-check --no-repeats > $O/rccl_one_rank.json 2> $O/rccl_one_rank.err
echo "rc=$?"; tail -c 3000 $O/rccl_one_rank.json
