#!/bin/bash
# AddressSanitizer + UBSan over the CPU builds (GPU sanitizers are not available on the pool): the host harness of the
# planning / per-frame code and of the ClipCommand scheduler (tests/cpu_harness) and the oracle, through the CPU tiers of the test suite;
# ThreadSanitizer over the request queue / parameter snapshots (zl_handoff.h).
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
H=tests/cpu_harness/_build/libzl_plan_host.so
S=tests/cpu_harness/_build/libzl_sched_host.so
O=oracle/_build/libzl_oracle.so
python3 -c "from libzl_amd import build; build.build_cpu_harness(); build.build_oracle()"
cp $H /tmp/zl_plan_host_backup.so; cp $S /tmp/zl_sched_host_backup.so; cp $O /tmp/zl_oracle_backup.so
restore() { cp /tmp/zl_plan_host_backup.so $H; cp /tmp/zl_sched_host_backup.so $S; cp /tmp/zl_oracle_backup.so $O; touch $H $S $O; }
trap restore EXIT
g++ -std=c++17 -O1 -g -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -Wl,-Bsymbolic \
    -I libzl_amd/csrc -I include -o $H tests/cpu_harness/plan_host.cpp
g++ -std=c++17 -O1 -g -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -Wl,-Bsymbolic \
    -I libzl_amd/csrc -I include -o $S tests/cpu_harness/sched_host.cpp
make -s -C oracle _build/libzl_oracle_asan.so && cp oracle/_build/libzl_oracle_asan.so $O
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_harness_parity.py tests/test_edge_cases.py tests/test_golden_cpu.py \
    tests/test_linear_runs.py tests/test_oracle_kat.py tests/test_scheduler.py tests/test_numpy_twin_random.py -x -q -m "not gpu"
# ThreadSanitizer over the cross-thread hand-off of the libzl-named layer (its own executable)
python3 -m pytest tests/test_handoff.py -x -q
