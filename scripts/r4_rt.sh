#!/bin/bash
# Real-time evidence of round 4 on one MI355X box: the C++ cycle probe (the published figure; every cycle over 1 ms attributed), the setter-
# latency probe, and the Python harness (same engine; its slow cycles attributed too: interpreter vs engine).  Usage: scripts/r4_rt.sh <tag>
O=gpurun_out/${1:-r4_rt}; mkdir -p $O
./scripts/probes/_build/rt_cycle_probe ${2:-6000} > $O/realtime_cpp.txt 2> $O/realtime_cpp.err; echo "probe rc=$?"
./scripts/probes/_build/rt_setter_latency 3000 > $O/rt_setter_latency.txt 2>&1; echo "setter rc=$?"
python3 scripts/realtime_latency.py --quick 2>&1 | grep -v amdgpu.ids > $O/realtime_python.txt; echo "python rc=$?"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > $O/host_rt.txt
cat $O/realtime_cpp.txt
