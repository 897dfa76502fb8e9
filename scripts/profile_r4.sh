#!/bin/bash
# The round's evidence run on one MI355X box (gpurun).  Usage: scripts/profile_r4.sh <tag>   (then: python3 scripts/publish_profile.py <tag> round4_x)
# Produces under gpurun_out/<tag>/: the default bench line (with cpu_baseline), rocprofv3 kernel-trace + stats and PMC passes (separate
# runs) for the timed (2 s sources) workload and for the no-reuse workload (12 s sources, plan windows of 2048 blocks = 10.9 s: HBM only), the config sweep with the binding
# unit of every shape (its own PMC pass per shape), the real-time probes, the bounce figures, the N > 1 rehearsals.
# (Rounds 2 and 3 collected profiles/round2_e_* and round3_f_* with the same bench / rocprofv3 commands.)
set -o pipefail
# A gpurun call is at most 20 minutes: the collection runs in three parts, each its own call -- scripts/profile_r4.sh <tag> a | b | c
#   a  bench line + kernel trace + five PMC passes, timed and HBM-only workload (~8 min)      b  config sweep with binding units (~12 min)
#   c  real-time probes, bounce, N > 1 rehearsals (~9 min)
tag=${1:-r4p}; part=${2:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
if [ $part = a ]; then
B="--no-cpu-baseline --no-reuse-check --no-spot-check --no-repeats --steps 4 --warmup 1"
python3 bench.py > $O/bench_line.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
for wl in "reuse:" "noreuse:--loop-seconds 12 --plan-window 2048"; do      # (noreuse = the shape of bench.py's own HBM-only leg: 12 s sources, plan windows of 2048 blocks = 10.9 s)
  name=${wl%%:*}; args=${wl#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_trace -- python3 bench.py $B $args > $O/${name}_trace.log 2>&1 || echo "trace $name failed"
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/${name}_pmc1 -- python3 bench.py $B $args > $O/${name}_pmc1.log 2>&1 || echo "pmc1 $name failed"
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${name}_pmc2 -- python3 bench.py $B $args > $O/${name}_pmc2.log 2>&1 || echo "pmc2 $name failed"
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/${name}_pmc3 -- python3 bench.py $B $args > $O/${name}_pmc3.log 2>&1 || echo "pmc3 $name failed"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/${name}_pmc4 -- python3 bench.py $B $args > $O/${name}_pmc4.log 2>&1 || echo "pmc4 $name failed"
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr --output-format csv -d $O/${name}_pmc5 -- python3 bench.py $B $args > $O/${name}_pmc5.log 2>&1 || echo "pmc5 $name failed"
  python3 scripts/summarize_prof.py $O/${name}_trace $O/${name}_trace_summary.txt
  for i in 1 2 3 4 5; do python3 scripts/summarize_prof.py $O/${name}_pmc$i $O/${name}_pmc${i}_summary.txt; done
  cat $O/${name}_trace_summary.txt $O/${name}_pmc?_summary.txt > $O/${name}_summary.txt
  rm -rf $O/${name}_trace $O/${name}_pmc?            # raw CSVs are large; the condensed summaries stay
done
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > $O/host.txt
fi
if [ $part = b ]; then
python3 scripts/config_sweep.py ${tag} > $O/config_sweep.log 2>&1; cp gpurun_out/config_sweep_${tag}.txt $O/config_sweep.txt; cat $O/config_sweep.txt
fi
if [ $part = c ]; then
bash scripts/r4_rt.sh ${tag} 3000 > $O/rt.log 2>&1
python3 scripts/bounce_bench.py 2>&1 | grep -v amdgpu.ids > $O/bounce.txt
bash scripts/r4_launch_rehearsal.sh ${tag} > $O/launch_rehearsal.log 2>&1
fi
ls -la $O | head -40
