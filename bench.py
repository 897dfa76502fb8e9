#!/usr/bin/env python3
"""bench.py -- voice-samples/sec of the sampler hot path on MI355X.

Workload (BASELINE.json metric: "voice-samples/sec at 1024 voices x 256-frame blocks; % HBM
roofline"): 1024 looping stereo voices on 8 buses x 128 voices, 48 kHz source and playback
(ratio 1), linear interpolation, reference-faithful mode, one distinct 2 s uniform(-1,1) source
per voice (786 MB, larger than the 256 MiB Infinity Cache), fractional-beat loops, per-clip
volume / pan, bus integer peaks every block.  One "step" = one zlhip_render_batch call of
--blocks-per-step consecutive 256-frame blocks with every input resident in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders its own 1024
voices (weak scaling) into a partial bus and the partial buses are summed onto rank 0 with one
RCCL reduce per step; rank 0 then scans the reduced bus for levels.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (K2 zl_k2_render):
algorithmic bytes per launch / its average duration measured with HIP events on the engine's
stream.  `cpu_baseline` times the CPU oracle (a port: the reference is not compilable here) on a
bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# One hardware queue per HIP stream in use (caller's stream, the engine's planning stream, RCCL's stream, the null
# stream): streams that share a hardware queue serialise behind each other.  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
VOICE_STATE_BYTES = 104        # sizeof(ZlVoiceState)


def build_scene(syn, torch, dev, voices_per_bus, num_buses, fs, loop_frames, seed, bpm=120, notes=(60, 60), source_rate=None, mono=False, beat_locked=False):
    """Registers one distinct stereo loop per voice (generated on the device) and starts every voice.
    `notes` = inclusive MIDI-note range drawn per voice (root note 60: 48..72 is pitch ratio 0.5..2)."""
    source_rate = source_rate or fs
    from libzl_amd import clip_command
    V = voices_per_bus * num_buses
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    for v in range(V):
        src = torch.rand((2, loop_frames), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0
        cid = syn.register_clip_device(src[0].data_ptr(), None if mono else src[1].data_ptr(), loop_frames, source_rate)
        assert cid == v
        del src
    torch.cuda.synchronize()
    rng = np.random.default_rng(seed)
    for v in range(V):
        p = syn.default_clip_params(loop_frames / source_rate)
        # fractional beat length -> deterministic sample-space loop wrap (SamplerSynthVoice.cpp:243-246)
        # an integer number of beats makes the restart clock-driven (beat-locked, :227-241) instead
        p.length_in_beats = 4.0 if beat_locked else 3.5
        p.length_seconds = float(np.float32((loop_frames - 64 - (v % 17)) / source_rate))
        p.volume_absolute = float(np.float32(rng.uniform(0.25, 1.0)))
        p.pan = float(np.float32(rng.uniform(-1.0, 1.0)))
        syn.set_clip_params(v, p)
    for v in range(V):
        bus, slot = divmod(v, voices_per_bus)
        note = 60 if notes[0] == notes[1] else int(rng.integers(notes[0], notes[1] + 1))
        cmd = clip_command(clip=v, midi_note=note, midi_channel=bus - 2, start_playback=1, looping=1,
                           change_volume=1, volume=float(np.float32(rng.uniform(0.1, 1.0))))
        assert syn.start_voice(bus, slot, cmd, 0) == 1


def cpu_baseline(args, seed):
    """Oracle (-O3 -march=native build) on a bounded sample of the same workload, buses partitioned
    over host threads (one RT thread per JACK client in the reference)."""
    from oracle import zl_oracle as zo
    from libzl_amd.engine import synthetic_clocks
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64, args.cpu_threads if args.cpu_threads > 0 else 16))
    V = 1024
    vpb = 8                                  # the reference's own voices per channel; 128 buses so threads can spread
    B = V // vpb
    loop = 24000                             # 0.5 s loops keep the sample's memory small (49 MB); arithmetic identical
    blocks = args.cpu_blocks
    osyn = zo.OracleSynth(B, vpb, args.fs, 0, max_sounds=V, fast=True)
    rng = np.random.default_rng(seed)
    for v in range(V):
        L = rng.uniform(-1, 1, loop).astype(np.float32)
        R = rng.uniform(-1, 1, loop).astype(np.float32)
        cid = osyn.register_clip(L, R, args.fs)
        clip = osyn.clips[cid]
        clip.lengthInBeats = 3.5
        clip.lengthInSeconds = float(np.float32((loop - 64 - (v % 17)) / args.fs))
        clip.volumeAbsolute = float(np.float32(rng.uniform(0.25, 1.0)))
        clip.pan = float(np.float32(rng.uniform(-1.0, 1.0)))
    for v in range(V):
        bus, slot = divmod(v, vpb)
        cmd = zo.clip_command(clip=v, midiNote=60, midiChannel=bus - 2, startPlayback=1, looping=1, changeVolume=1,
                              volume=float(np.float32(rng.uniform(0.1, 1.0))))
        assert osyn.start_voice(bus, slot, cmd, 0) == 1
    clocks = synthetic_clocks(blocks, args.frames, args.fs)
    osyn.render_batch(2, args.frames, clocks, threads=threads, want_reports=False)       # warm-up
    # bounded sample: repeat batches of `blocks` blocks until about args.cpu_seconds of CPU wall time have passed
    done = 0
    t0 = time.perf_counter()
    while True:
        osyn.render_batch(blocks, args.frames, clocks, threads=threads, want_reports=False)
        done += blocks
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds:
            break
    # the same oracle on one thread (a short sample): the reference renders a channel's voices on one RT thread
    t1 = time.perf_counter()
    osyn.render_batch(8, args.frames, clocks, threads=1, want_reports=False)
    single = V * 8 * args.frames / (time.perf_counter() - t1)
    return {
        "value": V * done * args.frames / dt, "unit": "voice-samples/s", "cores": threads, "kind": "port",
        "single_thread_value": single,
        "sample": f"{V} stereo voices (128 buses x 8, the reference's voices per channel) x {done} blocks x {args.frames} frames, "
                  f"0.5 s loops, ratio 1, faithful mode, oracle/zl_oracle.c built -O3 -march=native, {threads} threads "
                  f"(buses partitioned, one thread per JACK client as in the reference), {dt:.1f} s wall; {cores} host cores visible",
    }


def main():
    # stdout carries exactly ONE line, the JSON result: whatever native libraries print meanwhile (RCCL's version banner
    # at communicator creation, for one) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--buses", type=int, default=8)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--blocks-per-step", type=int, default=8192)
    ap.add_argument("--fs", type=float, default=48000.0)
    ap.add_argument("--loop-seconds", type=float, default=2.0)
    ap.add_argument("--notes", default="60,60", help="MIDI note range per voice (root 60); 48,72 = pitch ratio 0.5..2 (config 4)")
    ap.add_argument("--mono", action="store_true", help="mono sources (non-default variant)")
    ap.add_argument("--beat-locked", action="store_true", help="integer lengthInBeats: loops restart against the JACK clock (non-default variant)")
    ap.add_argument("--hermite", action="store_true", help="4-tap Hermite interpolation (ZLHIP_MODE_HERMITE, config 4)")
    ap.add_argument("--source-rate", type=float, default=0.0, help="sample rate of the sources (default: --fs)")
    ap.add_argument("--voices-per-task", type=int, default=0)
    ap.add_argument("--fanout", default="none", choices=["none", "fused", "separate"],
                    help="also produce the JackPassthrough fan-out [B][6][frames] of every bus (SURVEY 8f n1): fused into the bus "
                         "write (zlhip_render_batch_fanout) or as a separate pass over the bus (non-default variant, N = 1)")
    ap.add_argument("--plan-window", type=int, default=0)
    ap.add_argument("--cpu-blocks", type=int, default=64)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reuse-check", action="store_true", help="skip the auxiliary 10 s-loop measurement")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; the measured path) or gloo (rehearsal: reduce through host memory)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--reduce-algo", default="mesh", choices=["mesh", "reduce", "rank-order"],
                    help="N > 1: mesh = all-to-all + rank-order sum + gather over the xGMI mesh (default), reduce = one RCCL reduce")
    ap.add_argument("--rehearse-collectives", action="store_true", help="rehearsal only: run the N > 1 code path with one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU render path")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or args.rehearse_collectives
    if args.rehearse_collectives:
        os.environ["ZL_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from libzl_amd import SamplerSynth
    from libzl_amd.engine import synthetic_clocks

    V, B, N, KB = args.voices, args.buses, args.frames, args.blocks_per_step
    vpb = V // B
    notes = tuple(int(x) for x in args.notes.split(","))
    source_rate = args.source_rate or args.fs
    loop_frames = int(args.loop_seconds * source_rate)
    arena = (loop_frames + 16) * 8 * V + (1 << 20)
    syn = SamplerSynth(B, vpb, max_frames=N, max_batch_blocks=KB, max_sounds=V, mode=(4 if args.hermite else 0), playback_sample_rate=args.fs,
                       sound_arena_bytes=arena, voices_per_task=args.voices_per_task, plan_window_blocks=args.plan_window, device=local_rank)
    seed = 0x5A17 + 2 + 1000 * rank
    build_scene(syn, torch, dev, vpb, B, args.fs, loop_frames, seed, notes=notes, source_rate=source_rate, mono=args.mono, beat_locked=args.beat_locked)
    syn.set_profiling(not os.environ.get('ZL_BENCH_NOPROF'))      # diagnostic switch: cost of the per-launch HIP events

    # the engine renders into a torch-owned device buffer so the RCCL reduce needs no copy
    bus = torch.zeros((B, 2, KB * N), device=dev, dtype=torch.float32)
    # an explicit (non-default) torch stream: the engine launches on it and, for N > 1, RCCL orders its reduce after
    # the work queued on torch's *current* stream -- so this stream is made current for every step below
    stream = torch.cuda.Stream(device=dev)
    sptr = stream.cuda_stream
    clock_sets = [synthetic_clocks(KB, N, args.fs, start_block=i * KB) for i in range(args.warmup + args.steps)]

    render_ms = []
    plan_ms = []
    fin_ms = []
    src_bytes = 0
    launches = 1

    from libzl_amd import sharding
    overlapped = None
    if distributed and args.dist_backend == "nccl":
        overlapped = sharding.OverlappedBusReduce(syn, lambda: torch.zeros((B, 2, KB * N), device=dev, dtype=torch.float32), dst=0,
                                                  algorithm=args.reduce_algo)

    fan = fan_params = None
    if args.fanout != "none":
        from libzl_amd import PassthroughParams
        fan = torch.zeros((B, 6, KB * N), device=dev, dtype=torch.float32)
        fan_params = [PassthroughParams(0.9, 0.5, 0.25, 0.1 * (b % 3 - 1), 0) for b in range(B)]     # every pair multiplied

    def step(i, timed):
        # render this rank's voices, sum the partial buses onto rank 0 (one RCCL reduce), levels on the root
        if distributed and args.dist_backend != "nccl":
            # rehearsal path: the collective runs on a host copy of the partial bus
            syn.render_batch(KB, N, clock_sets[i], bus_out_dev=bus.data_ptr(), stream=sptr)
            torch.cuda.synchronize()
            host = bus.cpu()
            sharding.reduce_bus(host, dst=0)
            if rank == 0:
                bus.copy_(host)
                syn.levels_scan_device(bus.data_ptr(), KB, N, stream=sptr)
        elif distributed:
            overlapped.step(KB, N, clock_sets[i], stream=sptr)       # reduce of step i overlaps rendering of step i+1
        elif args.fanout == "fused":
            syn.render_batch(KB, N, clock_sets[i], bus_out_dev=bus.data_ptr(), stream=sptr, fan_params=fan_params, fan_out_dev=fan.data_ptr())
        else:
            syn.render_batch(KB, N, clock_sets[i], bus_out_dev=bus.data_ptr(), stream=sptr)
            if args.fanout == "separate":
                syn.passthrough(fan_params, bus.data_ptr(), fan.data_ptr(), KB * N, stream=sptr)

    torch.cuda.synchronize()
    torch.cuda.set_stream(stream)
    for i in range(args.warmup):
        try:
            step(i, False)
        except RuntimeError as err:
            # a collective this RCCL build refuses (every rank gets the same synchronous error): fall back to the plain
            # RCCL reduce, the most basic of the three exchanges, rather than losing the run
            if overlapped is None or overlapped.algorithm == "reduce":
                raise
            sys.stderr.write(f"bench: bus exchange '{overlapped.algorithm}' failed ({str(err)[:200]}); falling back to dist.reduce\n")
            args.reduce_algo = "reduce"
            overlapped = sharding.OverlappedBusReduce(syn, lambda: torch.zeros((B, 2, KB * N), device=dev, dtype=torch.float32), dst=0, algorithm="reduce")
            step(i, False)
    torch.cuda.synchronize()
    syn.profile_totals(reset=True)                                   # HIP-event sums start with the timed region
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # the steps are queued back to back: consecutive zlhip_render_batch calls pipeline (the planning of step i+1
    # overlaps the rendering of step i); the per-kernel HIP events of every step are read once, after the region
    for i in range(args.steps):
        step(args.warmup + i, True)
    if overlapped is not None:
        overlapped.flush(stream=sptr)                                # the last reduces + level scans are inside the timed region
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot, ncalls = syn.profile_totals()                               # sums of the HIP-event timings of the timed steps
    ncalls = max(1, ncalls)
    render_ms = [tot.render_ms / ncalls]; plan_ms = [tot.plan_ms / ncalls]; fin_ms = [tot.finalize_ms / ncalls]
    src_bytes = tot.source_bytes // ncalls
    slow = tot.slow_blocks
    launches = max(1, tot.render_launches // ncalls)
    if distributed:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # practical ceiling of this box: a device-to-device copy (read + write bytes / time), measured after the timed region
    copy_gbs = None
    if rank == 0:
        a = torch.empty(1 << 28, device=dev, dtype=torch.float32); b = torch.empty_like(a)      # 1 GiB each
        b.copy_(a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5): b.copy_(a)
        e1.record(stream); torch.cuda.synchronize()
        copy_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b

    # the same kernel with sources that are not re-read inside a plan window (10 s instead of 2 s loops: the Infinity
    # Cache cannot help): an auxiliary figure next to the headline one, N = 1 only, a few steps after the timed region
    no_reuse = None
    if rank == 0 and not distributed and not args.no_reuse_check and args.loop_seconds < 10.0:
        lf2 = int(10.0 * source_rate)
        syn2 = SamplerSynth(B, vpb, max_frames=N, max_batch_blocks=KB, max_sounds=V, mode=(4 if args.hermite else 0),
                            playback_sample_rate=args.fs, sound_arena_bytes=(lf2 + 16) * (4 if args.mono else 8) * V + (1 << 20),
                            voices_per_task=args.voices_per_task, plan_window_blocks=args.plan_window, device=local_rank)
        build_scene(syn2, torch, dev, vpb, B, args.fs, lf2, seed + 7, notes=notes, source_rate=source_rate, mono=args.mono)
        syn2.set_profiling(True)
        for i in range(2):
            syn2.render_batch(KB, N, clock_sets[i % len(clock_sets)], bus_out_dev=bus.data_ptr(), stream=sptr)
        syn2.profile_totals(reset=True)
        for i in range(4):
            syn2.render_batch(KB, N, clock_sets[(2 + i) % len(clock_sets)], bus_out_dev=bus.data_ptr(), stream=sptr)
        t2, n2 = syn2.profile_totals()
        b2 = (t2.source_bytes + n2 * B * 2 * N * 4 * KB) / max(1, t2.render_launches)
        ms2 = t2.render_ms / max(1, t2.render_launches)
        g2 = b2 / (ms2 * 1e-3) / 1e9
        no_reuse = {"loop_seconds": 10.0, "achieved": g2, "frac": g2 / HBM_PEAK_GBS, "avg_launch_ms": ms2,
                    "value": float(V) * KB * N * n2 / (t2.total_ms * 1e-3) if t2.total_ms > 0 else None}
        syn2.close()

    total_vs = float(V) * world * KB * N * args.steps
    value = total_vs / dt
    # algorithmic bytes of the K2 launches of one step: every source frame once per block
    # (summed by K1 per voice-block: (ceil(N*ratio)+taps-1)*channels*4) + the bus write; a step is `launches` K2 launches
    bus_bytes = B * 2 * N * 4 * KB * (4 if args.fanout == "fused" else 1)      # fused fan-out: three more stereo pairs per bus frame
    k2_bytes_step = src_bytes + bus_bytes
    state_bytes = V * 2 * VOICE_STATE_BYTES + B * 16 * KB
    k2_step_ms = float(np.mean(render_ms)) if render_ms else float("nan")      # sum of the step's K2 launches (HIP events)
    k2_avg_ms = k2_step_ms / launches
    k2_bytes = k2_bytes_step / launches
    achieved = k2_bytes / (k2_avg_ms * 1e-3) / 1e9 if k2_avg_ms > 0 else 0.0

    # HBM traffic of K2 per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this process):
    # traffic / algorithmic measured on this workload, applied to this run's algorithmic bytes; null for other workloads
    traffic = None
    default_workload = (V, B, N, KB, args.loop_seconds, notes, args.hermite, source_rate, args.mono, args.beat_locked) == (1024, 8, 256, 8192, 2.0, (60, 60), False, args.fs, False, False) and args.fanout == "none"
    pmc_file = os.path.join(ROOT, "profiles", "round1_d_pmc.json")
    if default_workload and os.path.exists(pmc_file):
        traffic = json.load(open(pmc_file))["traffic_over_algorithmic"] * k2_bytes

    if rank == 0:
        out = {
            "metric": "voice-samples/sec at 1024 voices x 256-frame blocks; % HBM roofline",
            "value": value, "unit": "voice-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{V} looping {'mono' if args.mono else 'stereo'} voices per GPU on {B} buses x {vpb}, {N}-frame blocks, {KB} blocks per step ({launches} K2 launches), "
                            + (f"fs=sr={args.fs:.0f} (ratio 1)" if notes[0] == notes[1] and source_rate == args.fs else
                               f"fs={args.fs:.0f}, sources at {source_rate:.0f}, MIDI notes {notes[0]}..{notes[1]} around root 60") +
                            f", {'4-tap Hermite' if args.hermite else 'linear'} interp, faithful mode, distinct {args.loop_seconds:g} s sources "
                            f"({arena / 1e6:.0f} MB), bus int peaks per block" + ({"none": "", "fused": ", JackPassthrough fan-out fused into the bus write", "separate": ", JackPassthrough fan-out as a separate pass"}[args.fanout]) + (f", bus reduce to rank 0 per step over RCCL ({args.reduce_algo}), overlapped with the next step" if distributed else ""),
                "voices_per_gpu": V, "buses": B, "frames_per_block": N, "blocks_per_step": KB, "parallelism": f"voices sharded x{world}",
            },
            "roofline": {
                "bound": "hbm", "kernel": "zl_k2_render", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/round1_d_pmc.json: rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic on this workload" if traffic else None,
                "no_reuse_variant": no_reuse,
                "device_copy_GBs": copy_gbs, "frac_of_device_copy": (achieved / copy_gbs) if copy_gbs else None,
                "algorithmic_bytes_per_launch": k2_bytes, "avg_launch_ms": k2_avg_ms, "launches_per_step": launches,
                "bytes_per_voice_sample": k2_bytes_step / (V * KB * N),
                "other_ms_per_step": {"planning_not_hidden (K0+K1+K1c of the first window; overlaps the previous step)": float(np.mean(plan_ms)),
                                      "K3 finalize + reports + launch gaps": float(np.mean(fin_ms))},
                "state_and_levels_bytes_per_step": state_bytes, "slow_blocks": int(slow),
                "note": "sources are 2 s loops re-read every 375 blocks: inside a plan window part of the re-reads is served by the "
                        "256 MiB Infinity Cache (bus-major launch order keeps one bus's 98 MB of sources hot); "
                        "no_reuse_variant is the same kernel on 10 s sources (DESIGN.md section 4)",
            },
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, seed)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    syn.close()
    if distributed:
        dist.barrier()                       # rank 0 is still timing the CPU baseline: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
