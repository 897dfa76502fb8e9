#!/usr/bin/env python3
"""bench.py -- voice-samples/sec of the sampler hot path on MI355X.

Workload (BASELINE.json metric: "voice-samples/sec at 1024 voices x 256-frame blocks; % HBM
roofline"): 1024 looping stereo voices on 8 buses x 128 voices, 48 kHz source and playback
(ratio 1), linear interpolation, reference-faithful mode, one distinct 2 s uniform(-1,1) source
per voice (786 MB, larger than the 256 MiB Infinity Cache), fractional-beat loops, per-clip
volume / pan, bus integer peaks every block.  One "step" = one zlhip_render_batch call of
--blocks-per-step consecutive 256-frame blocks with every input resident in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU), weak scaling -- every rank owns 1024 voices:
  default      bus-aligned partition: the job has N x 8 buses and every GPU owns 8 whole buses, so
               the only coupling of the path -- the per-bus sum of SamplerChannel::process, SamplerSynth.cpp:134-140 -- stays
               on one GPU and NO data-path collective runs; every rank meters its own buses.
  --span-buses the job has 8 buses that each span all ranks (BASELINE configs[3]: one stereo bus over 8 GPUs): partial
               buses are exchanged over the xGMI mesh (all-to-all), every rank sums the pieces it received in rank order and
               scans them for levels in ONE HIP kernel (zlhip_bus_reduce_sum_scan), the reduced pieces are gathered on rank 0.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (K2 zl_k2_render): algorithmic bytes per launch / its
average duration measured with HIP events on the launch stream, for the timed workload (Infinity-Cache assisted: 2 s sources
are re-read inside a plan window) AND for the same kernel on sources that are never re-read (frac_hbm_no_reuse: HBM only).
After the timed region one more step is rendered and 3 random (bus, block) rows of it are compared with the CPU oracle, bit
for bit; a mismatch fails the run.  `cpu_baseline` times the CPU oracle (a port: the reference is not compilable here) on a
bounded sample of the same workload on all of this box's host cores.
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


def scene_params(V, voices_per_bus, loop_frames, source_rate, seed, notes=(60, 60), beat_locked=False):
    """Per-voice clip / command parameters of the synthetic scene (host tables; the same draws feed the engine, the spot
    check's oracle and the CPU baseline)."""
    rng = np.random.default_rng(seed)
    vol, pan = [], []
    for v in range(V):
        vol.append(float(np.float32(rng.uniform(0.25, 1.0)))); pan.append(float(np.float32(rng.uniform(-1.0, 1.0))))
    note, vel = [], []
    for v in range(V):
        note.append(60 if notes[0] == notes[1] else int(rng.integers(notes[0], notes[1] + 1)))
        vel.append(float(np.float32(rng.uniform(0.1, 1.0))))
    return {
        # fractional beat length -> deterministic sample-space loop wrap (SamplerSynthVoice.cpp:243-246); an integer number
        # of beats makes the restart clock-driven (beat-locked, :227-241) instead
        "length_in_beats": 4.0 if beat_locked else 3.5,
        "length_seconds": [float(np.float32((loop_frames - 64 - (v % 17)) / source_rate)) for v in range(V)],
        "volume_absolute": vol, "pan": pan, "note": note, "velocity": vel,
    }


def build_scene(syn, torch, dev, voices_per_bus, num_buses, fs, loop_frames, seed, bpm=120, notes=(60, 60), source_rate=None, mono=False,
                beat_locked=False, keep_buses=()):
    """Registers one distinct stereo loop per voice (generated on the device) and starts every voice.
    `notes` = inclusive MIDI-note range drawn per voice (root note 60: 48..72 is pitch ratio 0.5..2).
    keep_buses: buses whose sources are also kept on the host (for the output spot check against the oracle).
    Returns (parameter tables, {voice: (left, right | None) host arrays})."""
    source_rate = source_rate or fs
    from libzl_amd import clip_command
    V = voices_per_bus * num_buses
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    kept = {}
    for v in range(V):
        src = torch.rand((2, loop_frames), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0
        cid = syn.register_clip_device(src[0].data_ptr(), None if mono else src[1].data_ptr(), loop_frames, source_rate)
        assert cid == v
        if v // voices_per_bus in keep_buses:
            h = src.cpu().numpy()
            kept[v] = (np.ascontiguousarray(h[0]), None if mono else np.ascontiguousarray(h[1]))
        del src
    torch.cuda.synchronize()
    P = scene_params(V, voices_per_bus, loop_frames, source_rate, seed, notes=notes, beat_locked=beat_locked)
    for v in range(V):
        p = syn.default_clip_params(loop_frames / source_rate)
        p.length_in_beats = P["length_in_beats"]
        p.length_seconds = P["length_seconds"][v]
        p.volume_absolute = P["volume_absolute"][v]
        p.pan = P["pan"][v]
        syn.set_clip_params(v, p)
    for v in range(V):
        bus, slot = divmod(v, voices_per_bus)
        cmd = clip_command(clip=v, midi_note=P["note"][v], midi_channel=bus - 2, start_playback=1, looping=1,
                           change_volume=1, volume=P["velocity"][v])
        assert syn.start_voice(bus, slot, cmd, 0) == 1
    return P, kept


def spot_check(syn, bus_np_rows, reports_before, P, kept, picks, args, clocks, voices_per_bus, loop_frames, source_rate, mode):
    """Output check of the benchmark itself: `picks` = [(bus, block)] rows of one rendered step against the CPU oracle.
    The oracle's voices of a picked bus are started like the engine's, set to the engine's reported position at the start of
    the step (every voice sits in sustain: the position is the whole state), and rendered up to the picked block.  Bit-exact
    or the run fails."""
    from oracle import zl_oracle as zo
    N = args.frames
    res = []
    for (b, k) in picks:
        osyn = zo.OracleSynth(1, voices_per_bus, args.fs, mode, max_sounds=voices_per_bus)
        for i in range(voices_per_bus):
            v = b * voices_per_bus + i
            L, R = kept[v]
            cid = osyn.register_clip(L, R, source_rate)
            clip = osyn.clips[cid]
            clip.lengthInBeats = P["length_in_beats"]; clip.lengthInSeconds = P["length_seconds"][v]
            clip.volumeAbsolute = P["volume_absolute"][v]; clip.pan = P["pan"][v]
            cmd = zo.clip_command(clip=cid, midiNote=P["note"][v], midiChannel=-2, startPlayback=1, looping=1, changeVolume=1, volume=P["velocity"][v])
            assert osyn.start_voice(0, i, cmd, 0) == 1
            r = reports_before[v]
            assert r.playing, f"voice {v} stopped playing"
            osyn.voices[i].sourceSamplePosition = r.source_sample_position
        ref, _ = osyn.render_batch(k + 1, N, clocks, want_reports=False)
        want = ref[0, :, k * N:(k + 1) * N]
        got = bus_np_rows[(b, k)]
        ok = bool(np.array_equal(want.view(np.int32), got.view(np.int32)))
        res.append({"bus": int(b), "block": int(k), "bit_exact": ok, "max_abs_diff": float(np.abs(want - got).max()), "peak": float(np.abs(want).max())})
    return res


def _host_cores():
    """(cores this process may run on, cores the scheduler lists): the affinity mask capped by the cgroup CPU quota (a GPU
    box hands a one-GPU job a share of its host, e.g. 16 of 256 cores: threads beyond the quota only take turns)."""
    try:
        listed = len(os.sched_getaffinity(0))
    except Exception:
        listed = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]              # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    usable = listed if quota is None else max(1, min(listed, int(quota + 0.5)))
    return usable, listed


def cpu_baseline(args, seed):
    """Oracle (-O3 -march=native build) on a bounded sample of the same workload on this box's host cores, buses partitioned
    over host threads (the reference runs one RT thread per JACK client, i.e. per channel).  Timed on ALL host cores the
    scene can use (min(usable cores, buses)) and, for comparison with round 1, on 16 threads and on one."""
    from oracle import zl_oracle as zo
    from libzl_amd.engine import synthetic_clocks
    cores, listed = _host_cores()
    V = 1024
    vpb = 8                                  # the reference's own voices per channel: 128 buses, so up to 128 threads can work
    B = V // vpb
    threads = max(1, min(cores, B, args.cpu_threads if args.cpu_threads > 0 else cores))
    loop = int(args.loop_seconds * args.fs)  # the GPU leg's loop length (2 s: 786 MB of sources on the host too)
    blocks = args.cpu_blocks
    P = scene_params(V, vpb, loop, args.fs, seed)
    osyn = zo.OracleSynth(B, vpb, args.fs, 0, max_sounds=V, fast=True)
    rng = np.random.default_rng(seed)
    for v in range(V):
        L = rng.random(loop, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
        R = rng.random(loop, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
        cid = osyn.register_clip(L, R, args.fs)
        clip = osyn.clips[cid]
        clip.lengthInBeats = P["length_in_beats"]
        clip.lengthInSeconds = P["length_seconds"][v]
        clip.volumeAbsolute = P["volume_absolute"][v]
        clip.pan = P["pan"][v]
    for v in range(V):
        bus, slot = divmod(v, vpb)
        cmd = zo.clip_command(clip=v, midiNote=60, midiChannel=bus - 2, startPlayback=1, looping=1, changeVolume=1, volume=P["velocity"][v])
        assert osyn.start_voice(bus, slot, cmd, 0) == 1
    clocks = synthetic_clocks(blocks, args.frames, args.fs)

    def timed(nthreads, seconds):
        osyn.render_batch(2, args.frames, clocks, threads=nthreads, want_reports=False)       # warm-up
        done, t0 = 0, time.perf_counter()
        while True:
            osyn.render_batch(blocks, args.frames, clocks, threads=nthreads, want_reports=False)
            done += blocks
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return V * done * args.frames / dt, done, dt
    # bounded sample: batches of `blocks` blocks until about args.cpu_seconds of wall time have passed (all cores), then
    # short samples on 16 threads and on one
    value, done, dt = timed(threads, args.cpu_seconds)
    v16 = timed(min(16, threads), min(3.0, args.cpu_seconds))[0] if threads > 16 else value
    single = timed(1, min(2.0, args.cpu_seconds))[0]

    # BASELINE configs[0] (the reference's own CPU-runnable case, SURVEY 8d cfg 1): ONE mono 44.1 kHz loop of 176 400
    # frames, 256-frame blocks, lengthInBeats 8 (clock-driven restart) -- one voice on one thread
    o1 = zo.OracleSynth(1, 1, 44100.0, 0, max_sounds=1, fast=True)
    L = np.random.default_rng(seed + 1).random(176400, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    cid = o1.register_clip(L, None, 44100.0)
    o1.lib.zlo_clip_set_length(C.byref(o1.clips[cid]), C.c_float(8.0), 120)
    o1.clips[cid].volumeAbsolute = 1.0; o1.clips[cid].pan = 0.0
    assert o1.start_voice(0, 0, zo.clip_command(clip=cid, midiNote=60, midiChannel=-2, startPlayback=1, looping=1, changeVolume=1, volume=1.0), 0) == 1
    c1 = synthetic_clocks(1400, 256, 44100.0)
    o1.render_batch(64, 256, c1, want_reports=False)
    t1, n1 = time.perf_counter(), 0
    while time.perf_counter() - t1 < min(2.0, args.cpu_seconds):
        o1.render_batch(1400, 256, c1, want_reports=False); n1 += 1400
    cfg1 = n1 * 256 / (time.perf_counter() - t1)
    return {
        "value": value, "unit": "voice-samples/s", "cores": threads, "kind": "port",
        "value_16_threads": v16, "single_thread_value": single, "host_cores_usable": cores, "host_cores_listed": listed,
        "config1_single_mono_loop_value": cfg1,
        "sample": f"{V} stereo voices as 128 buses x 8 (the reference's voices per channel; the GPU leg sums the same voices on 8 buses x 128 -- "
                  f"the per-voice arithmetic is identical, the bus write is 1.0 instead of 0.06 B per voice-sample) x {done} blocks x {args.frames} frames, "
                  f"{args.loop_seconds:g} s loops as on the GPU, ratio 1, faithful mode, oracle/zl_oracle.c built -O3 -march=native, {threads} threads = "
                  f"min(host cores, buses) (buses partitioned, one thread per JACK client as in the reference), {dt:.1f} s wall; {cores} host cores usable (affinity mask capped by the "
                  f"cgroup CPU quota; {listed} listed); "
                  f"config1_single_mono_loop_value: BASELINE configs[0] (one mono 44.1 kHz 4 s loop, 256-frame blocks, lengthInBeats 8) on one thread",
    }


def span_buses_leg(args, torch, dist, sharding, syn, dev, stream, clock_sets, rank, world, log):
    """The job seen as --buses buses that each SPAN all ranks (BASELINE configs[3]: one stereo bus over 8 GPUs): every rank's
    rendered bus is a partial bus, the partial buses are exchanged and summed -- the path's only coupling, SamplerSynth.cpp:134-140
    -- once with the mesh exchange (RCCL all-to-all over the point-to-point xGMI links, zlhip_bus_reduce_sum_scan in rank order on
    every rank's piece, gather on rank 0) and once with one plain RCCL reduce; the exchange of step i overlaps the rendering of
    step i + 1 (OverlappedBusReduce).  Per algorithm: K steps timed like the headline (barrier + synchronize both sides, max
    over ranks), the exchange alone (no rendering), and an output check on rank 0: the reduced rows of one more step against the
    rank-order sum of every rank's partial rows (gathered), bit for bit for the mesh exchange, within 1e-6 of the summed
    magnitudes for RCCL's own order.  (Every rank's partial rows are what the headline's output check holds against the oracle.)"""
    V, B, N, KB = args.voices, args.buses, args.frames, args.blocks_per_step
    steps = args.span_steps or args.steps
    nccl = args.dist_backend == "nccl"
    sptr = stream.cuda_stream
    bus_bytes = B * 2 * KB * N * 4
    rec = {"what": f"{B} buses spanning all {world} ranks ({V * world} voices, {V} per rank): partial buses of {bus_bytes / 1e6:.0f} MB per rank and step "
                   "exchanged and summed on rank 0, exchange of step i overlapped with the rendering of step i+1",
           "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "steps": steps, "warmup": args.warmup,
           "partial_bus_bytes_per_rank": bus_bytes}
    rng = np.random.default_rng(0xB05 + world)                        # the same rows on every rank
    picks = sorted((int(rng.integers(0, B)), int(rng.integers(0, KB))) for _ in range(3))

    def region_time(fn, after=None):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        if after is not None:
            after()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def host_exchange(algo, b):
        # rehearsal backends (gloo): the exchange runs on a host copy of the partial bus
        torch.cuda.synchronize()
        host = b.cpu()
        (sharding.reduce_bus_mesh if algo == "mesh" else sharding.reduce_bus)(host, dst=0)
        if rank == 0:
            b.copy_(host)
            syn.levels_scan_device(b.data_ptr(), KB, N, stream=sptr)

    def one(algo):
        r = {"algorithm": algo}
        make = lambda: torch.zeros((B, 2, KB * N), device=dev, dtype=torch.float32)
        if nccl:
            ov = sharding.OverlappedBusReduce(syn, make, dst=0, algorithm=algo)
            for i in range(max(1, args.warmup)):
                _, ov = ov.step_or_fall_back(KB, N, clock_sets[i % len(clock_sets)], stream=sptr, log=log)
            ov.flush(stream=sptr)
            r["algorithm"] = ov.algorithm                              # "reduce" if this RCCL build refused the mesh exchange's collectives
            dt = region_time(lambda i: ov.step(KB, N, clock_sets[i % len(clock_sets)], stream=sptr), after=lambda: ov.flush(stream=sptr))
            work = ov.bus[0]
            if ov.algorithm == "mesh":
                xfn = lambda i: sharding.exchange_bus_mesh(syn, work, KB, N, dst=0, scratch=ov.scratch[0])
            else:
                def xfn(i):
                    dist.reduce(work, dst=0, op=dist.ReduceOp.SUM)
                    if rank == 0:
                        syn.levels_scan_device(work.data_ptr(), KB, N, stream=sptr)
            xfn(0)
            dx = region_time(xfn)
        else:
            work = make()
            def full(i):
                syn.render_batch(KB, N, clock_sets[i % len(clock_sets)], bus_out_dev=work.data_ptr(), stream=sptr)
                host_exchange(algo, work)
            full(0)
            dt = region_time(full)
            dx = region_time(lambda i: host_exchange(algo, work))
        r["ms_per_step"] = dt / steps * 1e3
        r["value"] = float(V) * world * KB * N * steps / dt
        r["exchange_alone_ms_per_step"] = dx / steps * 1e3
        if r["algorithm"] == "mesh":
            r["bytes_per_link_per_step"] = {"all_to_all_each_direction": bus_bytes // world, "gather_into_rank_0": bus_bytes // world + (B * 2 * KB // world) * 8,
                                            "note": "every rank sends 1/N of its partial bus to every peer over that pair's own xGMI link, then its reduced 1/N piece (+ unit levels) to rank 0"}
        else:
            r["bytes_per_link_per_step"] = {"whole_partial_bus": bus_bytes, "note": "one RCCL reduce: schedule and order are RCCL's (a ring carries about the whole bus over every link of the ring)"}
        # ---- output check: one more step, exchange not overlapped
        chk = make()
        syn.render_batch(KB, N, clock_sets[-1], bus_out_dev=chk.data_ptr(), stream=sptr)
        syn.synchronize(); torch.cuda.synchronize()
        rows = torch.stack([chk[b, :, k * N:(k + 1) * N] for (b, k) in picks]).clone()     # this rank's partial rows
        if nccl:
            if r["algorithm"] == "mesh":
                sharding.exchange_bus_mesh(syn, chk, KB, N, dst=0)
            else:
                dist.reduce(chk, dst=0, op=dist.ReduceOp.SUM)
        else:
            host_exchange(algo, chk)
        torch.cuda.synchronize()
        rows_c = rows if nccl else rows.cpu()
        parts = [torch.empty_like(rows_c) for _ in range(world)] if rank == 0 else None
        if world > 1:
            dist.gather(rows_c, gather_list=parts, dst=0)
        else:
            parts = [rows_c]
        if rank == 0:
            got = torch.stack([chk[b, :, k * N:(k + 1) * N] for (b, k) in picks]).cpu().numpy()
            ps = [p.cpu().numpy() for p in parts]
            acc = np.zeros_like(ps[0])
            mag = np.zeros_like(ps[0])
            for p_ in ps:
                acc = acc + p_                                         # ((0 + p0) + p1) + ... : the kernel's and the oracle's grouped order
                mag = mag + np.abs(p_)
            exact = bool(np.array_equal(acc.view(np.int32), got.view(np.int32)))
            close = bool((np.abs(acc - got) <= 1e-6 * np.maximum(mag, 1.0)).all())
            r["output_check"] = {"rows": [[int(b), int(k)] for (b, k) in picks], "bit_exact_vs_rank_order_sum": exact, "within_1e-6_of_magnitude": close,
                                 "max_abs_diff": float(np.abs(acc - got).max()), "peak": float(np.abs(got).max()),
                                 "ok": exact if r["algorithm"] in ("mesh", "rank-order") else close}
        del chk
        return r

    for algo in ("mesh", "reduce"):
        try:
            rec[algo] = one(algo)
        except Exception as err:                                       # (a refusal is synchronous and the same on every rank)
            rec[algo] = {"algorithm": algo, "error": repr(err)[:400]}
            if log:
                log(f"span_buses leg '{algo}' failed: {err!r}")
    return rec


def kernel_source_digest():
    """SHA-256 over the engine's kernel sources: PMC traffic collected for one build is only quoted for the same sources."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libzl_amd", "csrc")
    # (what the kernels are compiled from; the host-side files -- engine, libzl-named layer, scheduler -- do not change a kernel's traffic)
    for f in ("zl_kernels.hip", "zl_kernels.h", "zl_plan.h", "zl_render.h", "zl_types.h"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
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
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(host cores, 128 buses))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reuse-check", action="store_true", help="skip the 10 s-loop (no source re-read inside a window: HBM only) measurement")
    ap.add_argument("--no-reuse-calls", type=int, default=12)
    ap.add_argument("--no-reuse-seconds", type=float, default=-1.0,
                    help="length of the HBM-only leg's sources; its plan windows are capped so that no launch is longer (default: one standard plan window, "
                         "12 s at the default shape; 0 = sources as long as a whole call, the headline's launch shape)")
    ap.add_argument("--no-repeats", action="store_true", help="skip the two extra timed regions that give value_per_gpu_repeats")
    ap.add_argument("--no-spot-check", action="store_true", help="skip the output check against the oracle after the timed region")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; the measured path) or gloo (rehearsal: exchange through host memory)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--span-buses", action="store_true",
                    help="N > 1 variant: make the spanning-bus exchange the HEADLINE value (the job has only --buses buses and each spans all ranks, "
                         "BASELINE configs[3]).  By default the headline of an N > 1 run is the bus-aligned partition (N x --buses buses, whole buses "
                         "per GPU: no data-path collective, SURVEY 8e) and the spanning-bus exchange is measured next to it (`span_buses` sub-record)")
    ap.add_argument("--reduce-algo", default="mesh", choices=["mesh", "reduce", "rank-order"],
                    help="--span-buses: mesh = all-to-all + rank-order sum kernel + gather over the xGMI mesh (default), reduce = one RCCL reduce")
    ap.add_argument("--no-span-leg", action="store_true", help="N > 1: skip the `span_buses` sub-record (the exchange of buses that span all ranks)")
    ap.add_argument("--span-steps", type=int, default=0, help="timed steps of each `span_buses` leg (default: --steps)")
    ap.add_argument("--span-timeout", type=float, default=240.0,
                    help="N > 1: seconds the `span_buses` legs may take in all; after that the line is printed with what is there and the ranks leave")
    ap.add_argument("--rehearse-collectives", action="store_true", help="rehearsal only: run the spanning-bus code path with one rank")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU, no engine: start the ranks, form the process group, push a small bus through the exchange's collectives and "
                         "print a line with n_gpus (CPU-tier check that `bench.py --gpus N` becomes N ranks by itself)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher.  Runs BEFORE torch is imported or
    HIP is touched in this process (a process that has initialised the GPU must never start or replace GPU programs): N fresh
    rank processes through `python -m torch.distributed.run` (one per GPU, rendezvous on 127.0.0.1), rank 0's one JSON line is
    forwarded to this process's stdout, the exit code is the children's."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write(f"bench: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}\n")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    elif p.returncode == 0:
        sys.stderr.write("bench: the ranks printed no result line\n")
        return 4
    return p.returncode


def launch_check(args, dist, rank, world):
    """--launch-check: the ranks exist, the group forms, and the spanning-bus exchange's collectives (all-to-all + gather of the
    mesh exchange, the plain reduce) carry a small bus on host tensors.  No engine, no GPU."""
    import torch
    from libzl_amd import sharding
    n = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(n)
    part = torch.full((2, 2, 16 * world), float(rank + 1), dtype=torch.float32)
    want = float(world * (world + 1) // 2)
    a = sharding.reduce_bus_mesh(part.clone(), dst=0)
    b = sharding.reduce_bus(part.clone(), dst=0)
    ok = bool(int(n.item()) == world and (rank != 0 or (bool((a == want).all()) and bool((b == want).all()))))
    return {"metric": "voice-samples/sec at 1024 voices x 256-frame blocks; % HBM roofline", "value": None, "unit": "voice-samples/s",
            "n_gpus": world, "launch_check": True, "ranks_counted": int(n.item()), "backend": dist.get_backend(),
            "exchange_ok": ok, "steps": 0, "warmup": 0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "data": "none (launch check: no engine ran)"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.rehearse_collectives:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    # stdout carries exactly ONE line, the JSON result: whatever native libraries print meanwhile (RCCL's version banner
    # at communicator creation, for one) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        args.gpus = world
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29562")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        line = launch_check(args, dist, rank, world)
        if rank == 0:
            os.dup2(real_stdout, 1)
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        if not line["exchange_ok"]:
            raise SystemExit(5)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU render path")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or args.rehearse_collectives
    if args.rehearse_collectives:
        os.environ["ZL_FORCE_COLLECTIVES"] = "1"                      # the exchange runs its collectives with one rank too
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    exchange = distributed and args.span_buses                    # a bus spans ranks: the only case with a data-path collective

    from libzl_amd import SamplerSynth
    from libzl_amd.engine import synthetic_clocks

    V, B, N, KB = args.voices, args.buses, args.frames, args.blocks_per_step
    vpb = V // B
    notes = tuple(int(x) for x in args.notes.split(","))
    source_rate = args.source_rate or args.fs
    loop_frames = int(args.loop_seconds * source_rate)
    mode = 4 if args.hermite else 0
    arena = (loop_frames + 16) * 8 * V + (1 << 20)
    syn = SamplerSynth(B, vpb, max_frames=N, max_batch_blocks=KB, max_sounds=V, mode=mode, playback_sample_rate=args.fs,
                       sound_arena_bytes=arena, voices_per_task=args.voices_per_task, plan_window_blocks=args.plan_window, device=local_rank)
    seed = 0x5A17 + 2 + 1000 * rank
    # the rows checked against the oracle after the timed region: 3 random (bus, block) of one step, drawn up front so that the
    # sources of those buses can be kept on the host while the scene is built
    pick_rng = np.random.default_rng(seed ^ 0xC0FFEE)
    picks = [] if (args.no_spot_check or args.beat_locked) else sorted((int(pick_rng.integers(0, B)), int(pick_rng.integers(0, KB))) for _ in range(3))
    P, kept = build_scene(syn, torch, dev, vpb, B, args.fs, loop_frames, seed, notes=notes, source_rate=source_rate, mono=args.mono,
                          beat_locked=args.beat_locked, keep_buses={b for b, _ in picks})
    syn.set_profiling(not os.environ.get('ZL_BENCH_NOPROF'))      # diagnostic switch: cost of the per-launch HIP events

    # the engine renders into a torch-owned device buffer (for --span-buses the exchange needs no copy)
    bus = torch.zeros((B, 2, KB * N), device=dev, dtype=torch.float32)
    # an explicit (non-default) torch stream: the engine launches on it and, for --span-buses, RCCL orders its exchange after
    # the work queued on torch's *current* stream -- so this stream is made current for every step below
    stream = torch.cuda.Stream(device=dev)
    sptr = stream.cuda_stream
    clock_sets = [synthetic_clocks(KB, N, args.fs, start_block=i * KB) for i in range(args.warmup + args.steps + 1)]

    from libzl_amd import sharding
    overlapped = None
    if exchange and args.dist_backend == "nccl":
        overlapped = sharding.OverlappedBusReduce(syn, lambda: torch.zeros((B, 2, KB * N), device=dev, dtype=torch.float32), dst=0,
                                                  algorithm=args.reduce_algo)

    fan = fan_params = None
    if args.fanout != "none":
        from libzl_amd import PassthroughParams
        fan = torch.zeros((B, 6, KB * N), device=dev, dtype=torch.float32)
        fan_params = [PassthroughParams(0.9, 0.5, 0.25, 0.1 * (b % 3 - 1), 0) for b in range(B)]     # every pair multiplied

    def step(i, timed, warm=False):
        nonlocal overlapped
        if exchange and args.dist_backend != "nccl":
            # rehearsal path: the exchange runs on a host copy of the partial bus
            syn.render_batch(KB, N, clock_sets[i], bus_out_dev=bus.data_ptr(), stream=sptr)
            torch.cuda.synchronize()
            host = bus.cpu()
            sharding.reduce_bus(host, dst=0)
            if rank == 0:
                bus.copy_(host)
                syn.levels_scan_device(bus.data_ptr(), KB, N, stream=sptr)
        elif exchange and warm:
            # a collective this RCCL build refuses fails in a warm-up step, synchronously and on every rank: the exchange is then
            # rebuilt on the plain reduce (OverlappedBusReduce.step_or_fall_back; tests/test_dist_gloo.py runs that path)
            _, overlapped = overlapped.step_or_fall_back(KB, N, clock_sets[i], stream=sptr, log=(lambda m: sys.stderr.write(f"bench: {m}\n")) if rank == 0 else None)
            args.reduce_algo = overlapped.algorithm
        elif exchange:
            overlapped.step(KB, N, clock_sets[i], stream=sptr)       # exchange of step i overlaps rendering of step i+1
        elif args.fanout == "fused":
            syn.render_batch(KB, N, clock_sets[i], bus_out_dev=bus.data_ptr(), stream=sptr, fan_params=fan_params, fan_out_dev=fan.data_ptr())
        else:
            # N = 1, and N > 1 with the bus-aligned partition: this rank's whole buses, mixed and metered locally
            syn.render_batch(KB, N, clock_sets[i], bus_out_dev=bus.data_ptr(), stream=sptr)
            if args.fanout == "separate":
                syn.passthrough(fan_params, bus.data_ptr(), fan.data_ptr(), KB * N, stream=sptr)

    torch.cuda.synchronize()
    torch.cuda.set_stream(stream)
    import gc
    gc.collect(); gc.disable()                                       # as timeit does: no collector pause of this harness inside a timed region
    # (collected here, in front of the warm-up steps, so that the GPU goes from them straight into the timed region)
    for i in range(args.warmup):
        step(i, False, warm=True)
    torch.cuda.synchronize()
    syn.profile_totals(reset=True)                                   # HIP-event sums start with the timed region
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # the steps are queued back to back: consecutive zlhip_render_batch calls pipeline (the planning of step i+1
    # overlaps the rendering of step i); the per-kernel HIP events of every step are read once, after the region
    host_ms = []                                                     # host time inside each step's calls (they return before the GPU is done)
    for i in range(args.steps):
        th = time.perf_counter()
        step(args.warmup + i, True)
        host_ms.append((time.perf_counter() - th) * 1e3)
    if overlapped is not None:
        overlapped.flush(stream=sptr)                                # the last exchanges + level scans are inside the timed region
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

    # ---- spread: the same K steps timed twice more (each region bracketed by a synchronisation like the first; `value` stays the
    #      first region's).  Not for --span-buses (the exchange pipeline would have to be flushed per region).
    repeats = []
    if not exchange and not args.no_repeats:
        for rep in range(2):
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + i, True)
            torch.cuda.synchronize()
            repeats.append(float(V) * KB * N * args.steps / (time.perf_counter() - tr0))

    # ---- output check (every rank, its own voices): one more step, untimed, rendered like the timed ones; 3 random
    #      (bus, block) rows of it against the CPU oracle, bit for bit.  A mismatch fails the run.
    check = None
    if picks:
        before = syn.voice_reports()                                 # state at the start of the checked step (waits for the engine)
        ci = args.warmup + args.steps
        syn.render_batch(KB, N, clock_sets[ci], bus_out_dev=bus.data_ptr(), stream=sptr)
        syn.synchronize(); torch.cuda.synchronize()
        rows = {(b, k): bus[b, :, k * N:(k + 1) * N].cpu().numpy() for (b, k) in picks}
        check = spot_check(syn, rows, before, P, kept, picks, args, clock_sets[ci], vpb, loop_frames, source_rate, mode)
        kept.clear()
        if not all(c["bit_exact"] for c in check):
            sys.stderr.write(f"bench: OUTPUT CHECK FAILED on rank {rank}: {check}\n")
            os.dup2(real_stdout, 1)
            raise SystemExit(3)

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

    # ---- the HBM-only figure: the same kernel, same launch shape, on sources long enough that no byte is re-read inside a launch
    #      (a plan window), so neither L2 nor the 256 MiB Infinity Cache can serve a source read.  N = 1 only, --no-reuse-calls
    #      calls after the timed region.
    no_reuse = None
    # The leg's launches are the engine's standard plan windows (512 Ki voice-frames at 1024 voices = 2048 blocks of 256 frames, four per
    # 8192-block call, as rounds 2-3 measured it) and its sources are a little longer than such a window (12 s), so no launch re-reads a
    # byte.  (A unit-ratio scene is otherwise planned in ONE window per call -- 43.7 s -- and would re-read a 12 s source inside it.)
    # --no-reuse-seconds S: S-second sources and launches of at most S seconds; 0: sources as long as a whole call (17.7 GB at the default
    # shape) and the headline's own one launch per call.  profiles/round4_noreuse_ab.txt: no systematic difference between the forms.
    std_window = max(1, min(16 << 20, max(2048 * 256, (2048 * 256 * 1024) // max(V, 1))) // N)
    if args.no_reuse_seconds < 0:
        nr_window = args.plan_window or min(std_window, KB)
        nr_seconds = float(int(nr_window * N / args.fs * 1.03) + 1)
    elif args.no_reuse_seconds > 0:
        nr_seconds = args.no_reuse_seconds
        nr_window = args.plan_window or min(std_window, max(1, int(nr_seconds * args.fs * 0.97) // N))
    else:
        nr_seconds, nr_window = float(int(KB * N / args.fs * 1.03) + 1), args.plan_window
    if rank == 0 and not distributed and not args.no_reuse_check and args.loop_seconds < 10.0:
        lf2 = int(nr_seconds * source_rate)
        syn2 = SamplerSynth(B, vpb, max_frames=N, max_batch_blocks=KB, max_sounds=V, mode=mode,
                            playback_sample_rate=args.fs, sound_arena_bytes=(lf2 + 16) * (4 if args.mono else 8) * V + (1 << 20),
                            voices_per_task=args.voices_per_task, plan_window_blocks=nr_window, device=local_rank)
        build_scene(syn2, torch, dev, vpb, B, args.fs, lf2, seed + 7, notes=notes, source_rate=source_rate, mono=args.mono)
        syn2.set_profiling(True)
        for i in range(2):
            syn2.render_batch(KB, N, clock_sets[i % len(clock_sets)], bus_out_dev=bus.data_ptr(), stream=sptr)
        syn2.profile_totals(reset=True)
        for i in range(args.no_reuse_calls):
            syn2.render_batch(KB, N, clock_sets[(2 + i) % len(clock_sets)], bus_out_dev=bus.data_ptr(), stream=sptr)
        t2, n2 = syn2.profile_totals()
        b2 = (t2.source_bytes + n2 * B * 2 * N * 4 * KB) / max(1, t2.render_launches)
        ms2 = t2.render_ms / max(1, t2.render_launches)
        g2 = b2 / (ms2 * 1e-3) / 1e9
        no_reuse = {"loop_seconds": nr_seconds, "plan_window_blocks": nr_window, "calls": n2, "launches": int(t2.render_launches), "achieved": g2, "frac": g2 / HBM_PEAK_GBS,
                    "avg_launch_ms": ms2, "algorithmic_bytes_per_launch": b2,
                    "value": float(V) * KB * N * n2 / (t2.total_ms * 1e-3) if t2.total_ms > 0 else None}
        syn2.close()

    eng_total, eng_arena = syn.memory_bytes()
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

    # HBM traffic of K2 per launch: PMC counters cannot be read inside this process, so `traffic` is quoted only from a PMC
    # collection of THIS build (same kernel-source digest) on THIS workload (profiles/*_pmc.json, scripts/publish_profile.py);
    # otherwise null
    traffic = traffic_src = None
    default_workload = (V, B, N, KB, args.loop_seconds, notes, args.hermite, source_rate, args.mono, args.beat_locked) == (1024, 8, 256, 8192, 2.0, (60, 60), False, args.fs, False, False) and args.fanout == "none"
    if default_workload:
        digest = kernel_source_digest()
        import glob
        for pmc_file in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc.json")), reverse=True):
            try:
                pj = json.load(open(pmc_file))
            except Exception:
                continue
            if pj.get("kernel_source_digest") == digest and "traffic_over_algorithmic" in pj:
                traffic = pj["traffic_over_algorithmic"] * k2_bytes
                traffic_src = f"{os.path.relpath(pmc_file, ROOT)}: rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE per launch; the factor 2 calibrated on the no-reuse workload, where every source byte must come from HBM) / algorithmic bytes, collected for this kernel build (digest {digest})"
                break

    out = None
    if rank == 0:
        out = {
            "metric": "voice-samples/sec at 1024 voices x 256-frame blocks; % HBM roofline",
            "value": value, "unit": "voice-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "value_per_gpu_repeats": repeats, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{V} looping {'mono' if args.mono else 'stereo'} voices per GPU on {B} buses x {vpb}, {N}-frame blocks, {KB} blocks per step ({launches} K2 launches), "
                            + (f"fs=sr={args.fs:.0f} (ratio 1)" if notes[0] == notes[1] and source_rate == args.fs else
                               f"fs={args.fs:.0f}, sources at {source_rate:.0f}, MIDI notes {notes[0]}..{notes[1]} around root 60") +
                            f", {'4-tap Hermite' if args.hermite else 'linear'} interp, faithful mode, distinct {args.loop_seconds:g} s sources "
                            f"({arena / 1e6:.0f} MB), bus int peaks + sums of squares per block" + ({"none": "", "fused": ", JackPassthrough fan-out fused into the bus write", "separate": ", JackPassthrough fan-out as a separate pass"}[args.fanout])
                            + ((f", {B} buses spanning all {world} ranks: partial buses exchanged over RCCL ({args.reduce_algo}) and summed in rank order, overlapped with the next step" if exchange
                                else f", bus-aligned partition: {world * B} buses, {B} whole buses per GPU, no data-path collective") if distributed else ""),
                "voices_per_gpu": V, "buses": B, "frames_per_block": N, "blocks_per_step": KB,
                "parallelism": (f"voices sharded x{world}, buses span ranks" if exchange else f"whole buses per GPU x{world}"),
            },
            "roofline": {
                "bound": "hbm", "kernel": "zl_k2_render", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                # the timed workload re-reads its 2 s sources inside a plan window: most of those re-reads are Infinity-Cache hits.
                # The HBM-only figures of the same kernel (no_reuse_variant, measured in this run) are first-class:
                "cache_assisted": bool(args.loop_seconds * args.fs < KB * N),
                "achieved_hbm_no_reuse": no_reuse["achieved"] if no_reuse else None,
                "frac_hbm_no_reuse": no_reuse["frac"] if no_reuse else None,
                "no_reuse_variant": no_reuse,
                "device_copy_GBs": copy_gbs, "frac_of_device_copy": (achieved / copy_gbs) if copy_gbs else None,
                "frac_no_reuse_of_device_copy": (no_reuse["achieved"] / copy_gbs) if (copy_gbs and no_reuse) else None,
                "algorithmic_bytes_per_launch": k2_bytes, "avg_launch_ms": k2_avg_ms, "launches_per_step": launches,
                "bytes_per_voice_sample": k2_bytes_step / (V * KB * N),
                "other_ms_per_step": {"planning_not_hidden (K0+K1+K1c of the first window; overlaps the previous step)": float(np.mean(plan_ms)),
                                      "K3 finalize + reports + launch gaps": float(np.mean(fin_ms))},
                "host_dispatch_ms_per_step": {"mean": float(np.mean(host_ms)), "max": float(np.max(host_ms)), "argmax": int(np.argmax(host_ms))},
                "state_and_levels_bytes_per_step": state_bytes, "slow_blocks": int(slow),
                "engine_device_bytes": {"total": eng_total, "source_arena": eng_arena, "beyond_sources": eng_total - eng_arena},
                "note": "achieved / frac are algorithmic bytes over the K2 launch time of the timed (BASELINE) workload, whose 2 s sources are re-read "
                        "every 375 blocks: inside a 2048-block plan window about 80 % of the source reads are re-reads served by the 256 MiB Infinity "
                        "Cache (bus-major launch order keeps one bus's 98 MB of sources hot), so that figure exceeds what HBM alone delivers on this "
                        "chip (~6.3 TB/s).  achieved_hbm_no_reuse / frac_hbm_no_reuse are the same kernel and launch shape on sources as long as a whole call "
                        "(no_reuse_variant.loop_seconds), where every source byte comes from HBM: that is the figure to hold against the HBM roofline (DESIGN.md section 4)",
            },
            "output_check": {"rows_vs_oracle": check, "what": "3 random (bus, block) rows of one extra step rendered after the timed region, bit-exact against oracle/zl_oracle.c"} if check is not None else None,
        }
        if not args.no_cpu_baseline:
            # the CPU leg is timed on rank 0 at N = 1 only (a reported baseline of the same workload, not part of the scaling curve)
            out["cpu_baseline"] = cpu_baseline(args, seed) if world == 1 else None

    def emit():
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    # ---- N > 1: the spanning-bus exchange next to the collective-free headline (same engine, same voices).  A watchdog bounds
    #      it: should a collective never complete on this node, rank 0 prints the line with what is there and every rank leaves.
    if distributed and not exchange and not args.no_span_leg and args.fanout == "none":
        import threading

        def give_up():
            if rank == 0:
                out["span_buses"] = {"error": f"not finished after {args.span_timeout:.0f} s (--span-timeout): a collective did not complete"}
                emit()
            else:
                time.sleep(2.0)
            os._exit(0)
        dog = threading.Timer(args.span_timeout, give_up)
        dog.daemon = True
        dog.start()
        span = span_buses_leg(args, torch, dist, sharding, syn, dev, stream, clock_sets, rank, world,
                              (lambda m: sys.stderr.write(f"bench: {m}\n")) if rank == 0 else None)
        dog.cancel()
        if rank == 0:
            out["span_buses"] = span
            bad = [a for a in ("mesh", "reduce") if isinstance(span.get(a), dict) and span[a].get("output_check") and not span[a]["output_check"]["ok"]]
            if bad:
                sys.stderr.write(f"bench: SPAN-BUSES OUTPUT CHECK FAILED: {[span[a]['output_check'] for a in bad]}\n")
                raise SystemExit(3)
    if rank == 0:
        emit()
    syn.close()
    if distributed:
        dist.barrier()                       # rank 0 is still timing the CPU baseline: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
