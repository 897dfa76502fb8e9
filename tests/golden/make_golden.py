#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the independent numpy restatement (oracle/np_restatement.py).

The reference ships no golden vectors and cannot be built or imported here, so these fixtures pin the
build's OWN specification: inputs + the outputs of the pure-numpy restatement, against which the C
oracle (CPU tier) and the HIP engine (GPU tier) are compared bit for bit.  Run from the repo root:
    python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_restatement as nr  # noqa: E402

f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def clocks_for(k, nframes, fs, bpm=120, playhead_fn=None, block0=0):
    period = int(round(1e6 * nframes / fs))
    sub = ((60000000000) // (bpm * 96)) // 1000
    c = nr.Clock((block0 + k) * period, (block0 + k + 1) * period, 0, 0, sub)
    if playhead_fn:
        c.playhead, c.playhead_usecs = playhead_fn(k, period, sub)
    return c


def run_scene(name, *, B, VPB, fs, mode, nframes, nblocks, sounds, clips, events, bpm=120, playhead=None, block0=0):
    """sounds: list of (L, R|None, sr); clips: list of dict of Clip field overrides; events: {block: [cmd dict + 'tick']}"""
    syn = nr.Synth(B, VPB, fs, mode)
    for (L, R, sr), cf in zip(sounds, clips):
        i = syn.register(L, R, sr)
        c = syn.clips[i]
        if "set_length" in cf:
            c.set_length(*cf["set_length"])
        for k, v in cf.items():
            if k == "set_length":
                continue
            if k == "adsr":
                c.adsr = tuple(f32(x) for x in v)
            elif k == "root_note":
                c.root_note = int(v)
            elif k == "slice_pos":
                c.slice_pos = [float(x) for x in v]
            else:
                setattr(c, k, f32(v))
    busL = np.zeros((B, nblocks * nframes), dtype=np.float32)
    busR = np.zeros((B, nblocks * nframes), dtype=np.float32)
    trace = np.full((nblocks, B * VPB, nframes), -1, dtype=np.int32)
    last_reports = {}
    clock_rows = []
    for k in range(nblocks):
        for ev in events.get(k, []):
            ev = dict(ev)
            tick = ev.pop("tick", 0)
            kind = ev.pop("kind", "cmd")                            # "cmd" (SamplerSynth::handleClipCommand) | "start" / "update" / "stopv" (one voice) | "enable"
            if kind == "enable":
                syn.enabled[ev["bus"]] = bool(ev["on"])
                continue
            if kind == "stopv":
                syn.stop_voice(ev["bus"], ev["slot"], bool(ev["tail"]))
                continue
            bus, slot = ev.pop("bus", None), ev.pop("slot", None)
            cmd = nr.Command(**{kk: (f32(vv) if kk in ("volume", "pitch_change", "speed_ratio", "gain_db") else vv) for kk, vv in ev.items()})
            if kind == "start":
                syn.start_voice(bus, slot, cmd, tick)
            elif kind == "update":
                syn.update_voice(bus, slot, cmd)
            else:
                syn.handle(cmd, tick)
        clk = clocks_for(k, nframes, fs, bpm, playhead, block0)
        clock_rows.append([clk.current_usecs, clk.next_usecs, clk.playhead, clk.playhead_usecs, clk.subbeat_usecs])
        L, R, reports = syn.process(nframes, clk)
        busL[:, k * nframes:(k + 1) * nframes] = L
        busR[:, k * nframes:(k + 1) * nframes] = R
        for (b, i), (valid, gain, prog, tr) in reports.items():
            trace[k, b * VPB + i] = tr
        last_reports = reports
    V = B * VPB
    rep = np.zeros((V, 3), dtype=np.float64)      # valid, gain, progress of the last block
    state = np.zeros((V, 2), dtype=np.float64)    # isPlaying, sourceSamplePosition
    for b in range(B):
        for i, v in enumerate(syn.voices[b]):
            state[b * VPB + i] = (1.0 if v.is_playing else 0.0, float(v.P))
            if (b, i) in last_reports:
                valid, gain, prog, _ = last_reports[(b, i)]
                rep[b * VPB + i] = (1.0 if valid else 0.0, float(gain), float(prog))
    clip_fields = []
    for c in syn.clips:
        clip_fields.append(dict(start_sec=float(c.start_sec), length_sec=float(c.length_sec), length_beats=float(c.length_beats),
                                volume_abs=float(c.volume_abs), pan=float(c.pan), duration=float(c.duration), root_note=c.root_note,
                                slice_pos=[float(x) for x in c.slice_pos], adsr=[float(x) for x in c.adsr]))
    meta = dict(name=name, B=B, VPB=VPB, fs=fs, mode=mode, nframes=nframes, nblocks=nblocks, bpm=bpm,
                events={str(k): v for k, v in events.items()}, clips=clip_fields,
                sample_rates=[s[2] for s in sounds], stereo=[s[1] is not None for s in sounds])
    arrays = dict(meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), busL=busL, busR=busR, trace=trace,
                  reports=rep, state=state, clocks=np.array(clock_rows, dtype=np.uint64))
    for i, (L, R, sr) in enumerate(sounds):
        arrays[f"snd{i}_L"] = np.asarray(L, dtype=np.float32)
        if R is not None:
            arrays[f"snd{i}_R"] = np.asarray(R, dtype=np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: peak |x| = {max(np.abs(busL).max(), np.abs(busR).max()):.4f}  -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} bytes)")


def config1_full_shape():
    """BASELINE configs[0] at its stated shape (SURVEY 8d cfg 1): ONE mono 44.1 kHz source of 176 400 frames (4.0 s = 8 beats
    at 120 bpm), 256-frame blocks, looping, volumeAbsolute 1, pan 0, default ADSR -- played twice: on bus 0 with
    lengthInBeats = 8 (an integer number of beats: the loop restarts against the JACK clock, current_usecs = k * 5805,
    SamplerSynthVoice.cpp:227-241) and on bus 1 with lengthInBeats = 7.5 (fractional: the sample-space wrap of :243-246).
    1400 blocks = two passes of the loop.  The fixture keeps the inputs (source quantised to 16 bits so that it stores as
    int16), the expected audio / source indices of the blocks around every loop restart, and SHA-256 digests of the WHOLE
    expected bus and index trace."""
    import hashlib
    rng = np.random.default_rng(0x5A17 + 0)
    q = rng.integers(-32768, 32768, 176400).astype(np.int16)
    L = q.astype(np.float32) / f32(32768.0)
    B, VPB, fs, N, K = 2, 1, 44100.0, 256, 1400
    syn = nr.Synth(B, VPB, fs, 0)
    for beats in (8.0, 7.5):
        i = syn.register(L, None, 44100.0)
        c = syn.clips[i]
        c.set_length(beats, 120)
        c.volume_abs = f32(1.0); c.pan = f32(0.0)
    for i, ch in ((0, -2), (1, -1)):
        syn.handle(nr.Command(clip=i, midi_channel=ch, midi_note=60, start=True, stop=True, looping=True, change_volume=True, volume=f32(1.0)), 0)
    bus = np.zeros((B, 2, K * N), dtype=np.float32)
    trace = np.full((K, B * VPB, N), -1, dtype=np.int32)
    clocks = []
    for k in range(K):
        clk = clocks_for(k, N, fs, 120)
        clocks.append([clk.current_usecs, clk.next_usecs, clk.playhead, clk.playhead_usecs, clk.subbeat_usecs])
        Lo, Ro, reports = syn.process(N, clk)
        bus[:, 0, k * N:(k + 1) * N] = Lo; bus[:, 1, k * N:(k + 1) * N] = Ro
        for (b, i), (valid, gain, prog, tr) in reports.items():
            trace[k, b * VPB + i] = tr
    # blocks in which a voice's source index steps back = loop restarts; keep them and their neighbours, plus both ends
    keep = {0, 1, 2, K - 2, K - 1}
    restarts = {}
    for v in range(B * VPB):
        flat = trace[:, v, :].reshape(-1)
        for idx in np.nonzero(np.diff(flat) < 0)[0]:
            k = int((idx + 1) // N)
            restarts.setdefault(v, []).append((k, int((idx + 1) % N)))
            keep.update(range(max(0, k - 2), min(K, k + 3)))
    keep = sorted(keep)
    assert all(len(r) == 2 for r in restarts.values()), restarts
    state = np.array([[1.0 if syn.voices[b][0].is_playing else 0.0, float(syn.voices[b][0].P)] for b in range(B)])
    meta = dict(name="c1_config1_shape", B=B, VPB=VPB, fs=fs, nframes=N, nblocks=K, bpm=120, beats=[8.0, 7.5],
                restarts={str(v): r for v, r in restarts.items()},
                bus_sha256=hashlib.sha256(bus.tobytes()).hexdigest(), trace_sha256=hashlib.sha256(trace.tobytes()).hexdigest())
    path = os.path.join(HERE, "c1_config1_shape.npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), source_q15=q,
                        keep=np.array(keep, dtype=np.int32),
                        bus_keep=np.stack([bus[:, :, k * N:(k + 1) * N] for k in keep]), trace_keep=trace[keep],
                        state=state, clocks=np.array(clocks, dtype=np.uint64)[[0, 1, K - 1]])
    print(f"c1_config1_shape: restarts {restarts}, {len(keep)} blocks kept -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} bytes)")


CMD_ORDER = ["clip", "midi_note", "midi_channel", "start", "stop", "change_slice", "slice", "change_looping", "looping", "change_pitch",
             "pitch_change", "change_speed", "speed_ratio", "change_gain_db", "gain_db", "change_volume", "volume"]     # ClipCommand.h:13-32
OP_KINDS = ["schedule", "start", "stop", "bpm", "qstart", "qstop", "tick"]


def scheduler_golden():
    """s1_scheduler.npz: a seeded session of SyncTimer calls (scheduleClipCommand with delays, start / stop / setBpm,
    queueClipToStart/Stop, timer ticks) through the numpy twin of the step ring (np_restatement.SyncTimerModel): every command
    dispatched per JACK cycle with all its fields and its tick, and the clock triple the voices read in that cycle."""
    rng = np.random.default_rng(0x5C4ED)
    N, fs, t0, ncycles = 256, 48000.0, 7_000_003, 700
    per = int(round(1e6 * N / fs))
    m = nr.SyncTimerModel()
    m.set_latency(N, fs)

    def rand_cmd():
        k = int(rng.integers(0, 5))
        c = nr.Command(clip=int(rng.integers(0, 4)), midi_channel=int(rng.integers(-2, 3)), midi_note=int(rng.choice([60, 60, 60, 64])))
        if k == 0:
            c.start = True; c.change_volume = True; c.volume = f32(1.0); c.looping = bool(rng.integers(0, 2)); c.stop = c.looping
        elif k == 1:
            c.stop = True
        elif k == 2:
            c.change_volume = True; c.volume = f32(rng.uniform(0, 1))
        elif k == 3:
            c.change_slice = True; c.slice = int(rng.integers(0, 4)); c.start = True; c.change_looping = True; c.looping = bool(rng.integers(0, 2))
        else:
            c.change_pitch = True; c.pitch_change = f32(rng.uniform(-2, 2)); c.change_gain_db = bool(rng.integers(0, 2)); c.gain_db = f32(-6.0)
            c.change_speed = bool(rng.integers(0, 2)); c.speed_ratio = f32(0.75)
        return c

    def row(c):
        return [float(getattr(c, f)) for f in CMD_ORDER]

    ops, disp, disp_cycle, clocks, running = [], [], [], [], []
    for k in range(ncycles):
        for _ in range(int(rng.integers(1, 5)) if rng.random() < 0.35 else 0):
            a = int(rng.integers(0, 14))
            if a < 7:
                c, d = rand_cmd(), int(rng.choice([0, 0, 0, 0, 1, 3, 24, 96, 400]))
                m.schedule(c, d); ops.append([k, 0, 0, d] + row(c))
            elif a == 7:
                b = int(rng.choice([60, 90, 120, 174, 230, 45])); m.start(b); ops.append([k, 1, b, 0] + [0.0] * 17)
            elif a == 8:
                m.stop(); ops.append([k, 2, 0, 0] + [0.0] * 17)
            elif a == 9:
                b = int(rng.choice([50, 100, 120, 140, 250])); m.set_bpm(b); ops.append([k, 3, b, 0] + [0.0] * 17)
            elif a in (10, 11):
                cl, ch = int(rng.integers(0, 4)), int(rng.integers(-2, 3)); m.queue_start(cl, ch); ops.append([k, 4, cl, ch] + [0.0] * 17)
            elif a == 12:
                cl, ch = int(rng.integers(0, 4)), int(rng.integers(-2, 3)); m.queue_stop(cl, ch); ops.append([k, 5, cl, ch] + [0.0] * 17)
            else:
                m.timer_callback(); ops.append([k, 6, 0, 0] + [0.0] * 17)
        cu, nx = t0 + k * per, t0 + (k + 1) * per
        for c, tick in m.process(N, cu, nx):
            disp.append(row(c) + [float(tick)]); disp_cycle.append(k)
        ck = m.clock(cu, nx)
        clocks.append([ck.playhead, ck.playhead_usecs, ck.subbeat_usecs])
        running.append(0 if m.paused else 1)
        if not m.paused:
            m.timer_callback()               # the timer thread's tick between two cycles, as libzl_hotpath_cycle models it
    meta = dict(name="s1_scheduler", N=N, fs=fs, t0=t0, ncycles=ncycles, cmd_order=CMD_ORDER, op_kinds=OP_KINDS)
    path = os.path.join(HERE, "s1_scheduler.npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), ops=np.array(ops, dtype=np.float64),
                        dispatch=np.array(disp, dtype=np.float64).reshape(-1, 18), dispatch_cycle=np.array(disp_cycle, dtype=np.int64),
                        clocks=np.array(clocks, dtype=np.uint64), running=np.array(running, dtype=np.uint8))
    print(f"s1_scheduler: {len(ops)} calls, {len(disp)} commands dispatched over {ncycles} cycles -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} bytes)")


def moving_playhead_goldens():
    """g4b: beat-locked loops against a MOVING SyncTimer playhead (the other goldens freeze it at (0, 0)).  The playhead and
    its time come from the numpy twin of SyncTimer itself: a timer started at time 0 with 200 bpm that has been running for
    12 000 JACK cycles when the scene begins (playhead > 10 000).  Commands carry the tick SyncTimer dispatches them with (the
    playhead before the cycle's first step); one carries a tick 200 ticks in the past, so that nextLoopTick lies BEHIND the
    playhead when the voice first renders: the unsigned wrap of SamplerSynthVoice.cpp:180-181,236-237 and a restart in three
    consecutive frames."""
    rng = np.random.default_rng(0x64B)

    def src(n, stereo=True):
        L = rng.uniform(-1, 1, n).astype(np.float32)
        return (L, rng.uniform(-1, 1, n).astype(np.float32) if stereo else None)

    play = lambda clip, ch=-2, loop=True, note=60, vol=1.0, **kw: dict(clip=clip, midi_channel=ch, midi_note=note, start=True, stop=loop,
                                                                      looping=loop, change_volume=True, volume=vol, **kw)
    N, fs, bpm, block0, nblocks = 128, 48000.0, 200, 12_000, 260
    per = int(round(1e6 * N / fs))
    m = nr.SyncTimerModel()
    m.set_latency(N, fs)
    m.start(bpm)
    rows = []
    for k in range(block0 + nblocks):
        m.process(N, k * per, (k + 1) * per)
        ck = m.clock(k * per, (k + 1) * per)
        rows.append((ck.playhead, ck.playhead_usecs))
        m.timer_callback()
    tick_of = lambda k: rows[block0 + k - 1][0]          # the playhead SyncTimer dispatches the cycle's first step with
    a, b, c, d = src(2600), src(2200, stereo=False), src(3100), src(2000)
    run_scene("g4b_beat_locked_moving_playhead", B=1, VPB=4, fs=fs, mode=0, nframes=N, nblocks=nblocks, bpm=bpm, block0=block0,
              sounds=[(a[0], a[1], 48000.0), (b[0], None, 44100.0), (c[0], c[1], 48000.0), (d[0], d[1], 48000.0)],
              clips=[dict(length_beats=1.0, length_sec=0.04, volume_abs=0.7, pan=0.2), dict(length_beats=1.0, length_sec=0.035, volume_abs=0.9, pan=-0.4),
                     dict(length_beats=1.0, length_sec=0.05, volume_abs=0.8, pan=0.0), dict(length_beats=2.0, length_sec=0.03, volume_abs=0.6, pan=0.5)],
              events={0: [dict(play(0, note=60, vol=0.9), tick=tick_of(0)), dict(play(3, note=64, vol=0.5), tick=tick_of(0))],
                      7: [dict(play(1, note=62, vol=0.8), tick=tick_of(7))],
                      20: [dict(play(2, note=57, vol=0.7), tick=tick_of(20) - 200)]},
              playhead=lambda k, period, sub: rows[block0 + k])


def command_patch_goldens():
    """g10: commands that PATCH playing voices (SamplerSynth.cpp:216-229 -> SamplerSynthVoice::setCurrentCommand, :58-100), which
    the other goldens only do with changeVolume: changeLooping in both directions (a loop that becomes a one-shot runs into its
    release tail and ends; a one-shot that becomes a loop wraps instead of ending), the stored-only fields (pitch / speed / gain:
    no audible effect in this version of the reference), equivalence by (note, channel) with two voices of ONE clip -- a patch and a
    stop addressed to one note leave the other alone --, equivalence by slice, a patch nobody matches, and slice tables that went
    through setSlices' shrink and grow branches (ClipAudioSource.cpp:495-528) with a slice index beyond the table (:261-277: the
    clip's own start / stop)."""
    rng = np.random.default_rng(0x610)

    def src(n, stereo=True):
        L = rng.uniform(-1, 1, n).astype(np.float32)
        return (L, rng.uniform(-1, 1, n).astype(np.float32) if stereo else None)

    play = lambda clip, ch=-2, loop=True, note=60, vol=1.0, **kw: dict(clip=clip, midi_channel=ch, midi_note=note, start=True, stop=loop,
                                                                      looping=loop, change_volume=True, volume=vol, **kw)
    a, b, c = src(2900), src(3300), src(2500, stereo=False)
    tbl = nr.Clip()
    for n in (16, 4, 6):                                           # ctor's 16 (:204), shrink, grow: [0, 1/16, 2/16, 3/16, 0.59375, 1.0]
        tbl.set_slices(n)
    events = {
        0: [play(0, note=60, vol=0.8), play(0, note=64, vol=0.7), play(1, loop=False, note=62, vol=0.9),
            play(2, note=60, vol=0.6, change_slice=True, slice=2),            # a slice voice (equivalence by slice from now on)
            play(2, loop=False, note=67, vol=0.5, change_slice=True, slice=9)],   # beyond the table: the clip's own start and stop
        3: [dict(clip=0, midi_channel=-2, midi_note=64, change_looping=True, looping=False, change_pitch=True, pitch_change=0.25,
                 change_speed=True, speed_ratio=1.5, change_gain_db=True, gain_db=-6.0)],       # note 64 only: loop -> one-shot
        5: [dict(clip=1, midi_channel=-2, midi_note=62, change_looping=True, looping=True)],     # one-shot -> loop, before its end
        7: [dict(clip=2, midi_channel=-2, midi_note=1, change_slice=True, slice=2, change_volume=True, volume=0.25),   # by slice (the note is not compared)
            dict(clip=2, midi_channel=-2, midi_note=60, change_slice=True, slice=3, change_volume=True, volume=1.0),   # no voice plays slice 3
            dict(clip=1, midi_channel=-1, midi_note=62, change_volume=True, volume=0.0)],                               # other channel: nobody
        11: [dict(clip=0, midi_channel=-2, midi_note=60, stop=True)],                                                    # note 60 only
        22: [dict(clip=1, midi_channel=-2, midi_note=62, change_looping=True, looping=False, change_volume=True, volume=0.5)],   # (it has wrapped once by now)
    }
    clips = [dict(length_beats=0.37, length_sec=0.03, volume_abs=0.8, pan=-0.3, adsr=(0.0, 0.1, 1.0, 0.004)),
             dict(length_beats=0.41, length_sec=0.05, volume_abs=0.9, pan=0.2, adsr=(0.002, 0.003, 0.7, 0.006)),
             dict(length_beats=0.29, length_sec=0.04, volume_abs=0.7, pan=0.5, slice_pos=tbl.slice_pos)]
    for mode, nm in ((0, "g10_command_patches"), (4, "g10_command_patches_hermite")):
        run_scene(nm, B=1, VPB=6, fs=48000.0, mode=mode, nframes=128, nblocks=40,
                  sounds=[(a[0], a[1], 48000.0), (b[0], b[1], 44100.0), (c[0], None, 48000.0)], clips=clips, events=events)


def voice_level_goldens():
    """g11: the voice-level calls and SamplerSynth::setChannelEnabled, which no other golden has -- setCurrentCommand on one playing voice
    with every patch, among them startPlayback = restart at the start of the voice's slice and changeSlice + startPlayback = jump to
    another slice (SamplerSynthVoice.cpp:58-100); stopNote with a tail and without (:146-169); a channel switched off while a loop, an
    envelope in its attack and a beat-locked loop play on it, a voice started on it meanwhile, a stop that releases an envelope which
    stands still, and everything going on when the channel comes back (SamplerSynth.cpp:116-123,343-351)."""
    rng = np.random.default_rng(0x611)

    def src(n, stereo=True):
        L = rng.uniform(-1, 1, n).astype(np.float32)
        return (L, rng.uniform(-1, 1, n).astype(np.float32) if stereo else None)

    play = lambda clip, ch=-2, loop=True, note=60, vol=1.0, **kw: dict(clip=clip, midi_channel=ch, midi_note=note, start=True, stop=loop,
                                                                      looping=loop, change_volume=True, volume=vol, **kw)
    a, b, c = src(3100), src(2700, stereo=False), src(3500)
    tbl = nr.Clip()
    for n in (16, 4):
        tbl.set_slices(n)
    clips = [dict(length_beats=0.37, length_sec=0.05, volume_abs=0.8, pan=-0.2, adsr=(0.0, 0.1, 1.0, 0.004), slice_pos=tbl.slice_pos),
             dict(length_beats=1.0, length_sec=0.04, volume_abs=0.9, pan=0.3, adsr=(0.0, 0.1, 1.0, 0.006)),            # beat-locked
             dict(length_beats=0.29, length_sec=0.06, volume_abs=0.7, pan=0.0, adsr=(0.012, 0.01, 0.6, 0.006))]
    ev = {
        0: [dict(play(0, ch=-2, note=60, vol=0.8), kind="start", bus=0, slot=0), dict(play(0, ch=-2, note=64, vol=0.7, change_slice=True, slice=1), kind="start", bus=0, slot=2),
            dict(play(1, ch=-1, note=62, vol=0.7), kind="cmd"), dict(play(2, ch=-1, loop=False, note=57, vol=0.9), kind="cmd"), dict(play(2, ch=0, note=66, vol=0.5), kind="cmd")],
        4: [dict(kind="update", bus=0, slot=0, clip=0, midi_channel=-2, midi_note=60, start=True),                                      # restart from the top
            dict(kind="update", bus=0, slot=1, clip=0, midi_channel=-2, midi_note=60, change_volume=True, volume=0.1)],                   # slot 1 does not play
        5: [dict(kind="enable", bus=1, on=False)],
        7: [dict(kind="update", bus=0, slot=2, clip=0, midi_channel=-2, midi_note=64, change_slice=True, slice=3, start=True, change_volume=True, volume=0.4)],
        8: [dict(play(0, ch=-1, note=67, vol=0.5), kind="cmd"), dict(kind="cmd", clip=1, midi_channel=-1, midi_note=62, change_volume=True, volume=0.2)],
        11: [dict(kind="stopv", bus=2, slot=0, tail=False), dict(kind="stopv", bus=0, slot=3, tail=True)],                               # hard stop mid-loop; slot 3 does not play
        14: [dict(kind="cmd", clip=2, midi_channel=-1, midi_note=57, stop=True)],                                                         # releases an envelope that stands still
        16: [dict(kind="stopv", bus=0, slot=0, tail=True), dict(play(1, ch=0, note=55, vol=0.6), kind="start", bus=2, slot=0)],           # the freed slot is taken again
        30: [dict(kind="enable", bus=1, on=True)],
        40: [dict(kind="enable", bus=0, on=False), dict(kind="enable", bus=2, on=False)],
        44: [dict(kind="enable", bus=0, on=True), dict(kind="update", bus=2, slot=0, clip=1, midi_channel=0, midi_note=55, change_looping=True, looping=False)],
        50: [dict(kind="enable", bus=2, on=True)],
    }
    for mode, nm in ((0, "g11_voice_level_and_channels"), (4, "g11_voice_level_and_channels_hermite")):
        run_scene(nm, B=3, VPB=4, fs=48000.0, mode=mode, nframes=128, nblocks=60, bpm=200,
                  sounds=[(a[0], a[1], 48000.0), (b[0], None, 44100.0), (c[0], c[1], 48000.0)], clips=clips, events=ev)


def main():
    if "--voice-level" in sys.argv:
        return voice_level_goldens()
    if "--config1" in sys.argv:
        return config1_full_shape()
    if "--patches" in sys.argv:
        return command_patch_goldens()
    if "--scheduler" in sys.argv:
        return scheduler_golden()
    if "--moving-playhead" in sys.argv:
        return moving_playhead_goldens()
    rng = np.random.default_rng(0x5A17)

    def src(n, stereo=True):
        L = rng.uniform(-1, 1, n).astype(np.float32)
        return (L, rng.uniform(-1, 1, n).astype(np.float32) if stereo else None)

    play = lambda clip, ch=-2, loop=True, note=60, vol=1.0, **kw: dict(clip=clip, midi_channel=ch, midi_note=note, start=True, stop=loop,
                                                                      looping=loop, change_volume=True, volume=vol, **kw)
    stop = lambda clip, ch=-2, note=60: dict(clip=clip, midi_channel=ch, midi_note=note, stop=True)

    # G1: single mono 44.1 kHz loop, ratio 1, fractional-beat loop (sample-space wrap), default ADSR -- BASELINE config 1 in miniature
    L, _ = src(1500, stereo=False)
    run_scene("g1_mono_loop", B=1, VPB=2, fs=44100.0, mode=0, nframes=128, nblocks=16, sounds=[(L, None, 44100.0)],
              clips=[dict(set_length=(0.0625, 120), volume_abs=1.0, pan=0.0)], events={0: [play(0)]})

    # G2: stereo clips, pitched and resampled (44.1k -> 48k), pan / volume, wraps inside blocks, two buses, slice playback
    s0, s1, s2 = src(2600), src(1900), src(2200, stereo=False)
    clips = [dict(set_length=(0.09, 120), volume_abs=0.8, pan=-0.35), dict(set_length=(0.07, 120), volume_abs=0.55, pan=0.6, root_note=62),
             dict(set_length=(0.083, 120), volume_abs=0.9, pan=0.15)]
    ev = {0: [play(0, ch=-2, note=63, vol=0.7), play(1, ch=-1, note=57, vol=0.9), play(2, ch=-2, note=60, vol=0.5, change_slice=True, slice=3)]}
    for mode, nm in ((0, "g2_stereo_pitched"), (3, "g2_stereo_pitched_fixed"), (4, "g2_stereo_pitched_hermite")):
        run_scene(nm, B=2, VPB=4, fs=48000.0, mode=mode, nframes=128, nblocks=12,
                  sounds=[(s0[0], s0[1], 44100.0), (s1[0], s1[1], 48000.0), (s2[0], None, 44100.0)], clips=clips, events=ev)

    # G3: one-shot (non-looping) voices: release tail (quirk Q7) and hard stop at the slice end; voice slots free up
    a, b = src(1400), src(1100, stereo=False)
    run_scene("g3_oneshot_tail", B=1, VPB=4, fs=48000.0, mode=0, nframes=128, nblocks=14,
              sounds=[(a[0], a[1], 48000.0), (b[0], None, 48000.0)],
              clips=[dict(set_length=(0.05, 120), adsr=(0.0, 0.1, 1.0, 0.004)), dict(set_length=(0.04, 120), adsr=(0.0, 0.1, 1.0, 0.0))],
              events={0: [play(0, loop=False, note=60, vol=0.8), play(1, loop=False, note=67, vol=0.6)], 9: [play(1, loop=False, note=55, vol=1.0)]})

    # G4: beat-locked loop (integer beats): restart driven by the JACK clock against nextLoopUsecs (Q9a)
    c0 = src(3000)
    run_scene("g4_beat_locked", B=1, VPB=2, fs=48000.0, mode=0, nframes=128, nblocks=20, bpm=200,
              sounds=[(c0[0], c0[1], 48000.0)], clips=[dict(length_beats=1.0, length_sec=0.05, volume_abs=0.7, pan=0.2)],
              events={0: [dict(play(0, note=60, vol=0.9), tick=0)]},
              playhead=lambda k, period, sub: (0, 0))

    # G5: attack / decay envelope, note-off by stop command (linear release), volume change and restart merge
    d0 = src(2400)
    run_scene("g5_adsr_commands", B=1, VPB=3, fs=48000.0, mode=0, nframes=128, nblocks=18,
              sounds=[(d0[0], d0[1], 48000.0)], clips=[dict(set_length=(0.1, 120), adsr=(0.006, 0.004, 0.6, 0.008), pan=-0.2)],
              events={0: [play(0, note=60, vol=0.9)],
                      5: [dict(clip=0, midi_channel=-2, midi_note=60, change_volume=True, volume=0.4)],
                      8: [stop(0)],
                      12: [play(0, note=65, vol=0.7)]})

    # G6: pitch ratios of 1/16 and 16 (binade crossings every frame / every few blocks) and loops of a few frames
    # (several restarts per block), linear and Hermite
    e0, e1, e2, e3 = src(4000), src(900), src(700, stereo=False), src(5000)
    clips6 = [dict(set_length=(0.11, 120), volume_abs=0.9, pan=0.1), dict(length_beats=0.013, length_sec=7.0 / 48000.0, volume_abs=0.7, pan=-0.5),
              dict(length_beats=0.011, length_sec=3.0 / 48000.0, volume_abs=0.8, pan=0.4), dict(set_length=(0.17, 120), volume_abs=0.6, pan=-0.2)]
    ev6 = {0: [play(0, note=12, vol=0.8), play(1, note=64, vol=0.6), play(2, note=55, vol=0.9), play(3, note=108, vol=0.5)]}
    for mode, nm in ((0, "g6_extreme_ratios_tiny_loops"), (4, "g6_extreme_ratios_tiny_loops_hermite")):
        run_scene(nm, B=1, VPB=4, fs=48000.0, mode=mode, nframes=128, nblocks=10,
                  sounds=[(e0[0], e0[1], 48000.0), (e1[0], e1[1], 48000.0), (e2[0], None, 48000.0), (e3[0], e3[1], 48000.0)],
                  clips=clips6, events=ev6)

    # G7: a loop longer than its file (quirk Q5: silence past the last frame), lengthInBeats = -1 (quirk Q10: the
    # float -> quint64 tick count saturates), mono source through Hermite, one-shot running off the end
    f0, f1, f2 = src(1500), src(1300, stereo=False), src(1000)
    clips7 = [dict(length_beats=0.3, length_sec=0.05, volume_abs=0.8, pan=0.25), dict(length_sec=0.02, volume_abs=0.9, pan=-0.3),
              dict(length_beats=0.25, length_sec=0.03, adsr=(0.0, 0.1, 1.0, 0.0))]
    ev7 = {0: [play(0, note=60, vol=0.7), play(1, note=62, vol=0.8), play(2, loop=False, note=57, vol=0.9)]}
    for mode, nm in ((0, "g7_past_the_end_q10"), (4, "g7_past_the_end_q10_hermite")):
        run_scene(nm, B=1, VPB=4, fs=48000.0, mode=mode, nframes=128, nblocks=12,
                  sounds=[(f0[0], f0[1], 48000.0), (f1[0], None, 48000.0), (f2[0], f2[1], 48000.0)], clips=clips7, events=ev7)

    # G8: 256-frame blocks, sources at 22.05 / 96 / 44.1 kHz played at 48 kHz, start offset + slice playback
    g0, g1, g2 = src(3000), src(9000), src(2500, stereo=False)
    clips8 = [dict(set_length=(0.12, 120), volume_abs=0.85, pan=0.3), dict(start_sec=0.01, set_length=(0.06, 120), volume_abs=0.5, pan=-0.6),
              dict(set_length=(0.09, 120), volume_abs=0.75, pan=0.0)]
    ev8 = {0: [play(0, note=60, vol=0.8), play(1, note=65, vol=0.6, change_slice=True, slice=5), play(2, note=53, vol=0.9)],
           4: [dict(clip=1, midi_channel=-2, midi_note=65, change_slice=True, slice=5, change_volume=True, volume=0.3)]}
    run_scene("g8_resampled_256", B=1, VPB=4, fs=48000.0, mode=0, nframes=256, nblocks=8,
              sounds=[(g0[0], g0[1], 22050.0), (g1[0], g1[1], 96000.0), (g2[0], None, 44100.0)], clips=clips8, events=ev8)

    # G9: envelopes that span many blocks: slow attack into a slow decay, sustain levels 0.35 / 0 / 1, note-off with a long
    # linear release that runs the voice out, a second note-off inside the attack
    h0, h1, h2 = src(6000), src(5000, stereo=False), src(7000)
    clips9 = [dict(set_length=(0.2, 120), adsr=(0.03, 0.05, 0.35, 0.04), pan=0.2), dict(set_length=(0.15, 120), adsr=(0.012, 0.02, 0.0, 0.02), pan=-0.4),
              dict(set_length=(0.25, 120), adsr=(0.08, 0.0, 1.0, 0.06), volume_abs=0.7)]
    ev9 = {0: [play(0, note=60, vol=0.9), play(1, note=64, vol=0.7), play(2, note=55, vol=0.8)],
           9: [stop(2, note=55)], 16: [stop(0, note=60)], 22: [play(1, note=67, vol=0.6)]}
    run_scene("g9_long_envelopes", B=1, VPB=4, fs=48000.0, mode=0, nframes=128, nblocks=40,
              sounds=[(h0[0], h0[1], 48000.0), (h1[0], None, 48000.0), (h2[0], h2[1], 44100.0)], clips=clips9, events=ev9)


if __name__ == "__main__":
    main()
    if "--config1" not in sys.argv and "--all" in sys.argv:
        config1_full_shape()
