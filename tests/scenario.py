"""Shared scene description for parity tests: the same seeded inputs are played through the CPU
oracle, the CPU harness (host build of the kernels' code) and the HIP engine (C-ABI)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from libzl_amd import _abi
from libzl_amd.engine import synthetic_clocks
from oracle import zl_oracle as zo

# reference field name (oracle, camelCase) -> engine ABI field name
_CMD_MAP = {
    "clip": "clip", "midiNote": "midi_note", "midiChannel": "midi_channel", "startPlayback": "start_playback",
    "stopPlayback": "stop_playback", "changeSlice": "change_slice", "slice": "slice", "changeLooping": "change_looping",
    "looping": "looping", "changePitch": "change_pitch", "pitchChange": "pitch_change", "changeSpeed": "change_speed",
    "speedRatio": "speed_ratio", "changeGainDb": "change_gain_db", "gainDb": "gain_db", "changeVolume": "change_volume",
    "volume": "volume",
}


def oracle_cmd(**f) -> zo.ClipCommand:
    return zo.clip_command(**f)


def engine_cmd(**f) -> _abi.ClipCommand:
    c = _abi.ClipCommand()
    c.clip = -1; c.midi_note = -1; c.midi_channel = -1; c.slice = -1
    for k, v in f.items():
        setattr(c, _CMD_MAP[k], v)
    return c


def play_cmd(clip, midi_channel=-2, loop=True, note=60, volume=1.0, slice=-1, **extra):
    """The ClipCommand ClipAudioSource::play builds (ClipAudioSource.cpp:415-429)."""
    f = dict(clip=clip, midiChannel=midi_channel, midiNote=note, changeVolume=1, volume=volume, looping=1 if loop else 0,
             startPlayback=1, slice=slice)
    if loop:
        f["stopPlayback"] = 1
    f.update(extra)
    return f


def stop_cmd(clip, midi_channel=-2, note=60, **extra):
    """ClipAudioSource::stop on one channel (ClipAudioSource.cpp:431-437)."""
    f = dict(clip=clip, midiChannel=midi_channel, midiNote=note, stopPlayback=1)
    f.update(extra)
    return f


def snapshot_clip(clip: zo.Clip) -> _abi.ClipParams:
    p = _abi.ClipParams()
    p.start_position_seconds = clip.startPositionInSeconds
    p.length_seconds = clip.lengthInSeconds
    p.length_in_beats = clip.lengthInBeats
    p.volume_absolute = clip.volumeAbsolute
    p.pan = clip.pan
    p.duration_seconds = clip.duration
    p.adsr_attack = clip.adsr.p.attack
    p.adsr_decay = clip.adsr.p.decay
    p.adsr_sustain = clip.adsr.p.sustain
    p.adsr_release = clip.adsr.p.release
    p.root_note = clip.rootNote
    p.num_slice_positions = clip.nSlicePositions
    for i in range(clip.nSlicePositions):
        p.slice_positions[i] = clip.slicePositions[i]
    return p


@dataclass
class Scene:
    num_buses: int = 12
    voices_per_bus: int = 8
    fs: float = 48000.0
    mode: int = 0
    mix_group: int = 0                      # voices per task; 0 = whole bus sequential (reference order)
    nframes: int = 256
    nblocks: int = 8
    sounds: List[Tuple[np.ndarray, Optional[np.ndarray], float]] = field(default_factory=list)
    # clip_setup[i](oracle_lib, clip_struct): configure clip i through the oracle's restated setters
    clip_setup: Dict[int, Callable] = field(default_factory=dict)
    # events[k] = list of actions applied before block k is rendered:
    #   ("cmd", fields, tick) | ("start", bus, slot, fields, tick) | ("clip", clip_id, fn)
    #   | ("update", bus, slot, fields) | ("stopv", bus, slot, allow_tail_off): the voice-level calls (zlhip_update_voice / _stop_voice)
    #   | ("enable", bus, flag): SamplerSynth::setChannelEnabled
    events: Dict[int, list] = field(default_factory=dict)
    clocks: Optional[Callable[[int, int], "C.Array"]] = None     # (start_block, n) -> Clock array
    bpm: int = 120
    moving_playhead: bool = False           # SyncTimer's playhead advances as a running timer's does (else frozen at tick 0 / usec 0)
    block0: int = 0                         # JACK cycle number of the scene's first block (a timer that has been running for a while)

    def make_clocks(self, start, n):
        if self.clocks is not None:
            return self.clocks(start, n)
        return synthetic_clocks(n, self.nframes, self.fs, start_block=self.block0 + start, bpm=self.bpm, moving_playhead=self.moving_playhead)

    def tick_at(self, block):
        """the playhead SyncTimer dispatches the first step of cycle `block` with (the playhead after the previous cycle)"""
        if not self.moving_playhead:
            return 0
        from libzl_amd.engine import running_playhead
        k = self.block0 + block
        return running_playhead(k - 1, int(round(1e6 * self.nframes / self.fs)), self.bpm)[0] if k > 0 else 0


def _segments(scene: Scene, batch: int):
    """Split [0, nblocks) at event blocks and into chunks of at most `batch` blocks."""
    cuts = sorted(set([0, scene.nblocks] + [k for k in scene.events if 0 <= k < scene.nblocks]))
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        k = a
        while k < b:
            n = min(batch, b - k)
            out.append((k, n))
            k += n
    return out


def run_oracle(scene: Scene, batch: int = 1 << 30, threads: int = 1, fast: bool = False):
    osyn = zo.OracleSynth(scene.num_buses, scene.voices_per_bus, scene.fs, scene.mode, mix_group=scene.mix_group,
                          max_sounds=max(8, len(scene.sounds)), fast=fast)
    for i, (L, R, sr) in enumerate(scene.sounds):
        cid = osyn.register_clip(L, R, sr)
        assert cid == i
        if i in scene.clip_setup:
            scene.clip_setup[i](osyn.lib, osyn.clips[i])
    buses = []
    reports = None
    for (k0, n) in _segments(scene, batch):
        for ev in scene.events.get(k0, []):
            if ev[0] == "cmd":
                osyn.handle_clip_command(oracle_cmd(**ev[1]), ev[2])
            elif ev[0] == "start":
                osyn.start_voice(ev[1], ev[2], oracle_cmd(**ev[3]), ev[4])
            elif ev[0] == "update":
                osyn.update_voice(ev[1], ev[2], oracle_cmd(**ev[3]))
            elif ev[0] == "stopv":
                osyn.stop_voice(ev[1], ev[2], ev[3])
            elif ev[0] == "enable":
                osyn.set_bus_enabled(ev[1], ev[2])
            elif ev[0] == "clip":
                ev[2](osyn.lib, osyn.clips[ev[1]])
        bus, reports = osyn.render_batch(n, scene.nframes, scene.make_clocks(k0, n), threads=threads)
        buses.append(bus)
    return np.concatenate(buses, axis=2), reports, osyn


def run_backend(scene: Scene, factory: Callable, batch: int = 1 << 30, trace: bool = False, force_slow: bool = False,
                pipelined: bool = False, no_periodic: bool = False, fanout=None, bounce=None, **factory_kw):
    """factory(**kwargs) -> object with the libzl_amd.SamplerSynth surface (engine or CPU harness).
    pipelined (GPU engine only): every call renders into its own device buffer on one HIP stream and nothing is read
    back or synchronised until the end, so consecutive zlhip_render_batch calls overlap.
    fanout (GPU engine only): a PassthroughParams per bus; every call also writes the fused JackPassthrough fan-out, which
    is left in syn.fan_result as [num_buses][6][frames].
    bounce (GPU engine only): (format, sub_blocks) -- every segment between events goes through zlhip_bounce in sub-batches of
    sub_blocks blocks (the engine is created with max_batch_blocks = sub_blocks); "pcm16" results are [num_buses][frames][2]."""
    # the oracle's setters are the single source of clip parameters for both sides
    ref = zo.OracleSynth(1, 1, scene.fs, scene.mode, max_sounds=max(8, len(scene.sounds)))
    syn = factory(num_buses=scene.num_buses, voices_per_bus=scene.voices_per_bus, mode=scene.mode,
                  playback_sample_rate=scene.fs, voices_per_task=scene.mix_group, max_frames=max(64, scene.nframes),
                  max_batch_blocks=max(1, min(bounce[1] if bounce else batch, scene.nblocks)), max_sounds=max(8, len(scene.sounds)),
                  sound_arena_bytes=max(1 << 20, sum((s[0].shape[0] + 16) * 8 for s in scene.sounds) + (1 << 16)), **factory_kw)
    for i, (L, R, sr) in enumerate(scene.sounds):
        assert ref.register_clip(L, R, sr) == i
        cid = syn.register_clip(L, R, sr)
        assert cid == i
        if i in scene.clip_setup:
            scene.clip_setup[i](ref.lib, ref.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    if trace or force_slow or no_periodic:
        syn.enable_trace(trace or force_slow, force_slow=force_slow, no_periodic=no_periodic)
    buses, traces, fans = [], [], []
    if pipelined or fanout is not None:
        import torch
    if pipelined:
        stream = torch.cuda.Stream()
    for (k0, n) in _segments(scene, batch):
        for ev in scene.events.get(k0, []):
            if ev[0] == "cmd":
                syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
            elif ev[0] == "start":
                syn.start_voice(ev[1], ev[2], engine_cmd(**ev[3]), ev[4])
            elif ev[0] == "update":
                syn.update_voice(ev[1], ev[2], engine_cmd(**ev[3]))
            elif ev[0] == "stopv":
                syn.stop_voice(ev[1], ev[2], ev[3])
            elif ev[0] == "enable":
                syn.set_bus_enabled(ev[1], ev[2])
            elif ev[0] == "clip":
                ev[2](ref.lib, ref.clips[ev[1]])
                syn.set_clip_params(ev[1], snapshot_clip(ref.clips[ev[1]]))
        fan_kw = {}
        if fanout is not None:
            fans.append(torch.full((scene.num_buses, 6, n * scene.nframes), 7.0, device="cuda", dtype=torch.float32))
            fan_kw = dict(fan_params=fanout, fan_out_dev=fans[-1].data_ptr())
        if bounce is not None:
            pageable = None
            if len(bounce) > 2 and bounce[2] == "pageable":        # the caller's own (pageable) memory: the copy-engine path
                pageable = np.full((scene.num_buses, n * scene.nframes, 2), 0x5a5a, dtype=np.int16) if bounce[0] == "pcm16" else \
                    np.full((scene.num_buses, 2, n * scene.nframes), 9.0, dtype=np.float32)
            buses.append(np.array(syn.bounce(n, scene.nframes, scene.make_clocks(k0, n), fmt=bounce[0], sub_blocks=bounce[1], out=pageable), copy=True))
            continue
        if pipelined:
            out = torch.zeros((scene.num_buses, 2, n * scene.nframes), device="cuda", dtype=torch.float32)
            syn.render_batch(n, scene.nframes, scene.make_clocks(k0, n), bus_out_dev=out.data_ptr(), stream=stream.cuda_stream, **fan_kw)
            buses.append(out)
            continue
        syn.render_batch(n, scene.nframes, scene.make_clocks(k0, n), **fan_kw)
        buses.append(np.array(syn.read_bus(), copy=True))
        if trace:
            traces.append(syn.read_trace())
    if pipelined:
        syn.synchronize()
        torch.cuda.synchronize()
        buses = [b.cpu().numpy() for b in buses]
    if fanout is not None:
        syn.synchronize()
        torch.cuda.synchronize()
        syn.fan_result = np.concatenate([f.cpu().numpy() for f in fans], axis=2)
    reports = syn.voice_reports()
    return np.concatenate(buses, axis=1 if (bounce and bounce[0] == "pcm16") else 2), reports, syn, (np.concatenate(traces, axis=0) if traces else None)


def oracle_trace(scene: Scene):
    """Per-voice per-frame (int)sourceSamplePosition from the oracle, rendering voices one by one
    (valid for scenes whose clips are not shared between voices): [nblocks][V][N]."""
    osyn = zo.OracleSynth(scene.num_buses, scene.voices_per_bus, scene.fs, scene.mode, max_sounds=max(8, len(scene.sounds)))
    for i, (L, R, sr) in enumerate(scene.sounds):
        osyn.register_clip(L, R, sr)
        if i in scene.clip_setup:
            scene.clip_setup[i](osyn.lib, osyn.clips[i])
    V = scene.num_buses * scene.voices_per_bus
    out = np.full((scene.nblocks, V, scene.nframes), -1, dtype=np.int32)
    for k in range(scene.nblocks):
        for ev in scene.events.get(k, []):
            if ev[0] == "cmd":
                osyn.handle_clip_command(oracle_cmd(**ev[1]), ev[2])
            elif ev[0] == "start":
                osyn.start_voice(ev[1], ev[2], oracle_cmd(**ev[3]), ev[4])
            elif ev[0] == "clip":
                ev[2](osyn.lib, osyn.clips[ev[1]])
        clk = scene.make_clocks(k, 1)[0]
        for v in range(V):
            if osyn.voices[v].isPlaying:
                _, _, tr, _ = osyn.voice_trace(v, scene.nframes, clk)
                out[k, v] = tr
    return out, osyn


def rand_source(rng, length, stereo=True):
    L = rng.uniform(-1, 1, length).astype(np.float32)
    R = rng.uniform(-1, 1, length).astype(np.float32) if stereo else None
    return L, R


def random_scene(seed, *, num_buses=3, voices_per_bus=8, nframes=128, nblocks=24, nclips=10, mode=0, mix_group=0,
                 fs=48000.0, min_len=1500, max_len=6000, events=True):
    """Seeded mixed scene: looping (sample-space and beat-locked) and one-shot clips, mono and stereo, pitched and
    resampled, envelopes with attack / decay / release, commands and clip-parameter edits between blocks."""
    rng = np.random.default_rng(seed)
    sc = Scene(num_buses=num_buses, voices_per_bus=voices_per_bus, fs=fs, mode=mode, mix_group=mix_group,
               nframes=nframes, nblocks=nblocks, bpm=int(rng.choice([90, 120, 174])))
    kinds = []
    for i in range(nclips):
        sr = float(rng.choice([44100.0, 48000.0, 22050.0]))
        n = int(rng.integers(min_len, max_len))
        L, R = rand_source(rng, n, stereo=bool(rng.random() < 0.7))
        sc.sounds.append((L, R, sr))
        kind = ["loop", "loop", "beat", "oneshot"][int(rng.integers(0, 4))]
        kinds.append(kind)
        beats = float(rng.uniform(0.03, 0.2)) if kind != "beat" else float(rng.integers(1, 3))
        vol, pan = float(rng.uniform(0.2, 1.0)), float(rng.uniform(-1, 1))
        adsr_kind = int(rng.integers(0, 4))
        start = float(rng.uniform(0, 0.01)) if rng.random() < 0.4 else 0.0
        dur = n / sr

        def setup(lib, clip, kind=kind, beats=beats, vol=vol, pan=pan, adsr_kind=adsr_kind, start=start, dur=dur):
            if kind == "beat":
                clip.lengthInBeats = beats                     # integer beats -> clock-driven restart (Q9a)
                clip.lengthInSeconds = float(np.float32(dur * 0.6))
            else:
                lib.zlo_clip_set_length(clip, C.c_float(beats), 120)
                if clip.lengthInSeconds > dur * 0.9:
                    clip.lengthInSeconds = float(np.float32(dur * 0.5))
            lib.zlo_clip_set_start_position(clip, C.c_float(start))
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(vol))
            lib.zlo_clip_set_pan(clip, C.c_float(pan))
            if adsr_kind == 1:
                clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = (0.004, 0.003, 0.7, 0.006)
            elif adsr_kind == 2:
                clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = (0.0, 0.1, 1.0, 0.0)
            elif adsr_kind == 3:
                clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = (0.0, 0.002, 0.5, 0.01)
        sc.clip_setup[i] = setup
    ev0 = []
    for i in range(nclips):
        ch = int(rng.integers(0, num_buses)) - 2
        ev0.append(("cmd", play_cmd(i, midi_channel=ch, loop=(kinds[i] != "oneshot"), note=int(rng.integers(52, 70)),
                                    volume=float(np.float32(rng.uniform(0.2, 1.0))),
                                    **({"changeSlice": 1, "slice": int(rng.integers(0, 12))} if rng.random() < 0.2 else {})), int(rng.integers(0, 50))))
    sc.events[0] = ev0
    if events:
        for k in sorted(set(int(x) for x in rng.integers(1, nblocks, size=6))):
            i = int(rng.integers(0, nclips))
            ch_all = [c for c in range(-2, num_buses - 2)]
            what = int(rng.integers(0, 7))
            lst = sc.events.setdefault(k, [])
            flip = bool(rng.integers(0, 2))
            if what == 6:            # SamplerSynth::setChannelEnabled: a channel stands still for a few blocks (or to the end)
                bus = int(rng.integers(0, num_buses)); back = k + int(rng.integers(1, 6))
                lst.append(("enable", bus, False))
                if back < nblocks:
                    sc.events.setdefault(back, []).append(("enable", bus, True))
                continue
            for ch in ch_all:
                if what == 0:
                    lst.append(("cmd", stop_cmd(i, midi_channel=ch, note=ev0[i][1]["midiNote"]), 0))
                elif what == 1:
                    lst.append(("cmd", dict(clip=i, midiChannel=ch, midiNote=ev0[i][1]["midiNote"], changeVolume=1,
                                            volume=float(np.float32(rng.uniform(0.1, 1.0)))), 0))
                elif what == 2:
                    lst.append(("cmd", play_cmd(i, midi_channel=ch, loop=(kinds[i] != "oneshot"), note=int(rng.integers(55, 66)), volume=0.6), k * 7))
                elif what == 4:      # a patch of the playing voice: loop <-> one-shot (+ the stored-only fields), SamplerSynthVoice.cpp:58-100
                    lst.append(("cmd", dict(clip=i, midiChannel=ch, midiNote=ev0[i][1]["midiNote"], changeLooping=1, looping=1 if flip else 0,
                                            changePitch=1, pitchChange=0.5, changeGainDb=1, gainDb=-2.0), 0))
                elif what == 5:      # a patch addressed by slice: reaches only voices started with that slice
                    lst.append(("cmd", dict(clip=i, midiChannel=ch, midiNote=60, changeSlice=1, slice=ev0[i][1].get("slice", 3),
                                            changeVolume=1, volume=float(np.float32(0.3 + 0.4 * flip))), 0))
            if what == 3:
                newpan, newvol = float(rng.uniform(-1, 1)), float(rng.uniform(0.1, 1))

                def edit(lib, clip, newpan=newpan, newvol=newvol):
                    lib.zlo_clip_set_pan(clip, C.c_float(newpan))
                    lib.zlo_clip_set_volume_absolute(clip, C.c_float(newvol))
                lst.append(("clip", i, edit))
    return sc


def compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, V, *, exact=True, tol=1e-6):
    """Asserts audio, voice state and reports of a backend run equal the oracle's."""
    if exact:
        assert np.array_equal(ref_bus.view(np.int32), bus.view(np.int32)), \
            f"audio differs: max abs diff {np.abs(ref_bus - bus).max()} at {np.argwhere(ref_bus != bus)[:3].tolist()}"
    else:
        scale = max(1.0, float(np.abs(ref_bus).max()))
        assert np.abs(ref_bus - bus).max() <= tol * scale
    for v in range(V):
        ov = ref_syn.voices[v]
        assert bool(ov.isPlaying) == bool(rep[v].playing), f"voice {v}: isPlaying {ov.isPlaying} vs {rep[v].playing}"
        if ov.isPlaying:
            assert ov.sourceSamplePosition == rep[v].source_sample_position, f"voice {v}: position"
        assert ref_rep[v].valid == rep[v].valid, f"voice {v}: report validity"
        if ref_rep[v].valid:
            assert ref_rep[v].gain == rep[v].gain and ref_rep[v].progress == rep[v].progress, \
                f"voice {v}: report ({ref_rep[v].gain}, {ref_rep[v].progress}) vs ({rep[v].gain}, {rep[v].progress})"
