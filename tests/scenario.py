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
    events: Dict[int, list] = field(default_factory=dict)
    clocks: Optional[Callable[[int, int], "C.Array"]] = None     # (start_block, n) -> Clock array
    bpm: int = 120

    def make_clocks(self, start, n):
        if self.clocks is not None:
            return self.clocks(start, n)
        return synthetic_clocks(n, self.nframes, self.fs, start_block=start, bpm=self.bpm)


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
            elif ev[0] == "clip":
                ev[2](osyn.lib, osyn.clips[ev[1]])
        bus, reports = osyn.render_batch(n, scene.nframes, scene.make_clocks(k0, n), threads=threads)
        buses.append(bus)
    return np.concatenate(buses, axis=2), reports, osyn


def run_backend(scene: Scene, factory: Callable, batch: int = 1 << 30, trace: bool = False, force_slow: bool = False):
    """factory(**kwargs) -> object with the libzl_amd.SamplerSynth surface (engine or CPU harness)."""
    # the oracle's setters are the single source of clip parameters for both sides
    ref = zo.OracleSynth(1, 1, scene.fs, scene.mode, max_sounds=max(8, len(scene.sounds)))
    syn = factory(num_buses=scene.num_buses, voices_per_bus=scene.voices_per_bus, mode=scene.mode,
                  playback_sample_rate=scene.fs, voices_per_task=scene.mix_group, max_frames=max(64, scene.nframes),
                  max_batch_blocks=max(1, min(batch, scene.nblocks)), max_sounds=max(8, len(scene.sounds)),
                  sound_arena_bytes=max(1 << 20, sum((s[0].shape[0] + 16) * 8 for s in scene.sounds) + (1 << 16)))
    for i, (L, R, sr) in enumerate(scene.sounds):
        assert ref.register_clip(L, R, sr) == i
        cid = syn.register_clip(L, R, sr)
        assert cid == i
        if i in scene.clip_setup:
            scene.clip_setup[i](ref.lib, ref.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    if trace or force_slow:
        syn.enable_trace(True, force_slow=force_slow)
    buses, traces = [], []
    for (k0, n) in _segments(scene, batch):
        for ev in scene.events.get(k0, []):
            if ev[0] == "cmd":
                syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
            elif ev[0] == "start":
                syn.start_voice(ev[1], ev[2], engine_cmd(**ev[3]), ev[4])
            elif ev[0] == "clip":
                ev[2](ref.lib, ref.clips[ev[1]])
                syn.set_clip_params(ev[1], snapshot_clip(ref.clips[ev[1]]))
        syn.render_batch(n, scene.nframes, scene.make_clocks(k0, n))
        buses.append(np.array(syn.read_bus(), copy=True))
        if trace:
            traces.append(syn.read_trace())
    reports = syn.voice_reports()
    return np.concatenate(buses, axis=2), reports, syn, (np.concatenate(traces, axis=0) if traces else None)


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
