"""Python binding of the C-ABI in include/zlhip.h (ctypes; no torch types cross the boundary).

Names follow the reference's domain: a SamplerSynth owns `num_buses` SamplerChannels of
`voices_per_bus` voices (SamplerSynth.cpp:254-278); clips are registered once
(SamplerSynth::registerClip, :285-295) and played through ClipCommands (ClipCommand.h:11-32).
Every sample is produced by the HIP kernels behind libzlhip.so; this module only marshals.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._abi import (Clock, ClipCommand, ClipParams, Config, Levels, PassthroughParams, Timings, VoiceReport,
                   MODE_FAITHFUL, MODE_FIX_DELAY, MODE_FIX_GAIN, MODE_HERMITE, ZlHipError)

__all__ = ["SamplerSynth", "Clock", "ClipCommand", "ClipParams", "Levels", "PassthroughParams", "VoiceReport",
           "MODE_FAITHFUL", "MODE_FIX_GAIN", "MODE_FIX_DELAY", "MODE_HERMITE", "ZlHipError", "clip_command", "synthetic_clocks", "running_playhead"]


def clip_command(lib=None, **fields) -> ClipCommand:
    """ClipCommand with the reference's defaults (ClipCommand.h:13-32) and the given fields set."""
    c = ClipCommand()
    c.clip = -1
    c.midi_note = -1
    c.midi_channel = -1
    c.slice = -1
    for k, v in fields.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


_running_playheads: dict = {}


def running_playhead(block: int, period: int, bpm: int):
    """(jackPlayhead, jackPlayheadUsecs) as a SyncTimer that was started at time 0 leaves them after its process call for the
    JACK cycle [block * period, (block + 1) * period): one step per subbeat while its time lies before the cycle's end, the step
    time accumulated as `quint64 += double` (SyncTimer.cpp:484,512,660-667,990-1004)."""
    key = (period, bpm)
    rows = _running_playheads.setdefault(key, [])
    sub = float((60000000000 // (bpm * 96))) / 1000.0
    # jackNextPlaybackPosition = current_usecs of the first cycle = 0; the step clock runs in parallel from 0 as well
    n, pos = rows[-1] if rows else (0, 0)
    while len(rows) <= block:
        nxt = (len(rows) + 1) * period
        while pos < nxt:
            n += 1
            pos = int(float(pos) + sub)
        rows.append((n, pos))
    return rows[block]


def synthetic_clocks(nblocks: int, nframes: int, sample_rate: float, start_block: int = 0, bpm: int = 120,
                     moving_playhead: bool = False) -> "C.Array[Clock]":
    """Monotone JACK-like cycle times: current_usecs = k * round(1e6 * nframes / fs).
    The SyncTimer playhead is held at tick 0 / usec 0 with the subbeat length of `bpm` (SyncTimer.cpp:180-183,959) -- a
    stopped timer before its first cycle -- or, with moving_playhead, advances as a running timer's does (running_playhead)."""
    period = int(round(1e6 * nframes / sample_rate))
    subbeat = ((1 * 60000000000) // (bpm * 96)) // 1000
    arr = (Clock * nblocks)()
    for k in range(nblocks):
        kk = start_block + k
        arr[k].current_usecs = kk * period
        arr[k].next_usecs = (kk + 1) * period
        if moving_playhead:
            arr[k].jack_playhead, arr[k].jack_playhead_usecs = running_playhead(kk, period, bpm)
        else:
            arr[k].jack_playhead = 0
            arr[k].jack_playhead_usecs = 0
        arr[k].jack_subbeat_length_usecs = subbeat
    return arr


def pinned_array(lib, shape, dtype) -> np.ndarray:
    """A numpy array in page-locked host memory (zlhip_host_alloc); freed when the array and its views are gone."""
    import weakref
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    if lib.zlhip_host_alloc(max(nbytes, 1), C.byref(p)) != 0 or not p.value:
        raise MemoryError(f"zlhip_host_alloc({nbytes}) failed")
    buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
    weakref.finalize(buf, lib.zlhip_host_free, p.value)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


@dataclass
class BatchResult:
    bus: np.ndarray            # [num_buses, 2, nblocks*nframes] float32
    reports: np.ndarray        # structured array of VoiceReport


class SamplerSynth:
    def __init__(self, num_buses: int = 12, voices_per_bus: int = 8, *, max_frames: int = 1024,
                 max_batch_blocks: int = 64, max_sounds: int = 1024, mode: int = MODE_FAITHFUL,
                 playback_sample_rate: float = 48000.0, sound_arena_bytes: int = 256 << 20,
                 voices_per_task: int = 0, plan_window_blocks: int = 0, device: int = 0, rt_idle_timeout_us: int = 0,
                 sound_arena_max_bytes: int = 0):
        self._lib = _abi.load()
        cfg = Config()
        self._lib.zlhip_config_default(C.byref(cfg))
        cfg.device = device
        cfg.num_buses = num_buses
        cfg.voices_per_bus = voices_per_bus
        cfg.max_frames = max_frames
        cfg.max_batch_blocks = max_batch_blocks
        cfg.max_sounds = max_sounds
        cfg.mode = mode
        cfg.playback_sample_rate = playback_sample_rate
        cfg.sound_arena_bytes = sound_arena_bytes
        cfg.voices_per_task = voices_per_task
        cfg.plan_window_blocks = plan_window_blocks
        cfg.rt_idle_timeout_us = rt_idle_timeout_us
        cfg.sound_arena_max_bytes = sound_arena_max_bytes
        self.cfg = cfg
        self._e = C.c_void_p()
        rc = self._lib.zlhip_engine_create(C.byref(cfg), C.byref(self._e))
        if rc != 0:
            raise ZlHipError(f"zlhip_engine_create: {self._lib.zlhip_strerror(rc).decode()} ({rc})")
        self.num_buses = num_buses
        self.voices_per_bus = voices_per_bus
        self.num_voices = num_buses * voices_per_bus
        self._last = (0, 0)

    # -- lifecycle --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_e", None) and self._e.value:
            self._lib.zlhip_engine_destroy(self._e)
            self._e = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc, what):
        return _abi.check(self._lib, self._e, rc, what)

    @property
    def handle(self):
        return self._e

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._ck(self._lib.zlhip_device_name(self._e, buf, 256), "device_name")
        return buf.value.decode()

    # -- clips ------------------------------------------------------------------------------
    def register_clip(self, left: np.ndarray, right: Optional[np.ndarray], sample_rate: float) -> int:
        """SamplerSynth::registerClip + SamplerSynthSound::loadSoundData: upload planar fp32."""
        left = np.ascontiguousarray(left, dtype=np.float32)
        rp = None
        if right is not None:
            right = np.ascontiguousarray(right, dtype=np.float32)
            assert right.shape == left.shape
            rp = right.ctypes.data
        out = C.c_int32(-1)
        self._ck(self._lib.zlhip_sound_upload(self._e, left.ctypes.data, rp, left.shape[0], float(sample_rate), C.byref(out)), "sound_upload")
        return out.value

    def register_clip_device(self, left_ptr: int, right_ptr: Optional[int], length: int, sample_rate: float) -> int:
        out = C.c_int32(-1)
        self._ck(self._lib.zlhip_sound_upload_device(self._e, left_ptr, right_ptr, length, float(sample_rate), C.byref(out)), "sound_upload_device")
        return out.value

    def register_clip_device_on(self, left_ptr: int, right_ptr: Optional[int], length: int, sample_rate: float, producer_stream: Optional[int]) -> int:
        """register_clip_device with the HIP stream the planes were produced on: the engine waits for that stream (an event), not for the device."""
        out = C.c_int32(-1)
        self._ck(self._lib.zlhip_sound_upload_device_on(self._e, left_ptr, right_ptr, length, float(sample_rate), producer_stream, C.byref(out)), "sound_upload_device_on")
        return out.value

    def unregister_clip(self, clip: int):
        self._ck(self._lib.zlhip_sound_release(self._e, clip), "sound_release")

    def default_clip_params(self, duration_seconds: float) -> ClipParams:
        p = ClipParams()
        self._lib.zlhip_clip_params_default(C.byref(p), float(duration_seconds))
        return p

    def set_clip_params(self, clip: int, params: ClipParams):
        self._ck(self._lib.zlhip_clip_set(self._e, clip, C.byref(params)), "clip_set")

    # -- commands ---------------------------------------------------------------------------
    def handle_clip_command(self, cmd: ClipCommand, current_tick: int = 0) -> int:
        return self._ck(self._lib.zlhip_handle_command(self._e, C.byref(cmd), current_tick), "handle_command")

    def handle_clip_commands(self, cmds: Sequence[ClipCommand], current_tick: int = 0, want_voices: bool = False):
        """A block's worth of commands in one call; returns the per-command results (1 taken / 0 dropped) -- with want_voices also
        the voice (bus * voices_per_bus + slot) each command started, -1 if none."""
        n = len(cmds)
        arr = (ClipCommand * n)(*cmds)
        taken = (C.c_int32 * n)()
        if want_voices:
            voices = (C.c_int32 * n)()
            self._ck(self._lib.zlhip_handle_commands_voices(self._e, arr, n, current_tick, taken, voices), "handle_commands")
            return list(taken), list(voices)
        self._ck(self._lib.zlhip_handle_commands(self._e, arr, n, current_tick, taken), "handle_commands")
        return list(taken)

    def set_bus_enabled(self, bus: int, enabled: bool) -> None:
        """SamplerSynth::setChannelEnabled for bus = channel + 2: a disabled bus takes commands but its voices stand still."""
        self._ck(self._lib.zlhip_bus_set_enabled(self._e, bus, 1 if enabled else 0), "bus_set_enabled")

    def start_voice(self, bus: int, slot: int, cmd: ClipCommand, current_tick: int = 0) -> int:
        return self._ck(self._lib.zlhip_start_voice(self._e, bus, slot, C.byref(cmd), current_tick), "start_voice")

    def stop_voice(self, bus: int, slot: int, allow_tail_off: bool = True) -> int:
        """SamplerSynthVoice::stopNote on one voice slot."""
        return self._ck(self._lib.zlhip_stop_voice(self._e, bus, slot, 1 if allow_tail_off else 0), "stop_voice")

    def update_voice(self, bus: int, slot: int, cmd: ClipCommand) -> int:
        """SamplerSynthVoice::setCurrentCommand on a playing voice."""
        return self._ck(self._lib.zlhip_update_voice(self._e, bus, slot, C.byref(cmd)), "update_voice")

    def voice_is_playing(self, bus: int, slot: int) -> bool:
        return self._ck(self._lib.zlhip_voice_is_playing(self._e, bus, slot), "voice_is_playing") == 1

    # -- render -----------------------------------------------------------------------------
    def process(self, nframes: int, clock: Clock):
        """One real-time cycle of every SamplerChannel; returns (left[B,N], right[B,N])."""
        L = np.empty((self.num_buses, nframes), dtype=np.float32)
        R = np.empty((self.num_buses, nframes), dtype=np.float32)
        self._ck(self._lib.zlhip_render(self._e, nframes, C.byref(clock), L.ctypes.data, R.ctypes.data), "render")
        self._last = (1, nframes)
        return L, R

    def process_into(self, nframes: int, clock: Clock, left: np.ndarray, right: np.ndarray, fan_params: Optional[Sequence[PassthroughParams]] = None,
                     fan: Optional[np.ndarray] = None):
        """One real-time cycle into the caller's arrays ([B, nframes] each, fan [B, 6, nframes]).  Page-locked arrays (pinned_array) are
        written by the kernels directly -- no host copy behind the cycle."""
        if fan is not None:
            arr = (PassthroughParams * self.num_buses)(*fan_params)
            self._ck(self._lib.zlhip_render_fanout(self._e, nframes, C.byref(clock), left.ctypes.data, right.ctypes.data, arr, fan.ctypes.data), "render_fanout")
        else:
            self._ck(self._lib.zlhip_render(self._e, nframes, C.byref(clock), left.ctypes.data, right.ctypes.data), "render")
        self._last = (1, nframes)

    def process_fanout(self, nframes: int, clock: Clock, fan_params: Sequence[PassthroughParams]):
        """One real-time cycle with the JackPassthrough client behind every bus (zlhip_render_fanout): returns
        (left[B,N], right[B,N], fan[B,6,N]) -- fan rows = dryL, dryR, fx1L, fx1R, fx2L, fx2R."""
        L = np.empty((self.num_buses, nframes), dtype=np.float32)
        R = np.empty((self.num_buses, nframes), dtype=np.float32)
        fan = np.empty((self.num_buses, 6, nframes), dtype=np.float32)
        arr = (PassthroughParams * self.num_buses)(*fan_params)
        self._ck(self._lib.zlhip_render_fanout(self._e, nframes, C.byref(clock), L.ctypes.data, R.ctypes.data, arr, fan.ctypes.data), "render_fanout")
        self._last = (1, nframes)
        return L, R, fan

    def render_batch(self, nblocks: int, nframes: int, clocks, bus_out_dev: Optional[int] = None, stream: Optional[int] = None,
                     fan_params: Optional[Sequence[PassthroughParams]] = None, fan_out_dev: Optional[int] = None):
        """fan_params + fan_out_dev: also write the JackPassthrough fan-out [num_buses][6][nblocks*nframes] (fused)."""
        if fan_out_dev is not None:
            arr = (PassthroughParams * self.num_buses)(*fan_params)
            self._ck(self._lib.zlhip_render_batch_fanout(self._e, nblocks, nframes, clocks, bus_out_dev, arr, fan_out_dev, stream), "render_batch_fanout")
        else:
            self._ck(self._lib.zlhip_render_batch(self._e, nblocks, nframes, clocks, bus_out_dev, stream), "render_batch")
        self._last = (nblocks, nframes)

    def synchronize(self):
        self._ck(self._lib.zlhip_synchronize(self._e), "synchronize")

    def bounce(self, nblocks: int, nframes: int, clocks, fmt: str = "f32", sub_blocks: int = 0, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Offline bounce to HOST memory (zlhip_bounce): "f32" -> float32 [num_buses, 2, nblocks*nframes]; "pcm16" -> int16
        [num_buses, nblocks*nframes, 2], the data chunk of one 16-bit stereo WAV per bus.  Without `out` the result lives in
        page-locked memory owned by the returned array."""
        pcm = {"f32": False, "pcm16": True}[fmt]
        shape = (self.num_buses, nblocks * nframes, 2) if pcm else (self.num_buses, 2, nblocks * nframes)
        dtype = np.int16 if pcm else np.float32
        if out is None:
            out = pinned_array(self._lib, shape, dtype)
        if out.shape != shape or out.dtype != dtype or not out.flags.c_contiguous:
            raise ValueError(f"bounce: out must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
        self._ck(self._lib.zlhip_bounce(self._e, nblocks, nframes, C.cast(clocks, C.c_void_p), out.ctypes.data, 1 if pcm else 0, sub_blocks), "bounce")
        # what read_bus / block_peaks / levels_tick see afterwards: the last chunk (the split of zlhip_bounce)
        sub = min(sub_blocks if sub_blocks > 0 else self.cfg.max_batch_blocks, self.cfg.max_batch_blocks, nblocks)
        self._last = (nblocks - sub * ((nblocks - 1) // sub), nframes)
        return out

    def read_bus(self) -> np.ndarray:
        K, N = self._last
        out = np.empty((self.num_buses, 2, K * N), dtype=np.float32)
        self._ck(self._lib.zlhip_read_bus(self._e, out.ctypes.data, out.size), "read_bus")
        return out

    def voice_reports(self):
        arr = (VoiceReport * self.num_voices)()
        self._ck(self._lib.zlhip_voice_reports(self._e, arr, self.num_voices), "voice_reports")
        return arr

    def enable_trace(self, enable: bool = True, force_slow: bool = False, no_periodic: bool = False):
        self._ck(self._lib.zlhip_debug_enable_trace(self._e, (1 if enable else 0) | (2 if force_slow else 0) | (4 if no_periodic else 0)), "enable_trace")

    def read_trace(self) -> np.ndarray:
        K, N = self._last
        out = np.empty((K, self.num_voices, N), dtype=np.int32)
        self._ck(self._lib.zlhip_debug_read_trace(self._e, out.ctypes.data, out.size), "read_trace")
        return out

    # -- levels -----------------------------------------------------------------------------
    def levels_tick(self, block_index: int = -1, with_hold_bus: int = -1):
        arr = (Levels * self.num_buses)()
        self._ck(self._lib.zlhip_levels_tick(self._e, block_index, with_hold_bus, arr), "levels_tick")
        return arr

    def block_peaks(self) -> np.ndarray:
        K, _ = self._last
        out = np.empty((K, self.num_buses, 2), dtype=np.int32)
        self._ck(self._lib.zlhip_block_peaks(self._e, out.ctypes.data, out.size), "block_peaks")
        return out

    def levels_scan_device(self, bus_dev_ptr: int, nblocks: int, nframes: int, stream: Optional[int] = None):
        self._ck(self._lib.zlhip_levels_scan_device(self._e, bus_dev_ptr, nblocks, nframes, stream), "levels_scan_device")
        self._last = (nblocks, nframes)

    # -- multi-GPU exchange (a bus that spans GPUs) -------------------------------------------
    def bus_reduce_sum_scan(self, pieces_dev_ptr: int, npieces: int, piece_stride_floats: int, units: int, nframes: int,
                            sum_out_dev_ptr: int, levels_out_dev_ptr: int, stream: Optional[int] = None):
        """Sum `npieces` received bus pieces in piece (= rank) order and scan the result for AudioLevels, one kernel."""
        self._ck(self._lib.zlhip_bus_reduce_sum_scan(self._e, pieces_dev_ptr, npieces, piece_stride_floats, units, nframes,
                                                     sum_out_dev_ptr, levels_out_dev_ptr, stream), "bus_reduce_sum_scan")

    def levels_import_units(self, units_dev_ptr: int, nblocks: int, nframes: int, stream: Optional[int] = None):
        """Unit levels of the whole bus ([bus][channel][block]) -> the block levels levels_tick / block_peaks read."""
        self._ck(self._lib.zlhip_levels_import_units(self._e, units_dev_ptr, nblocks, nframes, stream), "levels_import_units")
        self._last = (nblocks, nframes)

    # -- passthrough ------------------------------------------------------------------------
    def passthrough(self, params: Sequence[PassthroughParams], in_dev_ptr: int, out_dev_ptr: int, frames: int, stream: Optional[int] = None):
        arr = (PassthroughParams * self.num_buses)(*params)
        self._ck(self._lib.zlhip_passthrough_process(self._e, arr, in_dev_ptr, out_dev_ptr, frames, stream), "passthrough")

    # -- measurement ------------------------------------------------------------------------
    def set_profiling(self, on: bool = True):
        self._ck(self._lib.zlhip_set_profiling(self._e, 1 if on else 0), "set_profiling")

    def last_timings(self) -> Timings:
        t = Timings()
        self._ck(self._lib.zlhip_last_timings(self._e, C.byref(t)), "last_timings")
        return t

    def profile_totals(self, reset: bool = False):
        """(sums of the timings of the profiled calls since the last reset, number of calls); waits for them."""
        t = Timings()
        n = C.c_int32(0)
        self._ck(self._lib.zlhip_profile_totals(self._e, C.byref(t), C.byref(n), 1 if reset else 0), "profile_totals")
        return t, n.value

    def memory_bytes(self):
        """(HBM bytes the engine allocated at creation, the source arena's share of them)."""
        t, a = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._lib.zlhip_memory_bytes(self._e, C.byref(t), C.byref(a)), "memory_bytes")
        return t.value, a.value

    def rt_stats(self):
        """(launches of the resident real-time kernel, cycles it rendered)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._lib.zlhip_rt_stats(self._e, C.byref(a), C.byref(b)), "rt_stats")
        return a.value, b.value

    def rt_last_cycle(self):
        """Where the last real-time cycle spent its time (engines created with ZL_RT_TRACE=1): an _abi.RtCycleTrace."""
        from ._abi import RtCycleTrace
        t = RtCycleTrace()
        self._ck(self._lib.zlhip_rt_last_cycle(self._e, C.byref(t)), "rt_last_cycle")
        return t

    def rt_residency(self):
        """(is the resident real-time kernel on the device right now, the share of the device's resident capacity it takes)"""
        r, sh = C.c_int32(0), C.c_double(0.0)
        self._ck(self._lib.zlhip_rt_residency(self._e, C.byref(r), C.byref(sh)), "rt_residency")
        return bool(r.value), sh.value

    def bus_device_ptr(self) -> int:
        return self._lib.zlhip_bus_device_ptr(self._e)
