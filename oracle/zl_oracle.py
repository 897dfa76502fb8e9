"""ctypes binding of the CPU oracle (oracle/zl_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package libzl_amd never does.  Parity status: "parity unpinned" (see zl_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")

MODE_FAITHFUL, MODE_FIX_GAIN, MODE_FIX_DELAY, MODE_HERMITE = 0, 1, 2, 4
MAX_SLICES = 128
POSITION_COUNT = 32


class AdsrParams(C.Structure):
    _fields_ = [("attack", C.c_float), ("decay", C.c_float), ("sustain", C.c_float), ("release", C.c_float)]


class Adsr(C.Structure):
    _fields_ = [("p", AdsrParams), ("sampleRate", C.c_double), ("env", C.c_float), ("state", C.c_int32),
                ("attackRate", C.c_float), ("decayRate", C.c_float), ("releaseRate", C.c_float)]


class Sound(C.Structure):
    _fields_ = [("L", C.POINTER(C.c_float)), ("R", C.POINTER(C.c_float)), ("length", C.c_int32),
                ("valid", C.c_int32), ("sampleRate", C.c_double)]


class Position(C.Structure):
    _fields_ = [("id", C.c_int64), ("progress", C.c_float), ("gain", C.c_float), ("lastUpdated", C.c_int64)]


class Positions(C.Structure):
    _fields_ = [("pos", Position * POSITION_COUNT), ("updatePeakGain", C.c_int32), ("peakGain", C.c_float)]


class Clip(C.Structure):
    _fields_ = [
        ("startPositionInSeconds", C.c_float), ("lengthInSeconds", C.c_float), ("lengthInBeats", C.c_float),
        ("volumeAbsolute", C.c_float), ("pan", C.c_float), ("duration", C.c_float),
        ("rootNote", C.c_int32), ("slices", C.c_int32), ("nSlicePositions", C.c_int32),
        ("sliceBaseMidiNote", C.c_int32), ("keyZoneStart", C.c_int32), ("keyZoneEnd", C.c_int32), ("id", C.c_int32),
        ("slicePositions", C.c_double * MAX_SLICES), ("adsr", Adsr), ("positions", Positions),
    ]


class ClipMeter(C.Structure):
    _fields_ = [("currentLeveldB", C.c_double), ("prevLeveldB", C.c_double), ("firstPositionProgress", C.c_double),
                ("nextPositionUpdateTime", C.c_int64), ("nextGainUpdateTime", C.c_int64)]


class ClipCommand(C.Structure):
    _fields_ = [
        ("clip", C.c_int32), ("midiNote", C.c_int32), ("midiChannel", C.c_int32),
        ("startPlayback", C.c_int32), ("stopPlayback", C.c_int32),
        ("changeSlice", C.c_int32), ("slice", C.c_int32),
        ("changeLooping", C.c_int32), ("looping", C.c_int32),
        ("changePitch", C.c_int32), ("pitchChange", C.c_float),
        ("changeSpeed", C.c_int32), ("speedRatio", C.c_float),
        ("changeGainDb", C.c_int32), ("gainDb", C.c_float),
        ("changeVolume", C.c_int32), ("volume", C.c_float),
    ]


class Clock(C.Structure):
    _fields_ = [("current_usecs", C.c_uint64), ("next_usecs", C.c_uint64), ("jackPlayhead", C.c_uint64),
                ("jackPlayheadUsecs", C.c_uint64), ("jackSubbeatLengthInMicroseconds", C.c_uint64)]


class Step(C.Structure):
    _fields_ = [("clipCommands", C.POINTER(ClipCommand)), ("nClipCommands", C.c_int32), ("capClipCommands", C.c_int32),
                ("bpmCommands", C.POINTER(C.c_int32)), ("nBpmCommands", C.c_int32), ("capBpmCommands", C.c_int32), ("played", C.c_int32)]


class Dispatch(C.Structure):
    _fields_ = [("cmd", ClipCommand), ("tick", C.c_uint64)]


class SyncTimer(C.Structure):
    _fields_ = [("stepRing", C.POINTER(Step)), ("stepReadHead", C.c_uint64), ("stepNextPlaybackPosition", C.c_uint64),
                ("bpm", C.c_uint64), ("threadPaused", C.c_int32), ("isPaused", C.c_int32), ("jackPlayhead", C.c_uint64),
                ("jackPlayheadBpm", C.c_double), ("jackNextPlaybackPosition", C.c_uint64),
                ("jackSubbeatLengthInMicroseconds", C.c_uint64), ("jackLatency", C.c_uint64), ("scheduleAheadAmount", C.c_uint64),
                ("cumulativeBeat", C.c_uint64), ("beat", C.c_int32), ("stepReadHeadOnStart", C.c_uint64)]


class Voice(C.Structure):
    _fields_ = [
        ("hasCommand", C.c_int32), ("cmd", ClipCommand), ("clip", C.c_int32), ("clipPositionId", C.c_int64),
        ("startTick", C.c_uint64), ("nextLoopTick", C.c_uint64), ("nextLoopUsecs", C.c_uint64),
        ("pitchRatio", C.c_double), ("sourceSamplePosition", C.c_double), ("sourceSampleLength", C.c_double),
        ("lgain", C.c_float), ("rgain", C.c_float), ("adsr", Adsr), ("sound", C.c_int32), ("isPlaying", C.c_int32),
    ]


class Report(C.Structure):
    _fields_ = [("valid", C.c_int32), ("gain", C.c_float), ("progress", C.c_float)]


class Channel(C.Structure):
    _fields_ = [("voices", C.POINTER(Voice)), ("nvoices", C.c_int32), ("midiChannel", C.c_int32), ("enabled", C.c_int32)]


class LevelsChannel(C.Structure):
    _fields_ = [("peakA", C.c_int32), ("peakB", C.c_int32), ("peakAHoldSignal", C.c_float), ("peakBHoldSignal", C.c_float),
                ("peakDbA", C.c_float), ("peakDbB", C.c_float), ("combinedDb", C.c_float), ("holdDbA", C.c_float), ("holdDbB", C.c_float)]


class Passthrough(C.Structure):
    _fields_ = [("dryAmount", C.c_float), ("wetFx1Amount", C.c_float), ("wetFx2Amount", C.c_float),
                ("panAmount", C.c_float), ("muted", C.c_int32)]


_FP = C.POINTER(C.c_float)
_SIGS = {
    "zlo_adsr_init": (None, [C.POINTER(Adsr)]),
    "zlo_adsr_default_params": (None, [C.POINTER(AdsrParams)]),
    "zlo_adsr_set_sample_rate": (None, [C.POINTER(Adsr), C.c_double]),
    "zlo_adsr_set_parameters": (None, [C.POINTER(Adsr), C.POINTER(AdsrParams)]),
    "zlo_adsr_reset": (None, [C.POINTER(Adsr)]),
    "zlo_adsr_note_on": (None, [C.POINTER(Adsr)]),
    "zlo_adsr_note_off": (None, [C.POINTER(Adsr)]),
    "zlo_adsr_next": (C.c_float, [C.POINTER(Adsr)]),
    "zlo_adsr_is_active": (C.c_int, [C.POINTER(Adsr)]),
    "zlo_positions_init": (None, [C.POINTER(Positions)]),
    "zlo_positions_create": (C.c_int64, [C.POINTER(Positions), C.c_float, C.c_int64]),
    "zlo_positions_set_gain_and_progress": (None, [C.POINTER(Positions), C.c_int64, C.c_float, C.c_float, C.c_int64]),
    "zlo_positions_remove": (None, [C.POINTER(Positions), C.c_int64, C.c_int64]),
    "zlo_positions_peak_gain": (C.c_float, [C.POINTER(Positions)]),
    "zlo_positions_first_progress": (C.c_double, [C.POINTER(Positions)]),
    "zlo_positions_cleanup": (C.c_int, [C.POINTER(Positions), C.c_int64]),
    "zlo_clip_init": (None, [C.POINTER(Clip), C.c_float, C.c_double]),
    "zlo_clip_get_start_position": (C.c_float, [C.POINTER(Clip), C.c_int]),
    "zlo_clip_get_stop_position": (C.c_float, [C.POINTER(Clip), C.c_int]),
    "zlo_clip_set_start_position": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_clip_set_length": (None, [C.POINTER(Clip), C.c_float, C.c_int]),
    "zlo_clip_set_volume_absolute": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_clip_set_pan": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_clip_set_slices": (None, [C.POINTER(Clip), C.c_int]),
    "zlo_clip_slice_for_midi_note": (C.c_int, [C.POINTER(Clip), C.c_int]),
    "zlo_clip_set_adsr_attack": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_clip_set_adsr_decay": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_clip_set_adsr_sustain": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_clip_set_adsr_release": (None, [C.POINTER(Clip), C.c_float]),
    "zlo_subbeat_count_to_seconds": (C.c_float, [C.c_uint64, C.c_uint64]),
    "zlo_clip_meter_init": (None, [C.POINTER(ClipMeter)]),
    "zlo_sync_audio_level": (C.c_int, [C.POINTER(ClipMeter), C.POINTER(Clip), C.c_int64, C.POINTER(C.c_float)]),
    "zlo_sync_progress": (C.c_int, [C.POINTER(ClipMeter), C.POINTER(Clip), C.c_int, C.c_int64, C.POINTER(C.c_float)]),
    "zlo_clip_command_clear": (None, [C.POINTER(ClipCommand)]),
    "zlo_clip_command_equivalent": (C.c_int, [C.POINTER(ClipCommand), C.POINTER(ClipCommand)]),
    "zlo_sync_timer_new": (C.POINTER(SyncTimer), []),
    "zlo_sync_timer_free": (None, [C.POINTER(SyncTimer)]),
    "zlo_step_schedule": (C.c_int, [C.POINTER(ClipCommand), C.POINTER(C.c_int32), C.POINTER(ClipCommand)]),
    "zlo_schedule_clip_command": (None, [C.POINTER(SyncTimer), C.POINTER(ClipCommand), C.c_uint64]),
    "zlo_sync_timer_set_latency": (None, [C.POINTER(SyncTimer), C.c_uint32, C.c_double]),
    "zlo_sync_timer_set_bpm": (None, [C.POINTER(SyncTimer), C.c_uint64]),
    "zlo_sync_timer_start": (None, [C.POINTER(SyncTimer), C.c_int]),
    "zlo_sync_timer_stop": (None, [C.POINTER(SyncTimer)]),
    "zlo_sync_timer_callback": (None, [C.POINTER(SyncTimer)]),
    "zlo_sync_timer_queue_clip_to_start_on_channel": (None, [C.POINTER(SyncTimer), C.c_int32, C.c_int]),
    "zlo_sync_timer_queue_clip_to_stop_on_channel": (None, [C.POINTER(SyncTimer), C.c_int32, C.c_int]),
    "zlo_sync_timer_process": (C.c_int32, [C.POINTER(SyncTimer), C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.POINTER(Dispatch), C.c_int32]),
    "zlo_sync_timer_jack_playhead": (C.c_uint64, [C.POINTER(SyncTimer)]),
    "zlo_sync_timer_jack_playhead_usecs": (C.c_uint64, [C.POINTER(SyncTimer)]),
    "zlo_sync_timer_jack_subbeat_length_usecs": (C.c_uint64, [C.POINTER(SyncTimer)]),
    "zlo_voice_init": (None, [C.POINTER(Voice)]),
    "zlo_voice_set_current_command": (C.c_int, [C.POINTER(Voice), C.POINTER(ClipCommand), C.POINTER(Clip), C.POINTER(Sound)]),
    "zlo_voice_start_note": (None, [C.POINTER(Voice), C.c_int, C.c_float, C.c_int, C.POINTER(Sound), C.POINTER(Clip), C.c_double, C.c_int64]),
    "zlo_voice_stop_note": (None, [C.POINTER(Voice), C.c_int, C.POINTER(Clip), C.c_int64]),
    "zlo_voice_process": (None, [C.POINTER(Voice), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Clock), C.POINTER(Sound), C.POINTER(Clip),
                                 C.c_uint32, C.c_int64, C.POINTER(Report), C.c_void_p]),
    "zlo_channel_handle_command": (C.c_int, [C.POINTER(Channel), C.POINTER(ClipCommand), C.c_uint64, C.POINTER(Sound), C.POINTER(Clip), C.c_double, C.c_int64]),
    "zlo_channel_process": (None, [C.POINTER(Channel), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Clock), C.POINTER(Sound), C.POINTER(Clip),
                                   C.c_uint32, C.c_int64, C.POINTER(Report)]),
    "zlo_convert_to_dbfs": (C.c_float, [C.c_float]),
    "zlo_add_float_db": (C.c_float, [C.c_float, C.c_float]),
    "zlo_sample_to_peak_int": (C.c_int32, [C.c_float]),
    "zlo_levels_tick": (None, [C.POINTER(LevelsChannel), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "zlo_block_sumsq": (C.c_float, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "zlo_block_rms": (C.c_float, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "zlo_pcm16_sample": (C.c_int16, [C.c_float]),
    "zlo_pcm16_stereo": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zlo_passthrough_init": (None, [C.POINTER(Passthrough)]),
    "zlo_passthrough_process": (None, [C.POINTER(Passthrough), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32]),
    "zlo_render_batch": (None, [C.POINTER(Channel), C.c_int32, C.POINTER(Sound), C.POINTER(Clip), C.POINTER(Clock), C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Report), C.c_int32]),
    "zlo_render_batch_at": (None, [C.POINTER(Channel), C.c_int32, C.POINTER(Sound), C.POINTER(Clip), C.POINTER(Clock), C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Report), C.c_int32, C.c_int64]),
}


def _bind(lib):
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_libs = {}


def _make(target: str):
    res = subprocess.run(["make", "-C", HERE, target], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)


def load(fast: bool = False):
    """Parity build (-O2 -ffp-contract=off) or, with fast=True, the -O3 -march=native timing build.
    The timing build is compiled on the machine it runs on (per-CPU file name)."""
    key = "fast" if fast else "parity"
    if key in _libs:
        return _libs[key]
    if not fast:
        path = os.path.join(BUILD, "libzl_oracle.so")
        src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("zl_oracle.c", "zl_oracle.h"))
        if not os.path.exists(path) or os.path.getmtime(path) < src_m:
            _make("_build/libzl_oracle.so")
    else:
        try:
            flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags"))
        except Exception:
            flags = "unknown"
        tag = hashlib.sha1(flags.encode()).hexdigest()[:10]
        path = os.path.join(BUILD, f"libzl_oracle_fast.{tag}.so")
        src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("zl_oracle.c", "zl_oracle.h"))
        if not os.path.exists(path) or os.path.getmtime(path) < src_m:
            os.makedirs(BUILD, exist_ok=True)
            cmd = ["gcc", "-std=c11", "-O3", "-march=native", "-fPIC", "-shared", "-o", path,
                   os.path.join(HERE, "zl_oracle.c"), "-lm", "-lpthread"]
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("oracle fast build failed:\n" + res.stderr)
    _libs[key] = _bind(C.CDLL(path))
    return _libs[key]


class OracleSynth:
    """The reference's SamplerSynth (channels x voices) on the CPU oracle, driven like the engine."""

    def __init__(self, num_buses=12, voices_per_bus=8, playback_sample_rate=48000.0, mode=MODE_FAITHFUL,
                 max_sounds=1024, mix_group=0, fast=False):
        self.lib = load(fast=fast)
        self.B, self.VPB, self.fs, self.mode, self.mix_group = num_buses, voices_per_bus, float(playback_sample_rate), mode, mix_group
        self.sounds = (Sound * max_sounds)()
        self.clips = (Clip * max_sounds)()
        self._buffers = []           # keep numpy sources alive
        self.nsounds = 0
        self.voices = (Voice * (num_buses * voices_per_bus))()
        for v in self.voices:
            self.lib.zlo_voice_init(C.byref(v))
        self.channels = (Channel * num_buses)()
        for b in range(num_buses):
            ch = self.channels[b]
            ch.voices = C.cast(C.byref(self.voices, b * voices_per_bus * C.sizeof(Voice)), C.POINTER(Voice))
            ch.nvoices = voices_per_bus
            ch.midiChannel = b - 2          # SamplerSynth.cpp:270
            ch.enabled = 1
        self.now_ms = 0

    def register_clip(self, left, right, sample_rate):
        i = self.nsounds
        left = np.ascontiguousarray(left, dtype=np.float32)
        self._buffers.append(left)
        s = self.sounds[i]
        s.L = left.ctypes.data_as(_FP)
        if right is not None:
            right = np.ascontiguousarray(right, dtype=np.float32)
            self._buffers.append(right)
            s.R = right.ctypes.data_as(_FP)
        else:
            s.R = None
        s.length = left.shape[0]
        s.valid = 1
        s.sampleRate = float(sample_rate)
        # getDuration(): the edit length in seconds as float (file length / sample rate)
        self.lib.zlo_clip_init(C.byref(self.clips[i]), C.c_float(left.shape[0] / float(sample_rate)), float(sample_rate))
        self.clips[i].id = i
        self.nsounds += 1
        return i

    def clip(self, i) -> Clip:
        return self.clips[i]

    def handle_clip_command(self, cmd: ClipCommand, current_tick=0):
        bus = cmd.midiChannel + 2           # SamplerSynth.cpp:330-331
        if bus < 0 or bus >= self.B or cmd.clip < 0 or cmd.clip >= self.nsounds:
            return 0
        return self.lib.zlo_channel_handle_command(C.byref(self.channels[bus]), C.byref(cmd), current_tick, self.sounds, self.clips, self.fs, self.now_ms)

    def start_voice(self, bus, slot, cmd: ClipCommand, current_tick=0):
        """Engine extension zlhip_start_voice: start on an explicit slot (stop handling as handleCommand)."""
        ch = self.channels[bus]
        if cmd.stopPlayback:
            for i in range(self.VPB):
                v = ch.voices[i]
                if v.sound >= 0 and v.sound == cmd.clip and v.hasCommand and self.lib.zlo_clip_command_equivalent(C.byref(v.cmd), C.byref(cmd)):
                    self.lib.zlo_voice_stop_note(C.byref(v), 1, self.clips, self.now_ms)
        v = ch.voices[slot]
        if cmd.startPlayback and not v.isPlaying:
            self.lib.zlo_voice_set_current_command(C.byref(v), C.byref(cmd), self.clips, self.sounds)
            v.startTick = current_tick
            self.lib.zlo_voice_start_note(C.byref(v), cmd.midiNote, cmd.volume, cmd.clip, self.sounds, self.clips, self.fs, self.now_ms)
            return 1
        return 0

    def set_bus_enabled(self, bus, enabled):
        """SamplerSynth::setChannelEnabled (SamplerSynth.cpp:343-351)"""
        self.channels[bus].enabled = 1 if enabled else 0

    def update_voice(self, bus, slot, cmd: ClipCommand):
        """SamplerSynthVoice::setCurrentCommand on a playing voice (zlhip_update_voice; SamplerSynthVoice.cpp:58-100)"""
        v = self.channels[bus].voices[slot]
        if not v.isPlaying:
            return 0
        self.lib.zlo_voice_set_current_command(C.byref(v), C.byref(cmd), self.clips, self.sounds)
        return 1

    def stop_voice(self, bus, slot, allow_tail_off=True):
        """SamplerSynthVoice::stopNote on one voice (zlhip_stop_voice; SamplerSynthVoice.cpp:146-169)"""
        v = self.channels[bus].voices[slot]
        if not v.isPlaying:
            return 0
        self.lib.zlo_voice_stop_note(C.byref(v), 1 if allow_tail_off else 0, self.clips, self.now_ms)
        return 1

    def render_batch(self, nblocks, nframes, clocks, threads=1, want_reports=True):
        bus = np.zeros((self.B, 2, nblocks * nframes), dtype=np.float32)
        # zlo_render_batch wants busL/busR as [B][nblocks*nframes] planes
        busL = np.zeros((self.B, nblocks * nframes), dtype=np.float32)
        busR = np.zeros((self.B, nblocks * nframes), dtype=np.float32)
        reports = (Report * (self.B * self.VPB))()
        oclocks = (Clock * nblocks)()
        for k in range(nblocks):
            c = clocks[k]
            oclocks[k].current_usecs = c.current_usecs
            oclocks[k].next_usecs = c.next_usecs
            oclocks[k].jackPlayhead = getattr(c, "jack_playhead", getattr(c, "jackPlayhead", 0))
            oclocks[k].jackPlayheadUsecs = getattr(c, "jack_playhead_usecs", getattr(c, "jackPlayheadUsecs", 0))
            oclocks[k].jackSubbeatLengthInMicroseconds = getattr(c, "jack_subbeat_length_usecs", getattr(c, "jackSubbeatLengthInMicroseconds", 0))
        self.lib.zlo_render_batch_at(self.channels, self.B, self.sounds, self.clips, oclocks, nblocks, nframes, self.mode,
                                     self.mix_group, busL.ctypes.data, busR.ctypes.data, reports if want_reports else None, threads, self.now_ms)
        bus[:, 0, :] = busL
        bus[:, 1, :] = busR
        return bus, reports

    def voice_trace(self, voice_index, nframes, clock):
        """Render ONE block of one voice alone and return (L, R, pos_trace); advances that voice."""
        L = np.zeros(nframes, dtype=np.float32)
        R = np.zeros(nframes, dtype=np.float32)
        tr = np.zeros(nframes, dtype=np.int32)
        oc = Clock(clock.current_usecs, clock.next_usecs,
                   getattr(clock, "jack_playhead", 0), getattr(clock, "jack_playhead_usecs", 0), getattr(clock, "jack_subbeat_length_usecs", 0))
        rep = Report()
        self.lib.zlo_voice_process(C.byref(self.voices[voice_index]), L.ctypes.data, R.ctypes.data, nframes, C.byref(oc),
                                   self.sounds, self.clips, self.mode, self.now_ms, C.byref(rep), tr.ctypes.data)
        return L, R, tr, rep


def clip_command(**fields) -> ClipCommand:
    c = ClipCommand()
    c.clip = -1
    c.midiNote = -1
    c.midiChannel = -1
    c.slice = -1
    for k, v in fields.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


CMD_FIELDS = [f for f, _ in ClipCommand._fields_]


def cmd_tuple(c: ClipCommand):
    return tuple(getattr(c, f) for f in CMD_FIELDS)


class OracleSyncTimer:
    """SyncTimer's ClipCommand step ring (zlo_sync_timer_*), driven like libzl_hotpath_cycle drives the product's."""

    def __init__(self, fast: bool = False):
        self.lib = load(fast=fast)
        self.t = self.lib.zlo_sync_timer_new()
        self._out = (Dispatch * 4096)()

    def close(self):
        if self.t:
            self.lib.zlo_sync_timer_free(self.t)
            self.t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def schedule(self, cmd: ClipCommand, delay: int = 0):
        self.lib.zlo_schedule_clip_command(self.t, C.byref(cmd), delay)

    def set_latency(self, buffer_size, sample_rate): self.lib.zlo_sync_timer_set_latency(self.t, buffer_size, sample_rate)
    def set_bpm(self, bpm): self.lib.zlo_sync_timer_set_bpm(self.t, bpm)
    def start(self, bpm): self.lib.zlo_sync_timer_start(self.t, bpm)
    def stop(self): self.lib.zlo_sync_timer_stop(self.t)
    def timer_callback(self): self.lib.zlo_sync_timer_callback(self.t)
    def queue_start(self, clip, channel): self.lib.zlo_sync_timer_queue_clip_to_start_on_channel(self.t, clip, channel)
    def queue_stop(self, clip, channel): self.lib.zlo_sync_timer_queue_clip_to_stop_on_channel(self.t, clip, channel)

    def process(self, nframes, current_usecs, next_usecs, period_usecs=None):
        """-> [(ClipCommand copy, tick)] of the steps that fell due in this cycle"""
        if period_usecs is None:
            period_usecs = float(next_usecs - current_usecs)
        n = self.lib.zlo_sync_timer_process(self.t, nframes, current_usecs, next_usecs, C.c_float(period_usecs), self._out, len(self._out))
        assert n <= len(self._out)
        res = []
        for i in range(n):
            c = ClipCommand()
            C.memmove(C.byref(c), C.byref(self._out[i].cmd), C.sizeof(ClipCommand))
            res.append((c, int(self._out[i].tick)))
        return res

    def clock(self, current_usecs, next_usecs) -> Clock:
        """the clock a voice reads in this cycle: JACK cycle times + SyncTimer's getters (SyncTimer.cpp:990-1009)"""
        return Clock(current_usecs, next_usecs, self.lib.zlo_sync_timer_jack_playhead(self.t),
                     self.lib.zlo_sync_timer_jack_playhead_usecs(self.t), self.lib.zlo_sync_timer_jack_subbeat_length_usecs(self.t))
