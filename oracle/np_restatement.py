"""Independent restatement of libzl's sampler hot path in numpy scalar arithmetic.  TEST INFRASTRUCTURE.

Written from the reference text (paths under /root/reference/lib), not from oracle/zl_oracle.c, so the
two can be cross-checked: every float operation is an explicit np.float32 / np.float64 scalar
operation in the order the reference's C++ expressions associate.  Slow (pure Python loops): used on
small cases only and to generate tests/golden/*.npz.  Parity status: "parity unpinned" (the
reference has no golden vectors and cannot be built here; see oracle/zl_oracle.h).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

f32 = np.float32
f64 = np.float64

IDLE, ATTACK, DECAY, SUSTAIN, RELEASE = range(5)
BEAT_SUBDIVISIONS = 96            # SyncTimer.cpp:95
MODE_FIX_GAIN, MODE_FIX_DELAY, MODE_HERMITE = 1, 2, 4


def fma32(a, b, c):
    """fmaf: a * b + c with ONE rounding to float32.  The product of two float32 is exact in float64; the sum is rounded
    to odd in float64 (error term from TwoSum), after which the rounding to float32 is the correctly rounded result
    (53 >= 2 * 24 + 2 bits: Boldo & Melquiond, "Emulation of FMA and correctly rounded sums")."""
    import struct
    p = float(a) * float(b)
    cc = float(c)
    s = p + cc
    if s != s or s in (float("inf"), float("-inf")):
        return np.float32(s)
    bb = s - p
    err = (p - (s - bb)) + (cc - bb)
    if err != 0.0:
        bits = struct.unpack("<q", struct.pack("<d", s))[0]
        if (bits & 1) == 0:                                  # make the last bit odd, towards the exact value
            bits += 1 if (err > 0.0) == (s > 0.0) else -1
            s = struct.unpack("<d", struct.pack("<q", bits))[0]
    return np.float32(s)


def _u64(x: int) -> int:
    return x & 0xFFFFFFFFFFFFFFFF


def _f_to_u64(x) -> int:
    """float -> quint64 as aarch64 fcvtzu (the reference's target): saturating, NaN -> 0."""
    x = float(x)
    if not (x > 0.0):
        return 0
    if x >= 18446744073709551616.0:
        return 0xFFFFFFFFFFFFFFFF
    return int(x)


# ---------------------------------------------------------------- juce::ADSR (JUCE 6/7 juce_ADSR.h)
class ADSR:
    def __init__(self):
        self.attack, self.decay, self.sustain, self.release = f32(0.1), f32(0.1), f32(1.0), f32(0.1)
        self.sr = f64(44100.0)
        self.env = f32(0.0)
        self.state = IDLE
        self._recalc()

    @staticmethod
    def _rate(distance, t, sr):
        return f32(f64(distance) / (f64(t) * sr)) if t > f32(0.0) else f32(-1.0)

    def _next_state(self):
        if self.state == ATTACK:
            self.state = DECAY if self.decay_rate > f32(0.0) else SUSTAIN
        elif self.state == DECAY:
            self.state = SUSTAIN
        elif self.state == RELEASE:
            self.reset()

    def _recalc(self):
        self.attack_rate = self._rate(f32(1.0), self.attack, self.sr)
        self.decay_rate = self._rate(f32(1.0) - self.sustain, self.decay, self.sr)
        self.release_rate = self._rate(self.sustain, self.release, self.sr)
        if ((self.state == ATTACK and self.attack_rate <= f32(0.0))
                or (self.state == DECAY and (self.decay_rate <= f32(0.0) or self.env <= self.sustain))
                or (self.state == RELEASE and self.release_rate <= f32(0.0))):
            self._next_state()

    def set_sample_rate(self, sr):
        self.sr = f64(sr)

    def set_parameters(self, a, d, s, r):
        self.attack, self.decay, self.sustain, self.release = f32(a), f32(d), f32(s), f32(r)
        self._recalc()

    def reset(self):
        self.env = f32(0.0)
        self.state = IDLE

    def note_on(self):
        if self.attack_rate > f32(0.0):
            self.state = ATTACK
        elif self.decay_rate > f32(0.0):
            self.env = f32(1.0)
            self.state = DECAY
        else:
            self.env = self.sustain
            self.state = SUSTAIN

    def note_off(self):
        if self.state != IDLE:
            if self.release > f32(0.0):
                self.release_rate = f32(f64(self.env) / (f64(self.release) * self.sr))
                self.state = RELEASE
            else:
                self.reset()

    def next(self):
        if self.state == IDLE:
            return f32(0.0)
        if self.state == ATTACK:
            self.env = f32(self.env + self.attack_rate)
            if self.env >= f32(1.0):
                self.env = f32(1.0)
                self._next_state()
        elif self.state == DECAY:
            self.env = f32(self.env - self.decay_rate)
            if self.env <= self.sustain:
                self.env = self.sustain
                self._next_state()
        elif self.state == SUSTAIN:
            self.env = self.sustain
        elif self.state == RELEASE:
            self.env = f32(self.env - self.release_rate)
            if self.env <= f32(0.0):
                self._next_state()
        return self.env

    def active(self):
        return self.state != IDLE


# ---------------------------------------------------------------- clip / sound / command
@dataclass
class Sound:
    L: np.ndarray
    R: Optional[np.ndarray]
    sample_rate: float

    @property
    def length(self):
        return int(self.L.shape[0])


@dataclass
class Clip:
    """Fields of ClipAudioSource::Private the voice reads (ClipAudioSource.cpp:63-82)."""
    start_sec: np.float32 = f32(0.0)
    length_sec: np.float32 = f32(-1.0)
    length_beats: np.float32 = f32(-1.0)
    volume_abs: np.float32 = f32(1.0)
    pan: np.float32 = f32(0.0)
    duration: np.float32 = f32(0.0)
    root_note: int = 60
    slice_pos: List[float] = field(default_factory=list)
    slices: int = 0                                               # ClipAudioSource::Private::slices (the count setSlices compares with)
    adsr: tuple = (f32(0.0), f32(0.1), f32(1.0), f32(0.05))      # ctor :164-168

    def get_start(self, slice_):                                  # :261-268
        if -1 < slice_ < len(self.slice_pos):
            return f32(f64(self.start_sec) + (f64(self.length_sec) * f64(self.slice_pos[slice_])))
        return self.start_sec

    def get_stop(self, slice_):                                   # :270-277
        if slice_ > -1 and slice_ + 1 < len(self.slice_pos):
            return f32(f64(self.start_sec) + (f64(self.length_sec) * f64(self.slice_pos[slice_ + 1])))
        return f32(self.start_sec + self.length_sec)

    def set_length(self, beat, bpm):                              # :352-360, SyncTimer.cpp:180-183,936-939
        sub = _f_to_u64(f32(f32(beat) * f32(BEAT_SUBDIVISIONS)))
        b = min(max(int(bpm), 50), 200)
        ns = (sub * 60000000000) // (b * BEAT_SUBDIVISIONS)
        self.length_sec = f32(f32(ns) / f32(1000000000))
        self.length_beats = f32(beat)

    def set_slices(self, n):                                      # :495-528
        """ClipAudioSource::setSlices: 0 clears the table, fewer slices drop the last entries, more slices fit the NEW entries evenly
        between the last position and 1 (all double arithmetic: `1.0f - last`, divided by the int count of added slices)."""
        n = int(n)
        if self.slices == n:
            return
        if n == 0:
            self.slice_pos = []
        elif self.slices > n:
            while len(self.slice_pos) > n:
                self.slice_pos.pop()
        else:
            last = f64(self.slice_pos[-1]) if self.slice_pos else f64(0.0)
            inc = f64(f64(1.0) - last) / f64(n - self.slices)
            pos = f64(last + inc)
            if not self.slice_pos:
                self.slice_pos.append(0.0)
            while len(self.slice_pos) < n:
                self.slice_pos.append(float(pos))
                pos = f64(pos + inc)
        self.slices = n

    def set_volume_absolute(self, vol):                           # :328-336 (clamp; the fader readback taken as identity)
        self.volume_abs = f32(max(f32(0.0), min(f32(vol), f32(1.0))))

    def set_start_position(self, sec):                            # :255-259
        self.start_sec = f32(max(f32(0.0), f32(sec)))


@dataclass
class Command:
    clip: int = -1
    midi_note: int = -1
    midi_channel: int = -1
    start: bool = False
    stop: bool = False
    change_slice: bool = False
    slice: int = -1
    looping: bool = False
    change_looping: bool = False
    change_volume: bool = False
    volume: np.float32 = f32(0.0)
    change_pitch: bool = False
    pitch_change: np.float32 = f32(0.0)
    change_speed: bool = False
    speed_ratio: np.float32 = f32(0.0)
    change_gain_db: bool = False
    gain_db: np.float32 = f32(0.0)

    def equivalent(self, o):                                      # ClipCommand.h:33-39
        return self.clip == o.clip and (
            (self.change_slice and o.change_slice and self.slice == o.slice)
            or (not self.change_slice and not o.change_slice and self.midi_note == o.midi_note and self.midi_channel == o.midi_channel))


@dataclass
class Clock:
    current_usecs: int
    next_usecs: int
    playhead: int = 0
    playhead_usecs: int = 0
    subbeat_usecs: int = 0


# ---------------------------------------------------------------- SamplerSynthVoice
class Voice:
    def __init__(self):
        self.cmd: Optional[Command] = None
        self.clip: Optional[Clip] = None
        self.sound: Optional[Sound] = None
        self.is_playing = False
        self.start_tick = 0
        self.next_loop_tick = 0
        self.next_loop_usecs = 0
        self.pitch_ratio = f64(0.0)
        self.P = f64(0.0)
        self.src_len = f64(0.0)
        self.lgain = f32(0.0)
        self.rgain = f32(0.0)
        self.adsr = ADSR()

    def start_note(self, note, velocity, sound: Sound, clip: Clip, fs):     # SamplerSynthVoice.cpp:110-144
        self.sound = sound
        self.pitch_ratio = f64(math.pow(2.0, (note - clip.root_note) / 12.0)) * f64(sound.sample_rate) / f64(fs)
        self.clip = clip
        self.src_len = f64(clip.duration) * f64(sound.sample_rate)
        self.P = f64(int(f64(clip.get_start(self.cmd.slice)) * f64(sound.sample_rate)))
        self.next_loop_tick = _f_to_u64(f32(f32(self.start_tick) + f32(clip.length_beats * f32(BEAT_SUBDIVISIONS))))
        self.next_loop_usecs = 0
        self.lgain = f32(velocity)
        self.rgain = f32(velocity)
        self.adsr.reset()
        self.adsr.set_sample_rate(sound.sample_rate)
        self.adsr.set_parameters(*clip.adsr)
        self.adsr.note_on()

    def set_current_command(self, cmd):                                   # :58-100, on a voice that HAS a command: a patch
        if cmd.change_looping:
            self.cmd.looping = cmd.looping; self.cmd.change_looping = True
        if cmd.change_pitch:
            self.cmd.pitch_change = cmd.pitch_change; self.cmd.change_pitch = True
        if cmd.change_speed:
            self.cmd.speed_ratio = cmd.speed_ratio; self.cmd.change_speed = True
        if cmd.change_gain_db:
            self.cmd.gain_db = cmd.gain_db; self.cmd.change_gain_db = True
        if cmd.change_volume:
            self.cmd.volume = cmd.volume; self.cmd.change_volume = True
            self.lgain = f32(cmd.volume)
            self.rgain = f32(cmd.volume)
        if cmd.change_slice:
            self.cmd.slice = cmd.slice
        if cmd.start and self.sound is not None:                          # "restart playback": back to the start of the voice's slice (:86-91)
            self.P = f64(int(f64(self.clip.get_start(self.cmd.slice)) * f64(self.sound.sample_rate)))

    def stop_note(self, tail):                                            # :146-169
        if tail:
            self.adsr.note_off()
        else:
            self.sound = None
            self.adsr.reset()
            self.clip = None
            if self.cmd is not None:
                self.cmd = None
                self.is_playing = False
            self.next_loop_tick = 0
            self.next_loop_usecs = 0

    def process(self, L, R, nframes, clk: Clock, mode=0):                  # :174-270
        """Accumulates into L/R (np.float32 arrays of nframes).  Returns (valid, gain, progress, pos_trace)."""
        trace = [-1] * nframes
        if self.sound is None or self.cmd is None:
            return False, f32(0), f32(0), trace
        snd, clip = self.sound, self.clip
        if self.next_loop_usecs == 0:
            diff = _u64(self.next_loop_tick - clk.playhead)
            self.next_loop_usecs = _u64(clk.playhead_usecs + _u64(diff * clk.subbeat_usecs))
        upf = f64((clk.next_usecs - clk.current_usecs) // nframes)
        peak = f32(0.0)
        inL, inR = snd.L, snd.R
        vol = f32(clip.volume_abs)
        sr = f64(snd.sample_rate)
        stop_pos = int(f64(clip.get_stop(self.cmd.slice)) * sr)
        dur = snd.length - 1
        pan = f32(clip.pan)
        lpan = f32(f64(0.5) * (f64(1.0) + f64(pan)))
        rpan = f32(f64(0.5) * (f64(1.0) - f64(pan)))
        looping = self.cmd.looping
        fix_gain, fix_delay, hermite = bool(mode & 1), bool(mode & 2), bool(mode & 4)
        for frame in range(nframes):
            pos = int(self.P)
            alpha = f32(self.P - f64(pos))
            inv = f32(f32(1.0) - alpha)
            env = self.adsr.next()
            trace[frame] = pos
            inb = dur > pos
            if hermite and inb:
                wide = pos - 1 >= 0 and pos + 2 <= dur
                # Catmull-Rom in tap-weight form: four cubic weights of alpha, shared by both channels; every fma32 is one rounding
                t = f32(alpha * alpha)
                w0 = f32(fma32(fma32(f32(-0.5), alpha, f32(1.0)), alpha, f32(-0.5)) * alpha)
                w1 = fma32(fma32(f32(1.5), alpha, f32(-2.5)), t, f32(1.0))
                w2 = f32(fma32(fma32(f32(-1.5), alpha, f32(2.0)), alpha, f32(0.5)) * alpha)
                w3 = f32(fma32(f32(0.5), alpha, f32(-0.5)) * t)

                def interp(x):
                    if wide:
                        y0, y1, y2, y3 = x[pos - 1], x[pos], x[pos + 1], x[pos + 2]
                        return fma32(w3, y3, fma32(w2, y2, fma32(w1, y1, f32(w0 * y0))))
                    return f32(f32(x[pos] * inv) + f32(x[pos + 1] * alpha))
                # whole-sample gain, the gain product formed first: sample * ((gain * envelope) * volume)
                l = f32(interp(inL) * f32(f32(self.lgain * env) * vol))
                r = f32(interp(inR) * f32(f32(self.rgain * env) * vol)) if inR is not None else l
            elif fix_gain:
                l = f32(f32(f32(f32(f32(inL[pos] * inv) + f32(inL[pos + 1] * alpha)) * self.lgain) * env) * vol) if inb else f32(0)
                r = (f32(f32(f32(f32(f32(inR[pos] * inv) + f32(inR[pos + 1] * alpha)) * self.rgain) * env) * vol)
                     if (inR is not None and inb) else l)
            else:
                # :204-205 -- only the second tap carries gain * envelope * volume (Q1)
                l = (f32(f32(inL[pos] * inv) + f32(f32(f32(f32(inL[pos + 1] * alpha) * self.lgain) * env) * vol)) if inb else f32(0))
                r = (f32(f32(inR[pos] * inv) + f32(f32(f32(f32(inR[pos + 1] * alpha) * self.rgain) * env) * vol))
                     if (inR is not None and inb) else l)
            m = f32(f64(0.5) * f64(f32(l + r)))
            s = f32(l - r)
            l = f32(f32(lpan * m) + s)
            r = f32(f32(rpan * m) - s)
            g = f32(l + r)
            if g > peak:
                peak = g
            if fix_delay:
                L[frame] = f32(L[frame] + l)
                R[frame] = f32(R[frame] + r)
            elif frame + 1 < nframes:                      # Q2: pre-incremented pointers; last write is out of bounds
                L[frame + 1] = f32(L[frame + 1] + l)
                R[frame + 1] = f32(R[frame + 1] + r)
            self.P = f64(self.P + self.pitch_ratio)
            if looping:
                if f32(math.trunc(float(clip.length_beats))) == clip.length_beats:
                    if _u64(clk.current_usecs + _f_to_u64(f64(frame) * upf)) >= self.next_loop_usecs:
                        ticks = _f_to_u64(f32(clip.length_beats * f32(BEAT_SUBDIVISIONS)))
                        self.next_loop_tick = _u64(self.next_loop_tick + ticks)
                        diff = _u64(self.next_loop_tick - clk.playhead)
                        self.next_loop_usecs = _u64(clk.playhead_usecs + _u64(diff * clk.subbeat_usecs))
                        self.P = f64(int(f64(clip.get_start(self.cmd.slice)) * sr))
                elif self.P >= f64(stop_pos):
                    self.P = f64(int(f64(clip.get_start(self.cmd.slice)) * sr))
            else:
                if self.P >= f64(stop_pos):
                    self.stop_note(False)
                    break
                elif self.P >= f64(stop_pos) - (f64(self.adsr.release) * sr):
                    self.stop_note(True)
            if not self.adsr.active():
                self.stop_note(False)
                break
        if self.clip is not None:
            return True, f32(peak * f32(0.5)), f32(self.P / self.src_len), trace
        return False, f32(0), f32(0), trace


# ---------------------------------------------------------------- SamplerChannel / SamplerSynth
class Synth:
    def __init__(self, num_buses, voices_per_bus, fs, mode=0):
        self.B, self.VPB, self.fs, self.mode = num_buses, voices_per_bus, fs, mode
        self.voices = [[Voice() for _ in range(voices_per_bus)] for _ in range(num_buses)]
        self.enabled = [True] * num_buses                         # SamplerChannel::enabled (SamplerSynth.cpp:60,343-351)
        self.sounds: List[Sound] = []
        self.clips: List[Clip] = []

    def register(self, L, R, sr):
        self.sounds.append(Sound(np.asarray(L, dtype=np.float32), None if R is None else np.asarray(R, dtype=np.float32), sr))
        c = Clip()
        c.duration = f32(len(L) / sr)
        c.length_sec = c.duration
        c.set_slices(16)
        self.clips.append(c)
        return len(self.clips) - 1

    def handle(self, cmd: Command, tick=0):                               # SamplerSynth.cpp:328-341,187-230
        bus = cmd.midi_channel + 2
        if bus < 0 or bus >= self.B or not (0 <= cmd.clip < len(self.clips)):
            return
        voices = self.voices[bus]
        snd = self.sounds[cmd.clip]
        if cmd.stop or cmd.start:
            if cmd.stop:
                for v in voices:
                    if v.sound is snd and v.cmd is not None and v.cmd.equivalent(cmd):
                        v.stop_note(True)
            if cmd.start:
                for v in voices:
                    if not v.is_playing:
                        v.cmd = Command(**cmd.__dict__)
                        v.is_playing = True
                        v.start_tick = tick
                        v.start_note(cmd.midi_note, cmd.volume, snd, self.clips[cmd.clip], self.fs)
                        break
        else:
            for v in voices:
                if v.sound is snd and v.cmd is not None and v.cmd.equivalent(cmd):
                    v.set_current_command(cmd)

    # the voice-level surface (juce::SynthesiserVoice calls the sampler makes on ONE voice): bus / slot address a voice directly
    def update_voice(self, bus, slot, cmd: Command):
        v = self.voices[bus][slot]
        if v.is_playing:
            v.set_current_command(cmd)

    def stop_voice(self, bus, slot, tail):
        v = self.voices[bus][slot]
        if v.is_playing:
            v.stop_note(tail)

    def start_voice(self, bus, slot, cmd: Command, tick=0):
        """the engine's zlhip_start_voice: handleCommand's start with the slot given (stop handling as handleCommand)"""
        if not (0 <= cmd.clip < len(self.clips)):
            return
        snd = self.sounds[cmd.clip]
        if cmd.stop:
            for v in self.voices[bus]:
                if v.sound is snd and v.cmd is not None and v.cmd.equivalent(cmd):
                    v.stop_note(True)
        v = self.voices[bus][slot]
        if cmd.start and not v.is_playing:
            v.cmd = Command(**cmd.__dict__)
            v.is_playing = True
            v.start_tick = tick
            v.start_note(cmd.midi_note, cmd.volume, snd, self.clips[cmd.clip], self.fs)

    def process(self, nframes, clk: Clock):                               # SamplerSynth.cpp:116-148
        L = np.zeros((self.B, nframes), dtype=np.float32)
        R = np.zeros((self.B, nframes), dtype=np.float32)
        reports = {}
        for b in range(self.B):
            if not self.enabled[b]:                               # SamplerSynth.cpp:123: a disabled channel processes no voice
                continue
            for i, v in enumerate(self.voices[b]):
                if v.is_playing:
                    reports[(b, i)] = v.process(L[b], R[b], nframes, clk, self.mode)
        return L, R, reports


# ---------------------------------------------------------------- SyncTimer: ClipCommands on the step ring
# Written from /root/reference/lib/SyncTimer.cpp, independently of oracle/zl_oracle.c's zlo_sync_timer_*: the ring is a sparse
# dict here (a step that was never touched is "played" and empty, SyncTimer.cpp:78), integers are Python ints, the two
# `quint64 += double` accumulations (:665,671) go through float().
STEP_RING_COUNT = 32768           # SyncTimer.cpp:253


def schedule_into_step(step_commands: list, command: Command) -> bool:
    """SyncTimer::scheduleClipCommand once the step is known (:1014-1047).  True when the command was appended."""
    found = False
    for existing in step_commands:
        if existing.equivalent(command):
            if command.change_looping:
                existing.looping = command.looping; existing.change_looping = True
            if command.change_pitch:
                existing.pitch_change = command.pitch_change; existing.change_pitch = True
            if command.change_speed:
                existing.speed_ratio = command.speed_ratio; existing.change_speed = True
            if command.change_gain_db:
                existing.gain_db = command.gain_db; existing.change_gain_db = True
            if command.change_volume:
                existing.volume = command.volume; existing.change_volume = True
            if command.start:
                existing.start = True
            found = True
    if not found:
        step_commands.append(Command(**command.__dict__))
    return not found


class _Step:
    __slots__ = ("clips", "bpms", "played")

    def __init__(self):
        self.clips, self.bpms, self.played = [], [], True


class SyncTimerModel:
    def __init__(self):
        self.ring = {}                      # index -> _Step
        self.read_head = 0
        self.step_next_usecs = 0            # stepNextPlaybackPosition
        self.bpm = 120                      # SyncTimerThread::bpm
        self.paused = True                  # SyncTimerThread::paused == SyncTimerPrivate::isPaused (:750-752)
        self.playhead = 0                   # jackPlayhead
        self.playhead_bpm = 120.0           # jackPlayheadBpm (double)
        self.jack_next_usecs = 0            # jackNextPlaybackPosition
        self.subbeat_usecs = self.subbeat_ns(self.bpm, 1) // 1000          # :749
        self.latency_ms = 0
        self.cumulative_beat = 0
        self.beat = 0
        self.read_head_on_start = 0
        self._update_ahead()

    @staticmethod
    def subbeat_ns(bpm: int, count: int) -> int:                          # :180-183
        return (count * 60000000000) // (bpm * BEAT_SUBDIVISIONS)

    def _update_ahead(self):                                              # :704-707, :184-187
        ns = int(f32(f32(self.latency_ms) * f32(1000000)))
        self.ahead = int(f32(f32(ns // (60000000000 // (self.bpm * BEAT_SUBDIVISIONS))) + f32(1)))

    def set_latency(self, buffer_size: int, sample_rate: float):          # :730-741
        new = int((1000 * float(buffer_size)) / float(sample_rate))
        if new != self.latency_ms:
            self.latency_ms = new
            self._update_ahead()

    def _step(self, index: int) -> _Step:
        return self.ring.setdefault(index % STEP_RING_COUNT, _Step())

    def delayed_step(self, delay: int) -> _Step:                          # :364-378
        if self.paused:
            idx = self.read_head + delay + 1
        else:
            idx = self.read_head_on_start + max(self.cumulative_beat + delay, self.playhead + 1)
        st = self._step(idx)
        if st.played:                                                     # ensureFresh, :50-62
            st.played = False
            st.clips, st.bpms = [], []
        return st

    def schedule(self, command: Command, delay: int = 0):                 # :1011-1048
        schedule_into_step(self.delayed_step(delay).clips, command)

    def set_bpm(self, bpm: int):                                          # :954-975
        if self.bpm != bpm:
            self.bpm = bpm
            self.subbeat_usecs = self.subbeat_ns(bpm, 1) // 1000
            self._update_ahead()
            self.delayed_step(0).bpms.append(bpm)

    def start(self, bpm: int):                                            # :870-879
        self.set_bpm(bpm)
        self.read_head_on_start = self.read_head
        self.paused = False

    def stop(self):                                                       # :881-925
        self.paused = True
        self.beat = 0; self.cumulative_beat = 0; self.playhead = 0
        target = (self.read_head + 1) % STEP_RING_COUNT
        for off in range(STEP_RING_COUNT):      # ring order from the read head (a step created on the way is visited in its turn)
            idx = (self.read_head + off) % STEP_RING_COUNT
            st = self.ring.get(idx)
            if st is not None and not st.played:
                if idx != target:             # (a command re-scheduled into its own step folds into itself)
                    for c in list(st.clips):
                        c2 = Command(**c.__dict__)
                        c2.change_volume = True; c2.volume = f32(0.0)
                        self.schedule(c2, 0)
                st.played = True

    def timer_callback(self):                                             # :391-418
        while self.cumulative_beat < self.playhead + self.ahead * 2:
            self.beat = (self.beat + 1) % (BEAT_SUBDIVISIONS * 4)
            self.cumulative_beat += 1

    def queue_start(self, clip: int, channel: int):                       # :815-832
        c = Command(clip=clip, midi_channel=channel, midi_note=60, change_volume=True, volume=f32(1.0), looping=True, stop=True, start=True)
        bar = BEAT_SUBDIVISIONS * 4
        nzb = 0 if self.paused else bar - (self.cumulative_beat % bar)
        self.schedule(c, nzb + bar if self.cumulative_beat + nzb < self.playhead else nzb)

    def queue_stop(self, clip: int, channel: int):                        # :834-860
        for st in self.ring.values():
            if not st.played:
                for i, c in enumerate(st.clips):
                    if c.clip == clip:
                        del st.clips[i]
                        break
        self.delayed_step(0).clips.append(Command(clip=clip, midi_channel=channel, midi_note=60, stop=True))

    def process(self, nframes: int, current_usecs: int, next_usecs: int):  # :452-702 -> [(Command, tick)]
        out = []
        usecs_per_frame = (next_usecs - current_usecs) // nframes
        this_bpm = self.playhead_bpm
        this_len = float(self.subbeat_ns(int(self.playhead_bpm), 1)) / 1000.0                 # :484
        if not self.paused and self.playhead == 0:
            self.jack_next_usecs = current_usecs                                              # :490-497
        if self.step_next_usecs == 0:
            self.step_next_usecs = current_usecs                                              # :500-502
        first_free = 0
        while self.step_next_usecs < next_usecs and first_free < nframes:                     # :512
            st = self._step(self.read_head)
            self.read_head = (self.read_head + 1) % STEP_RING_COUNT
            if self.step_next_usecs <= current_usecs:                                         # :517-523
                first_free += 1
            else:
                rel = ((self.step_next_usecs - current_usecs) // usecs_per_frame) & 0xffffffff if usecs_per_frame else 0
                first_free = min(max(rel, first_free), nframes - 1)
            if not st.played:
                for c in st.clips:
                    out.append((Command(**c.__dict__), self.playhead))                        # :553-558
                i = 0
                while i < len(st.bpms):                                                       # SetBpmOperation, :606-612
                    nb = min(max(int(st.bpms[i]), 50), 200)
                    self.set_bpm(nb)
                    this_bpm = float(nb)
                    i += 1
                st.played = True
            if self.playhead_bpm != this_bpm:                                                 # :634-639
                self.playhead_bpm = this_bpm
                this_len = float(self.subbeat_ns(int(self.playhead_bpm), 1) // 1000)
            if not self.paused:                                                               # :660-667
                self.playhead += 1
                self.jack_next_usecs = int(float(self.jack_next_usecs) + this_len)
            self.step_next_usecs = int(float(self.step_next_usecs) + this_len)                # :671
        return out

    def clock(self, current_usecs: int, next_usecs: int) -> Clock:        # the getters, :990-1009
        if self.paused:
            return Clock(current_usecs, next_usecs, self.read_head, self.step_next_usecs, self.subbeat_usecs)
        return Clock(current_usecs, next_usecs, self.playhead, self.jack_next_usecs, self.subbeat_usecs)


# ---------------------------------------------------------------- AudioLevels (AudioLevels.cpp:330-412)
# ---------------------------------------------------------------- ClipAudioSourcePositionsModel + the per-clip level / progress chain
# Written from /root/reference/lib/ClipAudioSourcePositionsModel.cpp and ClipAudioSource.cpp:88-113,225-240, independently of
# oracle/zl_oracle.c's zlo_positions_* / zlo_sync_*: rows are dicts, the wall clock is passed in (QDateTime::currentMSecsSinceEpoch()).
POSITION_COUNT = 32               # ClipAudioSourcePositionsModel.cpp:5


class PositionsModel:
    def __init__(self):                                                   # :7-21
        self.rows = [dict(id=-1, progress=f32(0), gain=f32(0), updated=0) for _ in range(POSITION_COUNT)]
        self.update_peak = False
        self.peak = f32(0)

    def clean_up(self, now):                                              # :191-209: rows not updated for a second are orphans
        for r in self.rows:
            if r["id"] > -1 and r["updated"] < now - 1000:
                r["id"] = -1; r["gain"] = f32(0); r["progress"] = f32(0)

    def create(self, initial_progress, now):                              # :78-100
        row = -1
        found = None
        for r in self.rows:
            row += 1
            if r["id"] == -1:
                found = r
                break
        if found is not None:
            found["id"] = row; found["progress"] = f32(initial_progress); found["updated"] = now
            self.update_peak = True
            self.clean_up(now)
        return row                                                        # (all rows taken: the last row's number, though it is another voice's)

    def set_gain_and_progress(self, pid, gain, progress, now):            # :126-138
        if -1 < pid < POSITION_COUNT:
            r = self.rows[pid]
            r["gain"] = f32(gain); r["progress"] = f32(progress); r["updated"] = now
            self.update_peak = True

    def remove(self, pid, now):                                           # :140-153
        if -1 < pid < POSITION_COUNT:
            r = self.rows[pid]
            r["id"] = -1; r["gain"] = f32(0); r["progress"] = f32(0)
            self.update_peak = True
        self.clean_up(now)

    def peak_gain(self):                                                  # :160-173
        if self.update_peak:
            peak = f32(0)
            for r in self.rows:
                peak = max(peak, r["gain"])
            if abs(f64(f32(self.peak - peak))) > 0.01:                    # abs(float) > 0.01: a double comparison
                self.peak = peak
            self.update_peak = False
        return self.peak

    def first_progress(self):                                             # :175-185
        for r in self.rows:
            if r["id"] > -1:
                return f64(r["progress"])
        return f64(-1.0)


def _log10f(x):
    """std::log10(float): the C library's log10f -- its last bit is the platform's (glibc here, as for the C oracle and the product's host
    code; numpy's float32 log10 is another implementation and differs in the last place now and then)"""
    import ctypes
    global _libm
    if _libm is None:
        _libm = ctypes.CDLL("libm.so.6")
        _libm.log10f.restype = ctypes.c_float; _libm.log10f.argtypes = [ctypes.c_float]
    return f32(_libm.log10f(float(x)))


_libm = None


def _gain_to_db(gain, T):                                                 # juce::Decibels::gainToDecibels<T>, floor -100
    if not gain > 0:
        return T(-100.0)
    d = T(T(math.log10(gain) if T is f64 else _log10f(gain)) * T(20.0))
    return d if d > T(-100.0) else T(-100.0)


class ClipMeter:
    """ClipAudioSource::syncAudioLevel / syncProgress (ClipAudioSource.cpp:88-113,225-240) with the state of :68-69,84-87."""

    def __init__(self):
        self.current_db = f64(-400.0); self.prev_db = f64(-400.0)
        self.first_progress = f64(0.0)
        self.next_position_update = 0; self.next_gain_update = 0

    def sync_audio_level(self, model: PositionsModel, now):               # -> the callback's float argument, or None
        fired = None
        if self.next_gain_update < now:                                   # :89
            self.prev_db = self.current_db                                # :90
            self.current_db = f64(_gain_to_db(model.peak_gain(), f32))    # :92 (the tracktion level client reads its -100 dB floor)
            prev_level = f64(math.pow(10.0, float(self.prev_db) * 0.05)) if self.prev_db > -100.0 else f64(0.0)   # :98
            if self.prev_db > self.current_db:                            # :100-101
                self.current_db = _gain_to_db(f64(prev_level * f64(0.94)), f64)
            if abs(self.current_db - self.prev_db) > 0.1:                 # :104
                fired = f32(self.current_db)                              # :108
            self.next_gain_update = now + 30                              # :111
        return fired

    def sync_progress(self, model: PositionsModel, start_sec, duration, has_callback, now):
        fired = None
        if self.next_position_update < now:                               # :226
            new_position = f64(f32(f32(start_sec) / f32(duration)))       # :227 (float / float)
            if has_callback and model.first_progress() > f64(f32(-1.0)):  # :228
                new_position = model.first_progress()
            if abs(self.first_progress - new_position) > 0.001:           # :231
                self.first_progress = new_position
                if has_callback:
                    fired = f32(self.first_progress * f64(f32(duration)))  # :234: double * float -> the callback's float
                self.next_position_update = now + 100                     # :238
        return fired


def sample_to_peak_int(x) -> int:
    v = abs(float(f32(f32(131072.0) * f32(x))))
    return int(v)


def levels_tick(peak_a, peak_b, L, R):
    peak_a = max(0, peak_a - 10000)
    peak_b = max(0, peak_b - 10000)
    for x in L:
        peak_a = max(peak_a, sample_to_peak_int(x))
    for x in R:
        peak_b = max(peak_b, sample_to_peak_int(x))
    return peak_a, peak_b


def to_dbfs(raw):
    raw = f32(raw)
    if raw <= 0:
        return f32(-200)
    v = f32(f32(20) * f32(np.log10(raw, dtype=np.float32)))
    return f32(-200) if v < f32(-200) else v


# ---------------------------------------------------------------- JackPassthrough (JackPassthrough.cpp:45-115)
def passthrough(inL, inR, dry, fx1, fx2, pan, muted):
    n = len(inL)
    out = [np.zeros(n, dtype=np.float32) for _ in range(6)]
    if muted:
        return out
    pan = f32(pan)
    lm = min(f32(f32(1) - pan), f32(1.0))
    rm = min(f32(f32(1) + pan), f32(1.0))
    for k, amt in enumerate((f32(dry), f32(fx1), f32(fx2))):
        if pan == 0 and amt == 0:
            continue
        if pan == 0 and amt == 1:
            out[2 * k][:] = inL
            out[2 * k + 1][:] = inR
            continue
        for f in range(n):
            out[2 * k][f] = f32(f32(amt * f32(inL[f])) * lm)
            out[2 * k + 1][f] = f32(f32(amt * f32(inR[f])) * rm)
    return out


def pcm16(x):
    """16-bit samples of the recorder (AudioLevels.cpp:53-58: juce::WavAudioFormat 16 bit; restated from public JUCE source, version
    unpinned): clamp, INT_MAX * x in double rounded half to even, upper 16 bits.  NaN -> 0."""
    d = np.asarray(x, dtype=np.float32).astype(np.float64)
    q = np.rint(2147483647.0 * np.where(np.isnan(d), 0.0, np.clip(d, -1.0, 1.0)))
    q = np.where(d <= -1.0, -2147483648.0, np.where(d >= 1.0, 2147483647.0, q)).astype(np.int64)
    return (q >> 16).astype(np.int16)
