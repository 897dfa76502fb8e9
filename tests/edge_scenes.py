"""Edge-case scenes shared by the CPU (host harness) and GPU tiers: each builds a small Scene around one unusual
parameter combination; the oracle defines the expected behaviour, the engine must reproduce it bit for bit."""
import ctypes as C

import numpy as np

from scenario import Scene, play_cmd, rand_source, stop_cmd


def _base(seed, nframes=128, nblocks=14, nsounds=4, length=6000, sr=48000.0, mono=()):
    rng = np.random.default_rng(seed)
    sc = Scene(num_buses=2, voices_per_bus=4, fs=48000.0, nframes=nframes, nblocks=nblocks)
    for i in range(nsounds):
        L, R = rand_source(rng, length + 37 * i, stereo=(i not in mono))
        sc.sounds.append((L, R, sr))
    return sc


def _play_all(sc, notes, loop=True, **extra):
    sc.events[0] = [("cmd", play_cmd(i, midi_channel=(i % 2) - 2, loop=loop, note=notes[i % len(notes)], volume=0.5 + 0.1 * i, **extra), 0)
                    for i in range(len(sc.sounds))]


def extreme_ratios():
    sc = _base(1)
    _play_all(sc, [12, 108, 36, 96])               # pitch ratios 2^-4 .. 2^4
    for i in range(4):
        sc.clip_setup[i] = lambda lib, clip: lib.zlo_clip_set_length(clip, C.c_float(0.11), 120)
    return sc


def tiny_loops():
    sc = _base(2, nblocks=10)
    _play_all(sc, [60, 67, 55, 72])
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = 0.013                               # fractional: sample-space wrap
            clip.lengthInSeconds = float(np.float32((3 + 2 * i) / 48000.0))   # loops of 3..9 frames
        sc.clip_setup[i] = setup
    return sc


def stop_beyond_the_file():
    sc = _base(3, length=1500)
    _play_all(sc, [60, 64, 57, 70], loop=True)
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = 0.3
            clip.lengthInSeconds = float(np.float32(0.05 + 0.01 * i))         # 2400+ frames > the 1500-frame file (Q5 zeros)
        sc.clip_setup[i] = setup
    return sc


def negative_beats_q10():
    sc = _base(4)
    _play_all(sc, [60, 62, 58, 65])                # lengthInBeats = -1 (the default): beat-locked with a saturated tick count
    return sc


def start_near_the_end_and_slices():
    sc = _base(5)
    sc.events[0] = [("cmd", play_cmd(i, midi_channel=(i % 2) - 2, loop=True, note=60 + i, volume=0.6, changeSlice=1, slice=[15, 0, 7, 14][i]), 0)
                    for i in range(4)]
    for i in range(4):
        def setup(lib, clip, i=i):
            lib.zlo_clip_set_start_position(clip, C.c_float(0.1))
            lib.zlo_clip_set_length(clip, C.c_float(0.05), 120)
        sc.clip_setup[i] = setup
    return sc


def envelopes_at_their_limits():
    sc = _base(6, nblocks=18)
    _play_all(sc, [60, 60, 65, 55], loop=False)
    shapes = [(0.0, 0.0, 1.0, 0.0), (0.01, 0.0, 0.0, 0.02), (0.0, 0.003, 0.2, 0.0), (0.02, 0.02, 1.0, 0.03)]
    for i in range(4):
        def setup(lib, clip, i=i):
            a, d, s, r = shapes[i]
            clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = a, d, s, r
            lib.zlo_clip_set_length(clip, C.c_float(0.08), 120)
        sc.clip_setup[i] = setup
    sc.events[6] = [("cmd", stop_cmd(i, midi_channel=(i % 2) - 2, note=[60, 60, 65, 55][i]), 0) for i in range(4)]
    return sc


def mono_and_stereo_neighbours():
    sc = _base(7, nsounds=8, mono=(1, 2, 5))
    sc.voices_per_bus = 8; sc.num_buses = 1
    sc.events[0] = [("cmd", play_cmd(i, midi_channel=-2, loop=True, note=57 + i, volume=0.4 + 0.05 * i), 0) for i in range(8)]
    for i in range(8):
        sc.clip_setup[i] = lambda lib, clip: lib.zlo_clip_set_length(clip, C.c_float(0.07), 120)
    return sc


def resampled_sources():
    sc = _base(8, sr=22050.0)
    sc.sounds[1] = (sc.sounds[1][0], sc.sounds[1][1], 96000.0)
    sc.sounds[2] = (sc.sounds[2][0], sc.sounds[2][1], 44100.0)
    _play_all(sc, [60, 60, 61, 48])
    for i in range(4):
        sc.clip_setup[i] = lambda lib, clip: lib.zlo_clip_set_length(clip, C.c_float(0.09), 120)
    return sc


def beat_locked_with_irregular_clocks():
    """Beat-locked loops (integer lengthInBeats: restart against the JACK clock, Q9a) under a clock that jitters, changes
    its period and once steps backwards: the planner's walk over the blocks (no bisection) must find the same frames."""
    from libzl_amd._abi import Clock
    sc = _base(9, nblocks=300, length=9000)                            # 0.8 s: the 1-beat loops (345 ms at 174 bpm) restart twice, the 2-beat loops once
    _play_all(sc, [60, 63, 57, 66])
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = float(1 + (i % 2))                   # integer beats -> clock-driven restart
            clip.lengthInSeconds = float(np.float32(0.05 + 0.02 * i))
        sc.clip_setup[i] = setup
    sc.bpm = 174

    def clocks(start, n):
        arr = (Clock * n)()
        sub = ((60000000000) // (sc.bpm * 96)) // 1000
        for j in range(n):
            k = start + j
            period = 2667 + (37 if k % 3 == 0 else -21 if k % 5 == 0 else 0)          # changing period
            cur = k * 2667 + ((k * 7919) % 13) - (900 if k in (17, 128, 259) else 0)  # jitter, backward steps (one right where a restart falls)
            arr[j].current_usecs = cur
            arr[j].next_usecs = cur + period
            arr[j].jack_playhead = 0
            arr[j].jack_playhead_usecs = 0
            arr[j].jack_subbeat_length_usecs = sub
        return arr
    sc.clocks = clocks
    return sc


def beat_locked_long_regular():
    """Beat-locked loops over a long batch with a regular clock: restarts found by bisection over the window's blocks."""
    sc = _base(10, nblocks=700, length=30000)
    _play_all(sc, [60, 64, 55, 67])
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = float(1 + i)
            clip.lengthInSeconds = float(np.float32(0.3 + 0.05 * i))
        sc.clip_setup[i] = setup
    sc.bpm = 140
    return sc


def beat_locked_moving_playhead():
    """Beat-locked loops against a MOVING SyncTimer playhead (a timer that has been running for 9000 cycles: playhead ~ 7700):
    commands carry the tick they are dispatched with, a retrigger lands while the loop plays, one command carries a tick 150
    ticks in the past -- nextLoopTick behind the playhead: the u64 wrap of SamplerSynthVoice.cpp:180-181,236-237 and restarts in
    consecutive frames -- and one a tick in the future (a start that SamplerSynth's ring delayed)."""
    sc = _base(11, nblocks=260, length=9000)
    sc.bpm = 200; sc.moving_playhead = True; sc.block0 = 9000
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = float(1 + (i % 2))
            clip.lengthInSeconds = float(np.float32(0.04 + 0.01 * i))
        sc.clip_setup[i] = setup
    t = sc.tick_at
    sc.events[0] = [("cmd", play_cmd(0, midi_channel=-2, note=60, volume=0.8), t(0)), ("cmd", play_cmd(1, midi_channel=-1, note=64, volume=0.7), t(0))]
    sc.events[9] = [("cmd", play_cmd(2, midi_channel=-2, note=57, volume=0.6), t(9) - 150)]
    sc.events[31] = [("cmd", play_cmd(3, midi_channel=-1, note=67, volume=0.9), t(31) + 40)]
    sc.events[140] = [("cmd", play_cmd(0, midi_channel=-2, note=60, volume=0.5), t(140))]            # restart over the playing loop
    sc.events[200] = [("cmd", stop_cmd(1, midi_channel=-1, note=64), t(200))]
    return sc


def beat_locked_moving_playhead_long():
    """The same against windows: 2600 blocks of 64 frames in one batch (several plan windows with plan_window_blocks small),
    restarts found by bisection over blocks whose clocks carry a different playhead each."""
    sc = _base(12, nframes=64, nblocks=2600, length=30000)
    sc.bpm = 140; sc.moving_playhead = True; sc.block0 = 5000
    _play_all(sc, [60, 64, 55, 67])
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = float(1 + i)
            clip.lengthInSeconds = float(np.float32(0.3 + 0.05 * i))
        sc.clip_setup[i] = setup
    sc.events[0] = [("cmd", ev[1], sc.tick_at(0) + j) for j, ev in enumerate(sc.events[0])]
    return sc


def random_envelopes(seed):
    """A mixed random scene whose clips get random ADSR times (zero, tiny, short, longer than the scene) and sustain
    levels (0, 1, in between), with the scene's own commands (note-offs, restarts, volume changes) on top."""
    from scenario import random_scene
    rng = np.random.default_rng(seed)
    sc = random_scene(seed, nframes=int(rng.choice([64, 128, 256])), nblocks=int(rng.integers(24, 90)), nclips=int(rng.integers(5, 12)),
                      min_len=2000, max_len=30000, events=True)
    for i in list(sc.clip_setup):
        base = sc.clip_setup[i]
        a = float(rng.choice([0.0, 1e-4, rng.uniform(0.001, 0.2), rng.uniform(0.2, 1.5)]))
        d = float(rng.choice([0.0, 1e-4, rng.uniform(0.001, 0.2), rng.uniform(0.2, 1.5)]))
        sus = float(rng.choice([0.0, 1.0, rng.uniform(0.01, 0.99), 1e-3]))
        r = float(rng.choice([0.0, 1e-4, rng.uniform(0.001, 0.1), rng.uniform(0.1, 0.8)]))

        def setup(lib, clip, base=base, a=a, d=d, sus=sus, r=r):
            base(lib, clip)
            clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = a, d, sus, r
        sc.clip_setup[i] = setup
    return sc


def unit_step_loops_many_passes():
    """Playback at the source rate from an integer start (a pass is ONE exact linear run), 1500 blocks of 64 frames in
    one window: dozens of passes per voice -- more than the inline run list holds, so the periodic descriptor takes over."""
    rng = np.random.default_rng(77)
    sc = Scene(num_buses=2, voices_per_bus=4, fs=48000.0, nframes=64, nblocks=1500)
    for i in range(6):
        n = int(rng.integers(700, 1500))
        L, R = rand_source(rng, n, stereo=bool(i % 2))
        sc.sounds.append((L, R, 48000.0))

        def setup(lib, clip, n=n, i=i):
            lib.zlo_clip_set_length(clip, C.c_float(0.0217 + 0.0031 * i), 120)      # fractional beats: sample-space loop
            if clip.lengthInSeconds > n / 48000.0 * 0.9:
                clip.lengthInSeconds = float(np.float32(n / 48000.0 * 0.6))
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(0.7))
        sc.clip_setup[i] = setup
    sc.events[0] = [("cmd", play_cmd(i, midi_channel=(i % 2) - 2, loop=True, note=60, volume=0.8), 0) for i in range(6)]
    return sc


def positions_beyond_2_to_24():
    """A 17.2 M-frame source (6 minutes at 48 kHz) played from just below frame 2^24: the position crosses a binade of
    the fp64 recurrence inside the first blocks, alpha = (float)(P - pos) has to come from a P whose spacing is 2^-28, and
    the unit-step / interior variants index with 25-bit positions.  Four voices: playback rate, two pitched, a loop that
    restarts back below 2^24."""
    rng = np.random.default_rng(24)
    n = 17_200_000
    L = rng.uniform(-1, 1, n).astype(np.float32)
    sc = Scene(num_buses=2, voices_per_bus=4, fs=48000.0, nframes=128, nblocks=12)
    sc.sounds.append((L, None, 48000.0))                          # one mono source shared by the four clips' sounds
    for i in range(1, 4):
        sc.sounds.append((L, None, 48000.0))
    starts = [349.5250, 349.5249, 349.5251, 349.5240]             # seconds: 16 777 200 +- a few frames
    for i in range(4):
        def setup(lib, clip, i=i):
            lib.zlo_clip_set_start_position(clip, C.c_float(starts[i]))
            clip.lengthInBeats = 0.37                              # fractional: sample-space loop
            clip.lengthInSeconds = float(np.float32([5.0, 5.0, 5.0, 0.011][i]))   # the last one loops every 528 frames
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(0.8))
        sc.clip_setup[i] = setup
    sc.events[0] = [("cmd", play_cmd(i, midi_channel=(i % 2) - 2, loop=True, note=[60, 67, 53, 60][i], volume=0.7), 0) for i in range(4)]
    return sc


def loop_edits_while_playing():
    """Steady pitched loops (their recorded pass is being replayed) whose clip is edited between blocks: a longer loop, a
    shorter one, a moved start, a different sustain level, a loop turned beat-locked and back.  Every edit must drop the
    recorded pass (its key no longer matches, or the voice is no longer on it) and the planner must take over."""
    rng = np.random.default_rng(91)
    sc = Scene(num_buses=2, voices_per_bus=4, fs=48000.0, nframes=64, nblocks=420)
    for i in range(4):
        L, R = rand_source(rng, 9000 + 500 * i, stereo=bool(i % 2))
        sc.sounds.append((L, R, 44100.0))

        def setup(lib, clip, i=i):
            clip.lengthInBeats = 0.37 + 0.01 * i
            clip.lengthInSeconds = float(np.float32(0.021 + 0.004 * i))       # loops of ~1000 source frames
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(0.6))
        sc.clip_setup[i] = setup
    sc.events[0] = [("cmd", play_cmd(i, midi_channel=(i % 2) - 2, loop=True, note=[57, 64, 60, 67][i], volume=0.7), 0) for i in range(4)]

    def set_len(sec):
        return lambda lib, clip: setattr(clip, "lengthInSeconds", float(np.float32(sec)))

    def set_start(sec):
        return lambda lib, clip: lib.zlo_clip_set_start_position(clip, C.c_float(sec))

    def set_beats(b):
        return lambda lib, clip: setattr(clip, "lengthInBeats", b)

    def set_sustain(x):
        def f(lib, clip):
            clip.adsr.p.sustain = x
        return f
    sc.events[90] = [("clip", 0, set_len(0.035))]                 # longer loop
    sc.events[150] = [("clip", 1, set_len(0.012)), ("clip", 2, set_start(0.004))]
    sc.events[210] = [("clip", 3, set_beats(1.0))]                # integer beats: clock-driven restarts from now on
    sc.events[260] = [("clip", 3, set_beats(0.41)), ("clip", 0, set_sustain(0.8))]   # (sustain of a playing voice stays: set at noteOn)
    sc.events[330] = [("cmd", dict(clip=1, midiChannel=-1, midiNote=64, changeVolume=1, volume=0.3), 0), ("clip", 2, set_start(0.0))]
    return sc


def retrigger_onto_the_cached_pass():
    """Steady loops at the playback rate whose recorded pass the planner replays window after window; the voices are stopped without a
    tail and the SAME notes started again on the same slots, one slot then taken by another clip: the slot's cached pass belongs to a
    voice that is gone -- the new voice starts off the pass (its phase is unknown) and must be planned afresh, with a fresh clock state
    (nextLoopUsecs is formed at the top of ITS first block, SamplerSynthVoice.cpp:179-182) -- and later the clips get an integer
    lengthInBeats, so that value decides when the loops restart.  Moving playhead."""
    rng = np.random.default_rng(78)
    sc = Scene(num_buses=2, voices_per_bus=4, fs=48000.0, nframes=64, nblocks=420, bpm=200, moving_playhead=True, block0=9000)
    for i in range(4):
        n = int(rng.integers(800, 1400))
        L, R = rand_source(rng, n, stereo=bool(i % 2))
        sc.sounds.append((L, R, 48000.0))

        def setup(lib, clip, n=n, i=i):
            lib.zlo_clip_set_length(clip, C.c_float(0.0217 + 0.0031 * i), 120)      # fractional beats: sample-space loop
            clip.lengthInSeconds = float(np.float32(n / 48000.0 * 0.5))
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(0.7))
        sc.clip_setup[i] = setup
    slots = [(0, 0), (0, 2), (1, 1), (1, 3)]
    starts = lambda k: [("start", b, s, play_cmd(i, midi_channel=b - 2, loop=True, note=60, volume=0.8), sc.tick_at(k)) for i, (b, s) in enumerate(slots)]
    sc.events[0] = starts(0)
    sc.events[150] = [("stopv", b, s, False) for (b, s) in slots] + starts(150)       # hard stop, the same voices again on the same slots
    sc.events[151] = [("stopv", 0, 2, False), ("start", 0, 2, play_cmd(1, midi_channel=-2, loop=True, note=60, volume=0.5), sc.tick_at(151))]

    def set_beats(b):
        return lambda lib, clip: setattr(clip, "lengthInBeats", b)
    sc.events[230] = [("clip", 0, set_beats(1.0)), ("clip", 2, set_beats(2.0))]        # clock-driven restarts from here on (144 ms per beat at 200 bpm)
    return sc


def channels_disabled_and_enabled():
    """SamplerSynth::setChannelEnabled (SamplerSynth.cpp:343-351, SamplerChannel::process :116-123): a disabled channel still drains its
    commands -- a voice started on it is set up, a stop releases its envelope, a patch lands -- but none of its voices is processed: they
    keep position, envelope and clock state, report nothing, and go on from exactly there when the channel is enabled again (a beat-locked
    loop then finds its next restart time long past: it restarts in the first frame)."""
    sc = _base(12, nblocks=60, nsounds=4, length=5200)
    sc.num_buses = 3
    for i in range(4):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = [0.37, 1.0, 0.29, 0.41][i]         # clip 1 is beat-locked
            clip.lengthInSeconds = float(np.float32(0.04 + 0.01 * i))
            lib.zlo_clip_set_adsr_release(clip, C.c_float(0.006))
            if i == 2:
                clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain = (0.01, 0.01, 0.6)
        sc.clip_setup[i] = setup
    sc.bpm = 200
    sc.events[0] = [("cmd", play_cmd(0, midi_channel=-2, loop=True, note=60, volume=0.8), 0), ("cmd", play_cmd(1, midi_channel=-1, loop=True, note=62, volume=0.7), 0),
                    ("cmd", play_cmd(2, midi_channel=-1, loop=False, note=57, volume=0.9), 0), ("cmd", play_cmd(3, midi_channel=0, loop=True, note=64, volume=0.6), 0)]
    sc.events[5] = [("enable", 1, False)]                                            # channel -1 stands still: its loop and its one-shot (in its attack)
    sc.events[8] = [("cmd", play_cmd(3, midi_channel=-1, loop=True, note=67, volume=0.5), 0),      # started on the disabled channel: waits
                    ("cmd", dict(clip=1, midiChannel=-1, midiNote=62, changeVolume=1, volume=0.2), 0)]   # a patch lands
    sc.events[11] = [("enable", 0, False), ("enable", 0, True)]                      # off and on between two cycles: nothing happened
    sc.events[14] = [("cmd", stop_cmd(2, midi_channel=-1, note=57), 0)]              # the stop releases the envelope of a voice that stands still
    sc.events[30] = [("enable", 1, True)]                                            # 25 cycles later everything goes on
    sc.events[40] = [("enable", 2, False), ("enable", 1, False)]
    sc.events[44] = [("enable", 2, True)]
    sc.events[52] = [("enable", 1, True), ("enable", 1, True)]
    return sc


def voice_level_calls():
    """The JUCE SynthesiserVoice surface behind include/zlhip_voice_adapter.h: setCurrentCommand on a PLAYING voice with every patch
    (SamplerSynthVoice.cpp:58-100) -- among them startPlayback = "restart playback": the position goes back to the start of the voice's
    slice (:86-91), and changeSlice + startPlayback = jump to another slice's start --, stopNote with a tail and stopNote(.., false)
    in the middle of a loop (:146-169), a patch and a stop addressed to voices that do not play."""
    sc = _base(11, nblocks=26, nsounds=3, length=5200)
    for i in range(3):
        def setup(lib, clip, i=i):
            clip.lengthInBeats = 0.37
            clip.lengthInSeconds = float(np.float32(0.05 + 0.01 * i))
            lib.zlo_clip_set_slices(clip, 4)                        # [0, 1/16, 2/16, 3/16] (the constructor's 16, shrunk)
            lib.zlo_clip_set_adsr_release(clip, C.c_float(0.004))
        sc.clip_setup[i] = setup
    sc.events[0] = [("start", 0, 0, play_cmd(0, midi_channel=-2, loop=True, note=60, volume=0.8), 0),
                    ("start", 0, 2, play_cmd(1, midi_channel=-2, loop=True, note=64, volume=0.7, changeSlice=1, slice=1), 0),
                    ("start", 1, 1, play_cmd(2, midi_channel=-1, loop=False, note=57, volume=0.9), 0),
                    ("start", 1, 3, play_cmd(0, midi_channel=-1, loop=True, note=67, volume=0.5), 0)]
    sc.events[4] = [("update", 0, 0, dict(clip=0, midiChannel=-2, midiNote=60, startPlayback=1)),                       # restart from the top
                    ("update", 0, 1, dict(clip=0, midiChannel=-2, midiNote=60, changeVolume=1, volume=0.1))]            # slot 1 does not play
    sc.events[7] = [("update", 0, 2, dict(clip=1, midiChannel=-2, midiNote=64, changeSlice=1, slice=3, startPlayback=1, changeVolume=1, volume=0.4)),
                    ("update", 1, 1, dict(clip=2, midiChannel=-1, midiNote=57, changeLooping=1, looping=1, changePitch=1, pitchChange=0.5,
                                          changeSpeed=1, speedRatio=2.0, changeGainDb=1, gainDb=-3.0))]
    sc.events[11] = [("stopv", 1, 3, False), ("stopv", 1, 0, True)]                                                      # hard stop mid-loop; slot 0 does not play
    sc.events[14] = [("stopv", 0, 0, True), ("start", 1, 3, play_cmd(1, midi_channel=-1, loop=True, note=55, volume=0.6), 0)]   # the freed slot is taken again
    sc.events[18] = [("update", 1, 3, dict(clip=1, midiChannel=-1, midiNote=55, startPlayback=1, changeLooping=1, looping=0)),
                     ("stopv", 0, 2, False)]
    return sc


SCENES = {f.__name__: f for f in (voice_level_calls, channels_disabled_and_enabled, retrigger_onto_the_cached_pass, beat_locked_moving_playhead, beat_locked_moving_playhead_long, unit_step_loops_many_passes, positions_beyond_2_to_24, loop_edits_while_playing, beat_locked_with_irregular_clocks, beat_locked_long_regular, extreme_ratios, tiny_loops, stop_beyond_the_file, negative_beats_q10, start_near_the_end_and_slices,
                                  envelopes_at_their_limits, mono_and_stereo_neighbours, resampled_sources)}
for _seed in range(8):
    SCENES[f"random_envelopes_{_seed}"] = (lambda _seed=_seed: random_envelopes(9100 + _seed))
