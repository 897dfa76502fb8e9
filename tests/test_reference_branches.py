"""Branches of the reference path that scripts/oracle_coverage.sh (gcov over oracle/zl_oracle.c under both test tiers) showed no test
had taken: setSlices' shrink / clear / grow-from-a-shrunk-table (ClipAudioSource.cpp:495-528), the setter clamps
(:255-259,328-336; SyncTimer.cpp:180-183), more voices of one clip than its positions model has rows
(ClipAudioSourcePositionsModel.cpp:78-100: row 31 is handed out again), and the level chain's decay down to the -100 dB floor
(ClipAudioSource.cpp:88-113, juce::Decibels).  Patches of playing voices (changeLooping both ways, the stored-only fields, equivalence
by note and by slice) are the golden g10_command_patches (tests/golden/make_golden.py --patches); JACK time jumps are in
tests/test_scheduler.py.  CPU tier: oracle = numpy restatement = the product's host code; GPU tier: the product through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from oracle import np_restatement as npr
from oracle import zl_oracle as zo

f32 = np.float32


@pytest.fixture(scope="module")
def zl(built):
    from libzl_amd import libzl
    return libzl.load()


def _product_params(zl, c):
    from libzl_amd import _abi
    p = _abi.ClipParams()
    assert zl.libzl_hotpath_clip_params(c, C.byref(p)) == 0
    return p


@pytest.mark.parametrize("seq", [(4, 6, 0, 3), (16, 1, 2, 128, 5, 0, 0, 7), (0, 16, 15, 17, 17, 3, 40), (2, 1, 0, 1, 2, 3, 2, 9)])
def test_set_slices_shrinks_clears_and_grows_like_the_reference(zl, seq):
    """ClipAudioSource::setSlices after the constructor's setSlices(16): fewer slices drop the last entries, 0 clears the table, more
    slices fit only the ADDED entries between the last position and 1.0 (so 16 -> 4 -> 6 is [0, 1/16, 2/16, 3/16, 0.59375, 1.0]).
    The table as doubles, bit for bit, on the C oracle, the numpy restatement and the product's clip (what its engine receives)."""
    lib = zo.load()
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(2.0), 48000.0)
    nc = npr.Clip(); nc.set_slices(16)
    L = np.zeros(96000, dtype=np.float32)
    c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None, len(L), 48000.0, b"slices")
    try:
        for n in seq:
            lib.zlo_clip_set_slices(C.byref(oc), n); nc.set_slices(n); zl.ClipAudioSource_setSlices(c, n)
            want = [oc.slicePositions[i] for i in range(oc.nSlicePositions)]
            assert oc.nSlicePositions == n and want == nc.slice_pos
            p = _product_params(zl, c)
            assert p.num_slice_positions == n and [p.slice_positions[i] for i in range(n)] == want
            # start / stop of every slice index around the table's end (ClipAudioSource.cpp:261-277)
            nc.start_sec, nc.length_sec = f32(oc.startPositionInSeconds), f32(oc.lengthInSeconds)
            for sl in (-1, 0, n - 2, n - 1, n, n + 5):
                assert lib.zlo_clip_get_start_position(C.byref(oc), sl) == nc.get_start(sl)
                assert lib.zlo_clip_get_stop_position(C.byref(oc), sl) == nc.get_stop(sl)
    finally:
        zl.ClipAudioSource_destroy(c)


def test_set_slices_known_answer():
    """hand-derived from ClipAudioSource.cpp:495-528: 16 slices (the constructor) -> 4 keeps [0, 1/16, 2/16, 3/16]; -> 6 adds two entries
    spaced (1 - 3/16) / 2 = 0.40625 apart: 0.59375 and 1.0 (a slice that starts at the end); -> 0 clears; -> 3 from the empty table is
    [0, 1/3, 2/3] with the increments accumulated in double"""
    lib = zo.load()
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(2.0), 48000.0)
    tbl = lambda: [oc.slicePositions[i] for i in range(oc.nSlicePositions)]
    lib.zlo_clip_set_slices(C.byref(oc), 4); assert tbl() == [0.0, 0.0625, 0.125, 0.1875]
    lib.zlo_clip_set_slices(C.byref(oc), 6); assert tbl() == [0.0, 0.0625, 0.125, 0.1875, 0.59375, 1.0]
    lib.zlo_clip_set_slices(C.byref(oc), 0); assert tbl() == []
    lib.zlo_clip_set_slices(C.byref(oc), 3); assert tbl() == [0.0, 1.0 / 3.0, 1.0 / 3.0 + 1.0 / 3.0]


def test_setter_clamps(zl):
    """setVolumeAbsolute clamps to [0, 1], setStartPosition to >= 0, setLength's bpm to [50, 200] (qBound in
    SyncTimer::subbeatCountToSeconds): oracle, numpy restatement and the product's setters."""
    lib = zo.load()
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(2.0), 48000.0)
    nc = npr.Clip()
    L = np.zeros(96000, dtype=np.float32)
    c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None, len(L), 48000.0, b"clamps")
    try:
        for vol in (-0.5, 0.0, 0.3, 1.0, 7.0, -1e30):
            lib.zlo_clip_set_volume_absolute(C.byref(oc), C.c_float(vol)); nc.set_volume_absolute(vol); zl.ClipAudioSource_setVolumeAbsolute(c, vol)
            assert oc.volumeAbsolute == nc.volume_abs == _product_params(zl, c).volume_absolute == zl.ClipAudioSource_volumeAbsolute(c)
        assert oc.volumeAbsolute == 0.0
        for sec in (-1.0, 0.0, 0.25, -0.0):
            lib.zlo_clip_set_start_position(C.byref(oc), C.c_float(sec)); nc.set_start_position(sec); zl.ClipAudioSource_setStartPosition(c, sec)
            assert oc.startPositionInSeconds == nc.start_sec == _product_params(zl, c).start_position_seconds
        for beat, bpm in ((4.0, 250), (4.0, 200), (4.0, 30), (4.0, 50), (0.37, 1000), (2.5, 1), (1.0, 120)):
            lib.zlo_clip_set_length(C.byref(oc), C.c_float(beat), bpm); nc.set_length(beat, bpm); zl.ClipAudioSource_setLength(c, beat, bpm)
            p = _product_params(zl, c)
            assert oc.lengthInSeconds == nc.length_sec == p.length_seconds and oc.lengthInBeats == nc.length_beats == p.length_in_beats
        lib.zlo_clip_set_length(C.byref(oc), C.c_float(4.0), 250); a = oc.lengthInSeconds
        lib.zlo_clip_set_length(C.byref(oc), C.c_float(4.0), 200); assert a == oc.lengthInSeconds == f32(1.2)       # 4 beats at the 200 bpm cap
    finally:
        zl.ClipAudioSource_destroy(c)


def test_level_chain_decays_to_the_floor_and_falls_silent_on_the_oracle():
    """ClipAudioSource.cpp:88-113 after the sound has ended: the level falls by 20 log10(0.94) = -0.537 dB per 30 ms tick and every tick
    notifies, until gainToDecibels clamps at -100 dB (juce::Decibels: jmax(-100, ...)); the tick that lands on the floor is the last
    callback -- from then on |current - previous| = 0."""
    lib = zo.load()
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(1.0), 48000.0)
    m = zo.ClipMeter(); lib.zlo_clip_meter_init(C.byref(m))
    val = C.c_float()
    pid = lib.zlo_positions_create(C.byref(oc.positions), C.c_float(0.0), 1000)
    lib.zlo_positions_set_gain_and_progress(C.byref(oc.positions), pid, C.c_float(0.5), C.c_float(0.1), 1000)
    assert lib.zlo_sync_audio_level(C.byref(m), C.byref(oc), 1000, C.byref(val)) == 1
    assert abs(val.value - 20 * np.log10(0.5)) < 1e-5
    lib.zlo_positions_remove(C.byref(oc.positions), pid, 1001)                                 # the voice ends: the model's peak is 0
    fired, now = [], 1000
    for _ in range(260):
        now += 31
        if lib.zlo_sync_audio_level(C.byref(m), C.byref(oc), now, C.byref(val)):
            fired.append(val.value)
    steps = int(np.ceil((100 + 20 * np.log10(0.5)) / (-20 * np.log10(0.94))))                 # ticks from -6.02 dB down to -100
    assert len(fired) == steps and fired[-1] == -100.0 and fired[-2] > -100.0
    assert all(abs((a - b) - 20 * np.log10(0.94)) < 1e-3 for a, b in zip(fired[1:-1], fired[:-2]))


def test_a_flood_of_calls_with_no_cycle_to_drain_them(zl, capfd):
    """play / stop / queue calls post into a bounded lock-free queue that the cycle drains (zl_handoff.h).  With no cycle running --
    a host that never pulls audio -- the 4096 cells fill up: further calls are dropped with ONE line on stderr, nothing blocks, nothing
    is overwritten, and the library stays usable (the setters do not go through the queue)."""
    L = np.zeros(4800, dtype=np.float32)
    c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None, len(L), 48000.0, b"flood")
    try:
        for i in range(3000):
            zl.ClipAudioSource_play(c, bool(i & 1))
            zl.ClipAudioSource_stopOnChannel(c, i % 10)
            zl.SyncTimer_queueClipToStartOnChannel(c, -1)
        err = capfd.readouterr().err
        assert err.count("request queue full") == 1
        assert zl.libzl_hotpath_dropped_requests() >= 9000 - 4096      # the host can ask how many were lost (the callers return void)
        zl.ClipAudioSource_setPan(c, 0.25)
        assert _product_params(zl, c).pan == 0.25
    finally:
        zl.ClipAudioSource_destroy(c)


# ------------------------------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
def test_more_voices_of_one_clip_than_position_rows_and_the_level_floor(zl, tmp_path):
    """48 voices of ONE clip (4 notes on each of the 12 sampler channels) against a positions model of 32 rows: from the 33rd voice on
    createPositionID returns row 31 although it belongs to another voice (ClipAudioSourcePositionsModel.cpp:82-99), those voices write
    their gain and progress over it and the first of them to end frees it for all.  Then everything is stopped and the injected clock
    runs 31 ms per cycle until the level chain has decayed to its -100 dB floor.  Audio, peakGain, firstProgress every cycle; every
    level / progress callback with its cycle: equal to the oracle's."""
    from libzl_amd import libzl
    from libzl_amd.engine import synthetic_clocks
    from scenario import engine_cmd
    rng = np.random.default_rng(77)
    lib = zo.load()
    now = [5_000_000]
    clock_cb = libzl.CLOCK_MS(lambda: now[0])
    zl.libzl_hotpath_set_clock_ms(clock_cb)
    try:
        zl.initJuce()
        assert zl.libzl_hotpath_status() == 0
        osyn = zo.OracleSynth(12, 8, 48000.0, 0)
        n = 7000
        L = rng.uniform(-1, 1, n).astype(np.float32) * np.linspace(1.0, 0.2, n).astype(np.float32)
        R = rng.uniform(-1, 1, n).astype(np.float32)
        c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, R.ctypes.data, n, 48000.0, b"many")
        oid = osyn.register_clip(L, R, 48000.0)
        oc = osyn.clips[oid]
        zl.ClipAudioSource_setLength(c, 0.23, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(0.23), 120)
        zl.ClipAudioSource_setADSRRelease(c, 0.01); lib.zlo_clip_set_adsr_release(C.byref(oc), C.c_float(0.01))
        eid = zl.ClipAudioSource_engineClip(c)
        got_lvl, got_prog, want_lvl, want_prog, cycle = [], [], [], [], [0]
        lv = libzl.CB(lambda db: got_lvl.append((cycle[0], db)))
        pg = libzl.CB(lambda s: got_prog.append((cycle[0], s)))
        zl.ClipAudioSource_setAudioLevelChangedCallback(c, lv)
        zl.ClipAudioSource_setProgressCallback(c, pg)
        m = zo.ClipMeter(); lib.zlo_clip_meter_init(C.byref(m))

        def both(**f):
            """one command through SyncTimer's step ring (delay 0: dispatched at the top of the next cycle, tick = the clock's playhead)
            and, in the same order, to the oracle's sampler"""
            zl.libzl_hotpath_schedule_clip_command(C.byref(engine_cmd(**dict(f, clip=eid))), 0)
            osyn.handle_clip_command(zo.clip_command(**dict(f, clip=oid)), 0)

        N = 128
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        val = C.c_float()
        max_rows = 0
        for k in range(330):
            cycle[0] = k
            now[0] += 3 if k < 70 else 31
            osyn.now_ms = now[0]
            if k in (1, 2, 3, 4):                                    # 12 more voices per cycle, one-shots at four pitches, issued from the
                for ch in reversed(range(-2, 10)):                   # LAST channel to the first: rows are handed out in command order
                    both(midiChannel=ch, midiNote=57 + 3 * ((k + ch) % 4), changeVolume=1, volume=0.3 + 0.05 * (k - 1), looping=0, startPlayback=1)
            if k == 30:                                               # a second generation while rows come and go
                for ch in range(-2, 10, 2):
                    both(midiChannel=ch, midiNote=72, changeVolume=1, volume=0.4, looping=1, startPlayback=1, stopPlayback=1)
            if k == 55:
                for ch in range(-2, 10):
                    for note in (57, 60, 63, 66, 72):
                        both(midiChannel=ch, midiNote=note, stopPlayback=1)
            clk = synthetic_clocks(1, N, 48000.0, start_block=k)
            assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
            bus, _ = osyn.render_batch(1, N, clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            max_rows = max(max_rows, sum(1 for i in range(32) if oc.positions.pos[i].id > -1))
            if lib.zlo_sync_audio_level(C.byref(m), C.byref(oc), now[0], C.byref(val)):
                want_lvl.append((k, val.value))
            if lib.zlo_sync_progress(C.byref(m), C.byref(oc), 1, now[0], C.byref(val)):
                want_prog.append((k, val.value))
            assert zl.ClipAudioSource_peakGain(c) == lib.zlo_positions_peak_gain(C.byref(oc.positions)), k
            assert zl.ClipAudioSource_firstProgress(c) == lib.zlo_positions_first_progress(C.byref(oc.positions)), k
        assert max_rows == 32                                         # the model was full while 48 voices played
        assert got_lvl == want_lvl and got_prog == want_prog
        assert want_lvl[-1][1] < -99.4 and want_lvl[-1][0] < 320      # the chain came within one 0.54 dB step of its -100 dB floor and fell silent
        assert len(want_lvl) > 150 and len(want_prog) >= 3           # (progress notifies at most every 100 ms)
        zl.ClipAudioSource_destroy(c)
        zl.shutdownJuce()
    finally:
        zl.libzl_hotpath_set_clock_ms(libzl.CLOCK_MS())


@pytest.mark.gpu
def test_destroying_a_playing_clip_and_queueing_with_a_host_owned_transport(zl):
    """~ClipAudioSource stops the clip everywhere (ClipAudioSource.cpp:207-210) and the sampler lets its voices finish their release tail;
    the engine's sound slot is released only when no voice plays it any more, and is then used again by the next clip.  In the same
    session: SyncTimer_queueClipToStart / _queueClipToStop while the HOST owns the transport (libzl_hotpath_process: no running timer
    here, the calls take effect in the next cycle, SyncTimer.cpp:815-860 with a paused timer) -- a start queued and un-queued inside one
    cycle never sounds."""
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(91)
    lib = zo.load()
    zl.initJuce()
    try:
        assert zl.libzl_hotpath_status() == 0
        osyn = zo.OracleSynth(12, 8, 48000.0, 0)

        def make(n, name, stereo=True):
            L = rng.uniform(-1, 1, n).astype(np.float32); R = rng.uniform(-1, 1, n).astype(np.float32) if stereo else None
            c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None if R is None else R.ctypes.data, n, 48000.0, name)
            oid = osyn.register_clip(L, R, 48000.0)
            zl.ClipAudioSource_setLength(c, 0.31, 120); lib.zlo_clip_set_length(C.byref(osyn.clips[oid]), C.c_float(0.31), 120)
            return c, oid
        a, oa = make(6000, b"a")
        b, ob = make(5000, b"b", stereo=False)
        eid_a = zl.ClipAudioSource_engineClip(a)
        play = lambda oid, ch: osyn.handle_clip_command(zo.clip_command(clip=oid, midiChannel=ch, midiNote=60, changeVolume=1, volume=1.0, looping=1, startPlayback=1, stopPlayback=1), 0)
        stop = lambda oid, ch: osyn.handle_clip_command(zo.clip_command(clip=oid, midiChannel=ch, midiNote=60, stopPlayback=1), 0)
        N = 128
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        c3 = None
        for k in range(70):
            if k == 0:
                zl.ClipAudioSource_play(a, True); play(oa, -2)
                zl.ClipAudioSource_playOnChannel(a, True, 4); play(oa, 4)
            if k == 2:
                zl.SyncTimer_queueClipToStartOnChannel(b, 1); play(ob, 1)
            if k == 6:                                               # queued and taken back within one cycle: it never sounds; the stop finds no voice
                zl.SyncTimer_queueClipToStartOnChannel(b, 3); zl.SyncTimer_queueClipToStopOnChannel(b, 3); stop(ob, 3)
            if k == 10:                                              # the clip goes while two voices play it
                zl.ClipAudioSource_destroy(a)
                for ch in [-2, -1] + list(range(10)):
                    stop(oa, ch)
            if k == 14:
                zl.SyncTimer_queueClipToStopOnChannel(b, 1); stop(ob, 1)
            if k == 50:                                              # a's voices are long gone (release 50 ms = 19 cycles): its slot is free again
                c3, o3 = make(4000, b"c")
                assert zl.ClipAudioSource_engineClip(c3) == eid_a
                zl.ClipAudioSource_playOnChannel(c3, True, 2); play(o3, 2)
            clk = synthetic_clocks(1, N, 48000.0, start_block=k)
            assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
            bus, _ = osyn.render_batch(1, N, clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            if k == 12:
                assert np.abs(bus[[0, 6]]).max() > 0                # a's release tails still sound on its two channels
            if k == 8:
                assert np.abs(bus[5]).max() == 0                    # channel 3: nothing ever started
        assert np.abs(bus[4]).max() > 0                             # the new clip plays on channel 2
        zl.ClipAudioSource_destroy(b); zl.ClipAudioSource_destroy(c3)
    finally:
        zl.shutdownJuce()


@pytest.mark.gpu
def test_set_channel_enabled_through_the_libzl_layer(zl):
    """SamplerSynth::setChannelEnabled (SamplerSynth.cpp:343-351) behind the libzl-named layer: a channel disabled while its loop plays
    falls silent and stands still, a clip played on it meanwhile waits, and everything goes on from where it stood when the channel
    comes back; peakGain / firstProgress of the clips follow the oracle's positions models (no updates while the channel is off)."""
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(93)
    lib = zo.load()
    zl.initJuce()
    try:
        assert zl.libzl_hotpath_status() == 0
        osyn = zo.OracleSynth(12, 8, 48000.0, 0)
        clips = []
        for i in range(2):
            n = 5000 + 400 * i
            L = rng.uniform(-1, 1, n).astype(np.float32); R = rng.uniform(-1, 1, n).astype(np.float32)
            c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, R.ctypes.data, n, 48000.0, b"en%d" % i)
            oid = osyn.register_clip(L, R, 48000.0)
            zl.ClipAudioSource_setLength(c, 0.27 + 0.04 * i, 120); lib.zlo_clip_set_length(C.byref(osyn.clips[oid]), C.c_float(0.27 + 0.04 * i), 120)
            clips.append((c, oid))
        play = lambda oid, ch: osyn.handle_clip_command(zo.clip_command(clip=oid, midiChannel=ch, midiNote=60, changeVolume=1, volume=1.0, looping=1, startPlayback=1, stopPlayback=1), 0)
        N = 128
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        now = 1000
        for k in range(40):
            now += 3; osyn.now_ms = now
            if k == 0:
                zl.ClipAudioSource_playOnChannel(clips[0][0], True, 1); play(clips[0][1], 1)
                zl.ClipAudioSource_play(clips[1][0], True); play(clips[1][1], -2)
            if k == 6:
                zl.SamplerSynth_setChannelEnabled(1, False); osyn.set_bus_enabled(3, False)
                zl.SamplerSynth_setChannelEnabled(42, False)                           # no such channel: ignored
            if k == 9:
                zl.ClipAudioSource_playOnChannel(clips[1][0], True, 1); play(clips[1][1], 1)      # waits on the disabled channel
            if k == 20:
                zl.SamplerSynth_setChannelEnabled(1, True); osyn.set_bus_enabled(3, True)
            clk = synthetic_clocks(1, N, 48000.0, start_block=k)
            assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
            bus, _ = osyn.render_batch(1, N, clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            if 6 <= k < 20:
                assert np.abs(bus[3]).max() == 0 and np.abs(bus[0]).max() > 0
            if k in (5, 25):
                assert np.abs(bus[3]).max() > 0
            for c, oid in clips:
                assert zl.ClipAudioSource_firstProgress(c) == lib.zlo_positions_first_progress(C.byref(osyn.clips[oid].positions)), k
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()
