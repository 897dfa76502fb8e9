"""The ClipCommand scheduling front-end (/root/reference/lib/SyncTimer.cpp:364-378,391-418,452-702,815-925,954-1048):
hand-derived known answers from the reference text on the oracle, the oracle against its independent numpy twin, a committed
golden vector, and the PRODUCT's scheduler (libzl_amd/csrc/zl_sched.h, host build) against the oracle.  CPU tier."""
import os

import numpy as np
import pytest

from oracle import np_restatement as npr
from oracle import zl_oracle as zo

f32 = np.float32
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def to_o(c: npr.Command) -> zo.ClipCommand:
    return zo.clip_command(clip=c.clip, midiNote=c.midi_note, midiChannel=c.midi_channel, startPlayback=int(c.start), stopPlayback=int(c.stop),
                           changeSlice=int(c.change_slice), slice=c.slice, changeLooping=int(c.change_looping), looping=int(c.looping),
                           changePitch=int(c.change_pitch), pitchChange=float(c.pitch_change), changeSpeed=int(c.change_speed),
                           speedRatio=float(c.speed_ratio), changeGainDb=int(c.change_gain_db), gainDb=float(c.gain_db),
                           changeVolume=int(c.change_volume), volume=float(c.volume))


def to_e(c: npr.Command):
    from scenario import engine_cmd
    o = to_o(c)
    return engine_cmd(**{f: getattr(o, f) for f in zo.CMD_FIELDS})


def tup(c: npr.Command):
    return zo.cmd_tuple(to_o(c))


def play(clip, ch=-2, loop=False):
    """the command ClipAudioSource::play builds (ClipAudioSource.cpp:415-429): looping WITHOUT changeLooping"""
    return npr.Command(clip=clip, midi_channel=ch, midi_note=60, change_volume=True, volume=f32(1.0), looping=loop, stop=loop, start=True)


def stop(clip, ch=-2):
    return npr.Command(clip=clip, midi_channel=ch, midi_note=60, stop=True)


PER = 5333      # 256 frames at 48 kHz


def cycles(t, n, k0=0, N=256):
    out = []
    for k in range(k0, k0 + n):
        out.append(t.process(N, k * PER, (k + 1) * PER))
    return out


# ---- hand-derived known answers (the reference text, on the C oracle) ---------------------------------------------------------
def test_play_then_stop_in_one_step_keeps_playing():
    """SyncTimer.cpp:1015-1047: the stop is equivalent to the queued play (same clip, note 60, channel -2) and folds into it;
    stopPlayback is NOT among the copied fields, so one command reaches the sampler: start, no stop."""
    t = zo.OracleSyncTimer()
    t.schedule(to_o(play(0)), 0)
    t.schedule(to_o(stop(0)), 0)
    got = [d for cyc in cycles(t, 3) for d in cyc]
    assert len(got) == 1
    c, tick = got[0]
    assert (c.startPlayback, c.stopPlayback, c.changeVolume, c.volume) == (1, 0, 1, 1.0) and tick == 0
    # the other way round: the play folds into the queued stop -> ONE command that stops and starts (volume copied, looping not)
    t.schedule(to_o(stop(0)), 0)
    t.schedule(to_o(play(0, loop=True)), 0)
    got = [d for cyc in cycles(t, 3, k0=3) for d in cyc]
    assert len(got) == 1
    c, _ = got[0]
    assert (c.startPlayback, c.stopPlayback, c.changeVolume, c.volume, c.looping, c.changeLooping) == (1, 1, 1, 1.0, 0, 0)


def test_play_twice_in_one_step_is_one_voice_and_stop_everywhere_is_twelve_commands():
    t = zo.OracleSyncTimer()
    t.schedule(to_o(play(1, loop=True)), 0)
    t.schedule(to_o(play(1, loop=True)), 0)
    got = [d for cyc in cycles(t, 3) for d in cyc]
    assert len(got) == 1 and got[0][0].startPlayback == 1
    # ClipAudioSource::stop() (midiChannel -3, :438-453): channels -2, -1, 0..9 -- no two of them are equivalent
    for ch in [-2, -1] + list(range(10)):
        t.schedule(to_o(stop(1, ch)), 0)
    got = [d for cyc in cycles(t, 3, k0=3) for d in cyc]
    assert [c.midiChannel for c, _ in got] == [-2, -1] + list(range(10))
    # different notes / different slices are not equivalent; equal slices are, whatever note and channel say (ClipCommand.h:33-39)
    a = npr.Command(clip=2, midi_channel=0, midi_note=60, start=True, change_slice=True, slice=3)
    b = npr.Command(clip=2, midi_channel=5, midi_note=72, change_slice=True, slice=3, change_pitch=True, pitch_change=f32(2.0))
    c = npr.Command(clip=2, midi_channel=0, midi_note=61, start=True)
    for x in (a, b, c):
        t.schedule(to_o(x), 0)
    got = [d for cyc in cycles(t, 3, k0=6) for d in cyc]
    assert len(got) == 2
    assert (got[0][0].changePitch, got[0][0].pitchChange, got[0][0].midiChannel, got[0][0].midiNote) == (1, 2.0, 0, 60)


def test_paused_timer_steps_delay_and_playhead_getters():
    """Paused (the initial state): delayedStep(d) = read head + d + 1 (:366-368); process plays one step per 5208.33 us
    (120 bpm: 60e9 / (120 * 96) ns, :180-183,484) whenever stepNextPlaybackPosition < next_usecs (:512); the getters return
    the read head and stepNextPlaybackPosition (:990-1004); the dispatch tick is jackPlayhead, which does not move (:553-558,660)."""
    t = zo.OracleSyncTimer()
    t0 = 1_000_000
    assert t.process(256, t0, t0 + PER) == []            # steps 0 and 1 are played: t0 and t0 + 5208 lie before t0 + 5333
    ck = t.clock(t0, t0 + PER)
    assert (ck.jackPlayhead, ck.jackPlayheadUsecs, ck.jackSubbeatLengthInMicroseconds) == (2, t0 + 10416, 5208)
    t.schedule(to_o(play(0)), 0)                        # -> step 3
    t.schedule(to_o(play(1)), 2)                        # -> step 5
    seen = []
    for k in range(1, 6):
        r = t.process(256, t0 + k * PER, t0 + (k + 1) * PER)
        seen.append([(c.clip, tick) for c, tick in r])
        # u64 += double: the step clock advances by 5208.333... and is truncated at every step
    assert seen == [[], [(0, 0)], [], [(1, 0)], []]
    assert t.clock(0, 0).jackPlayhead == 7
    # 7 steps: the truncating accumulation of :671, step by step
    x = t0
    for _ in range(7):
        x = int(float(x) + 5208.333)
    assert t.clock(0, 0).jackPlayheadUsecs == x


def test_running_timer_playhead_ticks_and_schedule_ahead():
    """start(bpm) (:870-879) un-pauses; jackPlayhead then counts the steps played (:660-667), commands are dispatched with it,
    delay 0 addresses stepReadHeadOnStart + max(cumulativeBeat, jackPlayhead + 1) (:371) and the timer thread keeps
    cumulativeBeat 2 x scheduleAheadAmount ahead of the playhead (:395)."""
    t = zo.OracleSyncTimer()
    t.set_latency(256, 48000.0)                          # 5 ms -> scheduleAheadAmount = 5e6 / 5208333 + 1 = 1
    assert t.t.contents.scheduleAheadAmount == 1
    t.start(120)
    t.schedule(to_o(play(0)), 0)                        # cumulativeBeat 0, playhead 0 -> step 0 + max(0, 1) = 1
    r0 = t.process(256, 0, PER)                         # steps 0, 1 (0 and 5208 < 5333)
    assert [(c.clip, tick) for c, tick in r0] == [(0, 1)]
    assert t.clock(0, PER).jackPlayhead == 2 and t.clock(0, PER).jackPlayheadUsecs == 10416
    t.timer_callback()
    assert t.t.contents.cumulativeBeat == 4              # playhead 2 + 2 * 1
    t.schedule(to_o(play(1)), 0)                        # -> max(4, 3) = step 4
    t.schedule(to_o(play(2)), 96)                       # -> step 100
    seen = {}
    for k in range(1, 110):
        for c, tick in t.process(256, k * PER, (k + 1) * PER):
            seen[c.clip] = tick
        t.timer_callback()
    assert seen == {1: 4, 2: 100}


def test_set_bpm_reaches_the_step_clock_through_a_timer_command():
    """setBpm (:954-975) changes the subbeat length the voices read at once, and the step clock when the SetBpmOperation it
    scheduled is played (:606-612,634-639) -- with the clamp to [50, 200] there and a second command when the clamp bites."""
    t = zo.OracleSyncTimer()
    t.process(256, 1000, 1000 + PER)
    t.set_bpm(240)
    assert t.clock(0, 0).jackSubbeatLengthInMicroseconds == 60_000_000_000 // (240 * 96) // 1000 == 2604
    for k in range(1, 4):
        t.process(256, 1000 + k * PER, 1000 + (k + 1) * PER)
    assert t.t.contents.bpm == 200 and t.t.contents.jackPlayheadBpm == 200.0
    assert t.clock(0, 0).jackSubbeatLengthInMicroseconds == 60_000_000_000 // (200 * 96) // 1000 == 3125


def test_stop_reschedules_unplayed_commands_silently():
    """SyncTimer::stop (:881-925): commands of unplayed steps are re-scheduled at delay 0 with volume 0.  Those of the step
    behind the read head fold into themselves and die with it (marked played); later steps' commands clear that step
    (ensureFresh) and are dispatched from it at the next cycle, muted."""
    t = zo.OracleSyncTimer()
    t.start(120)
    t.process(256, 0, PER)
    t.timer_callback()
    t.schedule(to_o(play(0)), 0)            # step 4
    t.schedule(to_o(play(1)), 50)           # step 54
    t.stop()
    got = [d for cyc in cycles(t, 4, k0=1) for d in cyc]
    assert [(c.clip, c.startPlayback, c.changeVolume, c.volume, tick) for c, tick in got] == [(0, 1, 1, 0.0, 0), (1, 1, 1, 0.0, 0)]
    # only the step behind the read head holds a command: it folds into itself and is never dispatched
    t.schedule(to_o(play(2)), 0)
    t.stop()
    assert [d for cyc in cycles(t, 4, k0=5) for d in cyc] == []


def test_queue_clip_to_start_waits_for_the_bar_and_stop_removes_queued_commands():
    t = zo.OracleSyncTimer()
    t.set_latency(256, 48000.0)
    t.start(120)
    for k in range(10):
        t.process(256, k * PER, (k + 1) * PER)
        t.timer_callback()
    cb = t.t.contents.cumulativeBeat
    t.queue_start(0, -1)                    # :829-831: at the next multiple of 384 ticks of cumulativeBeat
    t.queue_start(1, 0)
    t.queue_stop(1, 0)                      # :837-850 removes the queued start of clip 1; the stop itself goes out at once
    seen = []
    for k in range(10, 420):
        for c, tick in t.process(256, k * PER, (k + 1) * PER):
            seen.append((c.clip, c.startPlayback, c.stopPlayback, c.looping, tick))
        t.timer_callback()
    due = cb + (384 - cb % 384)
    assert seen == [(1, 0, 1, 0, cb), (0, 1, 1, 1, due)]


# ---- oracle == numpy twin == golden == product ------------------------------------------------------------------------------------
def rand_cmd(rng):
    k = rng.integers(0, 5)
    c = npr.Command(clip=int(rng.integers(0, 3)), midi_channel=int(rng.integers(-2, 2)), midi_note=int(rng.choice([60, 60, 60, 62])))
    if k == 0:
        c.start = True; c.change_volume = True; c.volume = f32(1.0); c.looping = bool(rng.integers(0, 2)); c.stop = c.looping
    elif k == 1:
        c.stop = True
    elif k == 2:
        c.change_volume = True; c.volume = f32(rng.uniform(0, 1))
    elif k == 3:
        c.change_slice = True; c.slice = int(rng.integers(0, 4)); c.start = True; c.change_looping = True; c.looping = True
    else:
        c.change_pitch = True; c.pitch_change = f32(rng.uniform(-1, 1)); c.change_gain_db = True; c.gain_db = f32(-3.0)
        c.change_speed = bool(rng.integers(0, 2)); c.speed_ratio = f32(1.5)
    return c


def random_session(seed, ncycles=300, xruns=False):
    """A seeded list of operations per cycle: [("schedule", Command, delay) | ("start", bpm) | ("stop",) | ("bpm", bpm) |
    ("qstart", clip, ch) | ("qstop", clip, ch) | ("tick",) | ("gap", cycles)].  xruns: now and then JACK time jumps ahead by a few or by
    hundreds of cycles before a cycle (an xrun, a suspended host): the step loop then catches up one step per FRAME until the cycle's
    frames run out (SyncTimer.cpp:512,517-523) and goes on in the next cycle."""
    rng = np.random.default_rng(seed)
    N = int(rng.choice([64, 128, 256, 1024])); fs = float(rng.choice([44100.0, 48000.0, 96000.0]))
    t0 = int(rng.integers(0, 3)) * 1000003
    ops = []
    for _ in range(ncycles):
        cyc = []
        if xruns and rng.random() < 0.04:
            cyc.append(("gap", int(rng.choice([1, 2, 9, 60, 700]))))
            if rng.random() < 0.7:   # a clip queued right behind that cycle, before the timer thread has caught up with the playhead:
                cyc.append(("late_qstart", int(rng.integers(0, 3)), int(rng.integers(-2, 2))))   # the next bar lies BEHIND the playhead (:823-831)
        for _ in range(int(rng.integers(0, 4)) if rng.random() < 0.3 else 0):
            a = int(rng.integers(0, 12))
            if a < 6: cyc.append(("schedule", rand_cmd(rng), int(rng.choice([0, 0, 0, 1, 2, 7, 96]))))
            elif a == 6: cyc.append(("start", int(rng.choice([60, 90, 120, 174, 220, 40]))))
            elif a == 7: cyc.append(("stop",))
            elif a == 8: cyc.append(("bpm", int(rng.choice([50, 100, 120, 140, 250]))))
            elif a == 9: cyc.append(("qstart", int(rng.integers(0, 3)), int(rng.integers(-2, 2))))
            elif a == 10: cyc.append(("qstop", int(rng.integers(0, 3)), int(rng.integers(-2, 2))))
            else: cyc.append(("tick",))
        ops.append(cyc)
    return N, fs, t0, ops


def run_session(impl, conv, N, fs, t0, ops, running):
    """impl: OracleSyncTimer / SyncTimerModel / ProductScheduler -> ([(cmd tuple, tick)] per cycle, [clock triple] per cycle)"""
    per = int(round(1e6 * N / fs))
    impl.set_latency(N, fs)
    disp, clocks = [], []
    for k, cyc in enumerate(ops):
        for op in cyc:
            if op[0] == "gap": t0 += op[1] * per
            elif op[0] == "late_qstart": pass
            elif op[0] == "schedule": impl.schedule(conv(op[1]), op[2])
            elif op[0] == "start": impl.start(op[1])
            elif op[0] == "stop": impl.stop()
            elif op[0] == "bpm": impl.set_bpm(op[1])
            elif op[0] == "qstart": impl.queue_start(op[1], op[2])
            elif op[0] == "qstop": impl.queue_stop(op[1], op[2])
            else: impl.timer_callback()
        cu, nx = t0 + k * per, t0 + (k + 1) * per
        disp.append(impl.process(N, cu, nx))
        clocks.append(impl.clock(cu, nx) if not hasattr(impl, "ext_process") else impl.clock())
        for op in cyc:
            if op[0] == "late_qstart": impl.queue_start(op[1], op[2])
        if running(impl):
            impl.timer_callback()
    return disp, clocks


def oracle_session(N, fs, t0, ops):
    o = zo.OracleSyncTimer()
    disp, clocks = run_session(o, to_o, N, fs, t0, ops, lambda i: not i.t.contents.threadPaused)
    o.close()
    return ([[(zo.cmd_tuple(c), t) for c, t in cyc] for cyc in disp],
            [(c.jackPlayhead, c.jackPlayheadUsecs, c.jackSubbeatLengthInMicroseconds) for c in clocks])


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_its_numpy_twin(seed, built):
    N, fs, t0, ops = random_session(seed)
    od, oc = oracle_session(N, fs, t0, ops)
    m = npr.SyncTimerModel()
    md, mc = run_session(m, lambda c: c, N, fs, t0, ops, lambda i: not i.paused)
    assert [[(tup(c), t) for c, t in cyc] for cyc in md] == od
    assert [(c.playhead, c.playhead_usecs, c.subbeat_usecs) for c in mc] == oc
    assert sum(len(c) for c in od) > 10


@pytest.mark.parametrize("seed", range(12, 36))
def test_product_scheduler_equals_the_oracle(seed, built):
    """libzl_amd/csrc/zl_sched.h (what libzl_hotpath_cycle runs), built for the host, against zlo_sync_timer_*: every dispatched
    command with every field and its tick, and the clock triple the voices read, cycle by cycle."""
    from cpu_harness.sched import ProductScheduler
    N, fs, t0, ops = random_session(seed)
    od, oc = oracle_session(N, fs, t0, ops)
    p = ProductScheduler()
    pd, pc = run_session(p, to_e, N, fs, t0, ops, lambda i: None)
    # (the product harness cannot be asked whether it runs: redo with the oracle's paused flags)
    p.close()
    flags = []
    o = zo.OracleSyncTimer()
    run_session(o, to_o, N, fs, t0, ops, lambda i: flags.append(not i.t.contents.threadPaused) or flags[-1])
    o.close()
    it = iter(flags)
    p = ProductScheduler()
    pd, pc = run_session(p, to_e, N, fs, t0, ops, lambda i: next(it))
    p.close()
    assert pd == od
    assert pc == oc


@pytest.mark.parametrize("seed", range(100, 112))
def test_time_jumps_oracle_numpy_and_product_agree(seed, built):
    """Sessions with xruns (random_session(xruns=True)): the catch-up of the step loop -- one step per frame while the step clock is
    behind the cycle, the exit when the cycle's frames are used up (SyncTimer.cpp:512), a step placed behind the frames already taken
    (:524-531) -- on the oracle, its numpy twin and the product's scheduler."""
    from cpu_harness.sched import ProductScheduler
    N, fs, t0, ops = random_session(seed, ncycles=260, xruns=True)
    assert any(op[0] == "gap" for cyc in ops for op in cyc)
    od, oc = oracle_session(N, fs, t0, ops)
    m = npr.SyncTimerModel()
    md, mc = run_session(m, lambda c: c, N, fs, t0, ops, lambda i: not i.paused)
    assert [[(tup(c), t) for c, t in cyc] for cyc in md] == od
    assert [(c.playhead, c.playhead_usecs, c.subbeat_usecs) for c in mc] == oc
    flags = []
    o = zo.OracleSyncTimer()
    run_session(o, to_o, N, fs, t0, ops, lambda i: flags.append(not i.t.contents.threadPaused) or flags[-1])
    o.close()
    it = iter(flags)
    p = ProductScheduler()
    pd, pc = run_session(p, to_e, N, fs, t0, ops, lambda i: next(it))
    p.close()
    assert pd == od
    assert pc == oc


def test_scheduler_golden_vector(built):
    """tests/golden/s1_scheduler.npz (written by the numpy twin, tests/golden/make_golden.py): the C oracle and the product's
    scheduler reproduce every dispatched command and every clock triple."""
    from cpu_harness.sched import ProductScheduler
    from golden_util import scheduler_session_from_golden
    g = np.load(os.path.join(GOLDEN, "s1_scheduler.npz"))
    N, fs, t0, ops = scheduler_session_from_golden(g)
    want_d = [[(tuple(row[:-1]), int(row[-1])) for row in g["dispatch"][g["dispatch_cycle"] == k]] for k in range(len(ops))]
    want_c = [tuple(int(x) for x in row) for row in g["clocks"]]
    od, oc = oracle_session(N, fs, t0, ops)

    def norm(d):
        return [[(tuple(float(x) for x in c), t) for c, t in cyc] for cyc in d]
    assert norm(od) == norm(want_d) and oc == want_c
    it = iter(g["running"].astype(bool).tolist())
    p = ProductScheduler()
    pd, pc = run_session(p, to_e, N, fs, t0, ops, lambda i: next(it))
    p.close()
    assert norm(pd) == norm(want_d) and pc == want_c


def test_host_transport_schedule_merges_per_due_tick(built):
    """libzl_hotpath_process's schedule (the host's SyncTimer owns the transport): commands due at the same tick form one step
    with the merge of scheduleClipCommand; a delay counts ticks of the host's playhead."""
    from cpu_harness.sched import ProductScheduler, cmd_tuple
    p = ProductScheduler()
    p.ext_process(10_000)                               # the host's playhead when the requests are taken
    for c in (play(0), stop(0), play(1, loop=True), play(1, loop=True)):
        p.ext_schedule(to_e(c), 0)
    p.ext_schedule(to_e(play(2)), 3)
    lib = zo.load()
    import ctypes as C
    lst = (zo.ClipCommand * 8)(); n = C.c_int32(0)
    for c in (play(0), stop(0), play(1, loop=True), play(1, loop=True)):
        lib.zlo_step_schedule(lst, C.byref(n), C.byref(to_o(c)))
    assert n.value == 2
    assert p.ext_process(10_000) == [(zo.cmd_tuple(lst[i]), 10_000) for i in range(2)]
    assert p.ext_process(10_002) == []
    got = p.ext_process(10_004)                         # due at 10 003: the first cycle whose playhead has reached it
    assert [(c[0], t) for c, t in got] == [(2, 10_004)]
    p.close()
