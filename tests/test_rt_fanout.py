"""GPU tier: the JackPassthrough fan-out ON THE REAL-TIME CYCLE (reference JackPassthrough.cpp:45-115, setters libzl.h:117-175;
VERDICT r3 "What's missing" 2), the resident kernel at periods above 256 frames (SamplerSynth.cpp:116-148 takes any nframes), and the
process-wide capacity of resident kernels.

zlhip_render_fanout delivers, per cycle, the bus AND the three output pairs of the passthrough client behind it, written by the
resident kernel (or the launched kernels) from the registers that hold the mix; the parameters are taken per cycle -- changed from
another thread through the JackPassthrough_set* names while the session plays -- without a HIP call and without evicting the kernel.
Every cycle's six rows are held, bit for bit, to zlo_passthrough_process of the oracle's bus."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from oracle import zl_oracle as zo
from scenario import engine_cmd, oracle_cmd, random_scene, snapshot_clip

pytestmark = pytest.mark.gpu


def _oracle_fanout(bus, params):
    lib = zo.load()
    B, _, n = bus.shape
    out = np.zeros((B, 6, n), dtype=np.float32)
    for b in range(B):
        rows = [np.zeros(n, dtype=np.float32) for _ in range(6)]
        arr = (C.c_void_p * 6)(*[o.ctypes.data for o in rows])
        p = zo.Passthrough(params[b].dry_amount, params[b].wet_fx1_amount, params[b].wet_fx2_amount, params[b].pan_amount, params[b].muted)
        L = np.ascontiguousarray(bus[b, 0]); R = np.ascontiguousarray(bus[b, 1])
        lib.zlo_passthrough_process(C.byref(p), L.ctypes.data, R.ctypes.data, arr, n)
        out[b] = np.stack(rows)
    return out


def _zoo():
    from libzl_amd import PassthroughParams as P
    # every branch of JackPassthrough.cpp:55-113: copy (amount 1, pan 0), silence (amount 0, pan 0), the multiply, mute, negative
    # amounts, pan beyond +-1
    return [P(1.0, 0.0, 0.5, 0.0, 0), P(0.8, 1.0, -1.25, -0.3, 0), P(1.0, 1.0, 1.0, 0.0, 1), P(-0.5, 2.0, 0.0, 1.5, 0), P(1.0, 1.0, 1.0, 0.0, 0),
            P(0.0, 0.0, 0.0, 0.0, 0), P(0.3, 0.6, 0.9, 0.75, 0)]


def _engines(sc, max_frames, mode=0):
    from libzl_amd import SamplerSynth
    ref = zo.OracleSynth(1, 1, sc.fs, mode, max_sounds=max(8, len(sc.sounds)))
    osyn = zo.OracleSynth(sc.num_buses, sc.voices_per_bus, sc.fs, mode, max_sounds=max(8, len(sc.sounds)))
    syn = SamplerSynth(num_buses=sc.num_buses, voices_per_bus=sc.voices_per_bus, mode=mode, playback_sample_rate=sc.fs, max_frames=max_frames,
                       max_batch_blocks=4, max_sounds=max(8, len(sc.sounds)),
                       sound_arena_bytes=max(1 << 20, sum((s[0].shape[0] + 16) * 8 for s in sc.sounds) + (1 << 16)))
    for i, (L, R, sr) in enumerate(sc.sounds):
        assert ref.register_clip(L, R, sr) == i and syn.register_clip(L, R, sr) == i and osyn.register_clip(L, R, sr) == i
        if i in sc.clip_setup:
            sc.clip_setup[i](ref.lib, ref.clips[i]); sc.clip_setup[i](osyn.lib, osyn.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    for ev in sc.events[0]:
        syn.handle_clip_command(engine_cmd(**ev[1]), ev[2]); osyn.handle_clip_command(oracle_cmd(**ev[1]), ev[2])
    return syn, osyn


def _clock(sc, t, N):
    from libzl_amd._abi import Clock
    per = int(round(1e6 * N / sc.fs))
    clk = Clock(); clk.current_usecs = t; clk.next_usecs = t + per; clk.jack_playhead = 0; clk.jack_playhead_usecs = 0
    clk.jack_subbeat_length_usecs = ((60000000000) // (sc.bpm * 96)) // 1000
    return clk, t + per


CASES = {
    # name: (scene kwargs, period, mode, environment, resident launches expected (None = do not care))
    "narrow_128":        (dict(num_buses=4, voices_per_bus=8, nclips=14), 128, 0, {}, 1),
    "narrow_period_100": (dict(num_buses=5, voices_per_bus=8, nclips=14), 100, 0, {}, 1),           # lanes behind the block's end store nothing
    "narrow_period_17":  (dict(num_buses=3, voices_per_bus=8, nclips=10), 17, 0, {}, 1),
    "delay_fixed":       (dict(num_buses=3, voices_per_bus=8, nclips=10), 128, 2, {}, 1),
    "two_frame_tiles":   (dict(num_buses=3, voices_per_bus=8, nclips=10), 512, 0, {}, 1),           # the resident workgroup walks two tiles
    "four_frame_tiles":  (dict(num_buses=2, voices_per_bus=8, nclips=8), 1024, 0, {}, 1),
    "ragged_tiles_441":  (dict(num_buses=3, voices_per_bus=8, nclips=10), 441, 0, {}, 1),
    "wide_resident":     (dict(num_buses=2, voices_per_bus=40, nclips=50), 128, 0, {}, 1),          # one workgroup per voice, the bus's last arrival fans out
    "wide_resident_512": (dict(num_buses=2, voices_per_bus=32, nclips=40), 512, 0, {}, 1),
    "launched":          (dict(num_buses=4, voices_per_bus=8, nclips=14), 128, 0, {"ZL_RT_PERSISTENT": "0"}, 0),
    "launched_wide_512": (dict(num_buses=2, voices_per_bus=40, nclips=50), 512, 0, {"ZL_RT_PERSISTENT": "0"}, 0),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_fanout_on_the_real_time_cycle(built, case):
    kw, N, mode, env, want_starts = CASES[case]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sc = random_scene(900 + N, nframes=64, nblocks=1, events=False, mode=mode, **kw)
        syn, osyn = _engines(sc, max_frames=max(64, N), mode=mode)
        zoo = _zoo()
        B = sc.num_buses
        t = 0
        cycles = 40
        for k in range(cycles):
            # a knob moves every third cycle; in between the table is the same (the kernel keeps what it has)
            params = [zoo[(b + k // 3) % len(zoo)] for b in range(B)]
            clk, t = _clock(sc, t, N)
            if k % 7 == 5:
                L, R = syn.process(N, clk)                            # a cycle without fan-out in between: same kernel, same cycle
                fan = None
            else:
                L, R, fan = syn.process_fanout(N, clk, params)
            bus, _ = osyn.render_batch(1, N, [clk])
            assert np.array_equal(L.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(R.view(np.int32), bus[:, 1].view(np.int32)), (case, k)
            if fan is not None:
                want = _oracle_fanout(bus, params)
                assert np.array_equal(fan.view(np.int32), want.view(np.int32)), (case, k, np.argwhere(fan.view(np.int32) != want.view(np.int32))[:4].tolist())
        # the meters of the last cycle are the oracle's too (the fused scan carries on across frame tiles)
        peaks = syn.block_peaks()
        exp = np.abs(np.float32(131072.0) * bus).astype(np.int64).max(axis=2)
        assert np.array_equal(peaks[-1].astype(np.int64), exp), case
        starts, done = syn.rt_stats()
        syn.close()
        if want_starts is not None:
            assert starts == want_starts and done == (cycles if want_starts else 0), (case, starts, done)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_resident_kernel_at_long_periods_rms_and_reports(built):
    """512- and 1024-frame cycles through the resident kernel: RMS extension (a defined summation order over ALL tiles of the block)
    and the voice reports (peak over the whole block) against the oracle, with a period change in between."""
    sc = random_scene(977, num_buses=3, voices_per_bus=8, nclips=12, nframes=64, nblocks=1, events=False)
    syn, osyn = _engines(sc, max_frames=1024)
    lib = zo.load()
    t = 0
    for k, N in enumerate([512, 512, 1024, 1024, 256, 1024, 512, 441, 1000]):
        clk, t = _clock(sc, t, N)
        L, R = syn.process(N, clk)
        bus, orep = osyn.render_batch(1, N, [clk])
        assert np.array_equal(L.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(R.view(np.int32), bus[:, 1].view(np.int32)), (k, N)
        lv = syn.levels_tick(-1, -1)
        for b in range(sc.num_buses):
            l = np.ascontiguousarray(bus[b, 0]); r = np.ascontiguousarray(bus[b, 1])
            assert lv[b].rms_a == lib.zlo_block_rms(l.ctypes.data, N, 1), (k, N, b)
            assert lv[b].rms_b == lib.zlo_block_rms(r.ctypes.data, N, 1), (k, N, b)
        rep = syn.voice_reports()
        for v in range(sc.num_buses * sc.voices_per_bus):
            assert (rep[v].valid, rep[v].gain, rep[v].progress) == (orep[v].valid, orep[v].gain, orep[v].progress), (k, N, v)
    starts, done = syn.rt_stats()
    syn.close()
    assert done == 9 and starts >= 5


def test_two_wide_engines_share_the_device(built):
    """Two engines whose resident kernels each fit the device alone but not together (2 x 256 voices, one voice per workgroup): the
    second one renders with launches instead of waiting for slots the first one's spinners hold (real time, second item).
    Interleaved cycles, both bit-exact, no error; zlhip_rt_residency tells which one is resident."""
    old = os.environ.pop("ZL_RT_WIDE", None)
    try:
        pairs = []
        for i in range(2):
            sc = random_scene(1200 + i, num_buses=4, voices_per_bus=64, nclips=100, nframes=64, nblocks=1, events=False)
            pairs.append((sc,) + _engines(sc, max_frames=128))
        t = [0, 0]
        for k in range(30):
            for i, (sc, syn, osyn) in enumerate(pairs):
                clk, t[i] = _clock(sc, t[i], 128)
                L, R = syn.process(128, clk)
                bus, _ = osyn.render_batch(1, 128, [clk])
                assert np.array_equal(L.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(R.view(np.int32), bus[:, 1].view(np.int32)), (k, i)
        res = [syn.rt_residency() for (_, syn, _) in pairs]
        stats = [syn.rt_stats() for (_, syn, _) in pairs]
        shares = [r[1] for r in res]
        assert all(0.0 < s <= 0.75 for s in shares), shares
        if shares[0] + shares[1] > 0.75:
            # they do not fit together: at any time ONE of them is resident, the other renders with launches.  (Which one may change
            # hands: a device-wide wait -- the other engine growing a pinned buffer -- makes the resident kernel leave for a moment, and
            # the next engine to ask finds the room.)
            assert res[0][0] != res[1][0], res
            assert 0 < stats[0][1] + stats[1][1] <= 60 and max(stats[0][1], stats[1][1]) >= 20, stats
        else:
            assert stats[0][1] == 30 and stats[1][1] == 30, stats
        # the resident engine goes away: the other one's next cycles find room
        gone = 0 if res[0][0] else 1
        pairs[gone][1].close()
        sc, syn, osyn = pairs[1 - gone]
        before = syn.rt_stats()[1]
        for k in range(5):
            clk, t[1 - gone] = _clock(sc, t[1 - gone], 128)
            L, R = syn.process(128, clk)
            bus, _ = osyn.render_batch(1, 128, [clk])
            assert np.array_equal(L.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(R.view(np.int32), bus[:, 1].view(np.int32)), k
        assert syn.rt_residency()[0] and syn.rt_stats()[1] >= before + 4
        syn.close()
    finally:
        if old is not None:
            os.environ["ZL_RT_WIDE"] = old


def test_the_jackpassthrough_names_drive_the_fanout_of_a_playing_session(built):
    """Through the libzl names: 200 cycles of libzl_hotpath_cycle_fanout on the 12 x 8 engine, dry / wet / pan / mute of the passthrough
    clients (GlobalPlayback = channel -1 = bus 1, FXPassthrough-Channel<n> = channel n-1 = bus n+1) changed mid-session by ANOTHER
    thread's JackPassthrough_set* calls -- incl. the 0 and 1 fast paths -- every cycle's 12 x 6 rows against the oracle's passthrough of
    the oracle's bus.  The resident kernel is launched once for the whole session."""
    from libzl_amd import libzl, PassthroughParams as P
    from libzl_amd.engine import synthetic_clocks
    zl = libzl.load()
    lib = zo.load()
    rng = np.random.default_rng(77)
    zl.initJuce()
    assert zl.libzl_hotpath_status() == 0
    dropped0 = zl.libzl_hotpath_dropped_requests()
    try:
        osyn = zo.OracleSynth(12, 8, 48000.0, 0)
        clips = []
        for i in range(5):
            n = 6000 + 900 * i
            L = rng.uniform(-1, 1, n).astype(np.float32); R = rng.uniform(-1, 1, n).astype(np.float32) if i != 2 else None
            c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None if R is None else R.ctypes.data, n, 48000.0, f"fan{i}".encode())
            oid = osyn.register_clip(L, R, 48000.0)
            oc = osyn.clips[oid]
            zl.ClipAudioSource_setLength(c, 0.21 + 0.03 * i, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(0.21 + 0.03 * i), 120)
            zl.ClipAudioSource_setPan(c, -0.6 + 0.3 * i); lib.zlo_clip_set_pan(C.byref(oc), C.c_float(-0.6 + 0.3 * i))
            clips.append((c, oid))
        chans = [-2, -1, 0, 3, 9]                                   # buses 0, 1, 2, 5, 11
        for (c, oid), ch in zip(clips, chans):
            zl.ClipAudioSource_playOnChannel(c, True, ch)
            osyn.handle_clip_command(zo.clip_command(clip=oid, midiChannel=ch, midiNote=60, changeVolume=1, volume=1.0, looping=1, startPlayback=1, stopPlayback=1), 0)
        # host view of the clients' members (defaults: JackPassthrough.cpp:27-31)
        state = {ch: dict(dry=1.0, fx1=1.0, fx2=1.0, pan=0.0, muted=False) for ch in range(-1, 10)}
        script = {
            10: [(-1, "Dry", 0.5)], 11: [(-1, "Pan", -0.25)], 25: [(0, "WetFx1", 0.0), (0, "WetFx2", 1.0)],       # memset / memcpy fast paths (pan 0)
            40: [(3, "Muted", True)], 55: [(3, "Muted", False), (3, "Pan", 0.4)], 70: [(9, "Dry", 0.0), (9, "Pan", 1.5)],
            90: [(0, "Pan", -1.0)], 120: [(-1, "Dry", 1.0), (-1, "Pan", 0.0)], 150: [(ch, "WetFx2", 0.33) for ch in range(-1, 10)],
            151: [(9, "Pan", 0.0)], 180: [(5, "Dry", -0.75)],
        }
        key = {"Dry": "dry", "WetFx1": "fx1", "WetFx2": "fx2", "Pan": "pan", "Muted": "muted"}

        def turn(knobs):
            for ch, what, val in knobs:
                (zl.JackPassthrough_setMuted if what == "Muted" else getattr(zl, f"JackPassthrough_set{what}Amount"))(ch, val)

        N = 128
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32); fan = np.zeros((12, 6, N), dtype=np.float32)
        for k in range(200):
            if k in script:
                th = threading.Thread(target=turn, args=(script[k],))          # another thread's calls, finished before the cycle starts
                th.start(); th.join()
                for ch, what, val in script[k]:
                    state[ch][key[what]] = val
            clk = synthetic_clocks(1, N, 48000.0, start_block=k)
            assert zl.libzl_hotpath_process_fanout(N, clk, outL.ctypes.data, outR.ctypes.data, fan.ctypes.data) == 0
            bus, _ = osyn.render_batch(1, N, clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            params = [P(1.0, 1.0, 1.0, 0.0, 0)] + [P(np.float32(state[b - 2]["dry"]), np.float32(state[b - 2]["fx1"]), np.float32(state[b - 2]["fx2"]),
                                                       np.float32(state[b - 2]["pan"]), 1 if state[b - 2]["muted"] else 0) for b in range(1, 12)]
            want = _oracle_fanout(bus, params)
            assert np.array_equal(fan.view(np.int32), want.view(np.int32)), (k, np.argwhere(fan.view(np.int32) != want.view(np.int32))[:4].tolist())
        a, b = C.c_uint64(0), C.c_uint64(0)
        assert zl.zlhip_rt_stats(zl.libzl_hotpath_engine(), C.byref(a), C.byref(b)) == 0
        assert (a.value, b.value) == (1, 200), (a.value, b.value)   # one launch of the resident kernel for the whole session
        assert zl.libzl_hotpath_dropped_requests() == dropped0
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()


def test_a_bpm_that_would_divide_by_zero_is_ignored(built):
    """SyncTimer_setBpm / SyncTimer_startTimer are applied on the cycle's thread: a bpm of 0 (or one beyond 625 000 000, where a subbeat
    has no nanoseconds) would be an integer division by zero in the step clock -- SIGFPE on the audio thread.  Such a request
    is ignored: the session with the three bad calls renders exactly what the same session renders without them, and the transport
    still runs at 120 bpm."""
    from libzl_amd import libzl
    from libzl_amd._abi import Clock
    zl = libzl.load()
    L = np.random.default_rng(5).uniform(-1, 1, 5000).astype(np.float32)
    N = 128
    per = int(round(1e6 * N / 48000.0))

    def session(bad):
        zl.initJuce()
        assert zl.libzl_hotpath_status() == 0
        try:
            c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None, len(L), 48000.0, b"bpm0")
            zl.ClipAudioSource_playOnChannel(c, True, 0)
            out = np.zeros((14, 2, 12, N), dtype=np.float32)
            for k in range(14):
                if bad and k == 3:
                    zl.SyncTimer_setBpm(0)
                if bad and k == 5:
                    zl.SyncTimer_startTimer(0)
                if bad and k == 7:
                    zl.SyncTimer_setBpm(4000000000)
                assert zl.libzl_hotpath_cycle(N, 1000 + k * per, 1000 + (k + 1) * per, float(per), out[k, 0].ctypes.data, out[k, 1].ctypes.data) == 0
                tr = Clock()
                assert zl.libzl_hotpath_transport(C.byref(tr)) == 0 and tr.jack_subbeat_length_usecs == 5208      # still 120 bpm
            zl.ClipAudioSource_destroy(c)
            return out
        finally:
            zl.shutdownJuce()

    a, b = session(True), session(False)
    assert np.abs(b).max() > 0.1 and np.array_equal(a.view(np.int32), b.view(np.int32))


@pytest.mark.parametrize("case", ["narrow_128", "two_frame_tiles", "wide_resident", "launched", "launched_wide_512", "ragged_tiles_441"])
def test_cycles_delivered_straight_into_page_locked_buffers(built, case):
    """zlhip_render / zlhip_render_fanout into the caller's PAGE-LOCKED out_left / out_right / fan_out: the kernels write them directly (no
    staging rows, no host copy behind the cycle) at the caller's strides -- separate left and right planes, each [B][nframes].  Same bits
    as the oracle, through the resident kernel (narrow, tiled, wide) and the launched path; alternating with pageable buffers on the
    same engine (which go through the staging rows); the planes may sit anywhere relative to each other (right BEFORE left here)."""
    from libzl_amd.engine import pinned_array
    kw, N, mode, env, want_starts = CASES[case]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sc = random_scene(1300 + N, nframes=64, nblocks=1, events=False, mode=mode, **kw)
        syn, osyn = _engines(sc, max_frames=max(64, N), mode=mode)
        B = sc.num_buses
        planes = pinned_array(syn._lib, (2, B, N), np.float32)        # [0] = right, [1] = left: the channel stride is negative
        R, L = planes[0], planes[1]
        fan = pinned_array(syn._lib, (B, 6, N), np.float32)
        zoo = _zoo()
        t = 0
        for k in range(30):
            params = [zoo[(b + k // 4) % len(zoo)] for b in range(B)]
            clk, t = _clock(sc, t, N)
            bus, _ = osyn.render_batch(1, N, [clk])
            if k % 5 == 3:                                            # a cycle into pageable arrays in between: staged
                l2, r2, f2 = syn.process_fanout(N, clk, params)
                got = (l2, r2, f2)
            elif k % 5 == 1:                                          # no fan-out this cycle
                L[:] = 7.0; R[:] = 7.0
                syn.process_into(N, clk, L, R)
                got = (L, R, None)
            else:
                L[:] = 7.0; R[:] = 7.0; fan[:] = 7.0
                syn.process_into(N, clk, L, R, params, fan)
                got = (L, R, fan)
            assert np.array_equal(got[0].view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(got[1].view(np.int32), bus[:, 1].view(np.int32)), (case, k)
            if got[2] is not None:
                want = _oracle_fanout(bus, params)
                assert np.array_equal(got[2].view(np.int32), want.view(np.int32)), (case, k)
        peaks = syn.block_peaks()
        exp = np.abs(np.float32(131072.0) * bus).astype(np.int64).max(axis=2)
        assert np.array_equal(peaks[-1].astype(np.int64), exp), case
        starts, done = syn.rt_stats()
        syn.close()
        assert starts == want_starts and done == (30 if want_starts else 0), (case, starts, done)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
