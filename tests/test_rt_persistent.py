"""GPU tier: zlhip_render through the resident real-time kernel (the default for narrow buses; ZL_RT_PERSISTENT=0 disables it) -- the same bits as the launched
path, block by block, across commands and parameter edits (applied by the resident kernel itself), idle spells (the kernel leaves and is
started again), block-size changes, batch calls in between (which stop the kernel) and device-wide waits of other engines."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from golden_util import golden_names, load_golden
from scenario import engine_cmd, random_scene, run_oracle, snapshot_clip

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rt_env():
    old = os.environ.get("ZL_RT_PERSISTENT")
    os.environ["ZL_RT_PERSISTENT"] = "1"
    yield
    if old is None:
        os.environ.pop("ZL_RT_PERSISTENT", None)
    else:
        os.environ["ZL_RT_PERSISTENT"] = old


def _play_blockwise(sc, *, pause_at=(), batch_at=(), edit_at=()):
    """The scene cycle by cycle through SamplerSynth.process (zlhip_render): returns (bus [B][2][K*N], last reports, synth)."""
    from libzl_amd import SamplerSynth
    from oracle import zl_oracle as zo
    ref = zo.OracleSynth(1, 1, sc.fs, sc.mode, max_sounds=max(8, len(sc.sounds)))
    syn = SamplerSynth(num_buses=sc.num_buses, voices_per_bus=sc.voices_per_bus, mode=sc.mode, playback_sample_rate=sc.fs,
                       max_frames=max(64, sc.nframes), max_batch_blocks=4, max_sounds=max(8, len(sc.sounds)),
                       sound_arena_bytes=max(1 << 20, sum((s[0].shape[0] + 16) * 8 for s in sc.sounds) + (1 << 16)))
    for i, (L, R, sr) in enumerate(sc.sounds):
        assert ref.register_clip(L, R, sr) == i and syn.register_clip(L, R, sr) == i
        if i in sc.clip_setup:
            sc.clip_setup[i](ref.lib, ref.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    N = sc.nframes
    out = np.zeros((sc.num_buses, 2, sc.nblocks * N), dtype=np.float32)
    for k in range(sc.nblocks):
        for ev in sc.events.get(k, []):
            if ev[0] == "cmd":
                syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
            elif ev[0] == "start":
                syn.start_voice(ev[1], ev[2], engine_cmd(**ev[3]), ev[4])
            elif ev[0] == "clip":
                ev[2](ref.lib, ref.clips[ev[1]])
                syn.set_clip_params(ev[1], snapshot_clip(ref.clips[ev[1]]))      # host-only: the resident kernel applies the edit at the next cycle
            elif ev[0] == "update":
                syn.update_voice(ev[1], ev[2], engine_cmd(**ev[3]))
            elif ev[0] == "stopv":
                syn.stop_voice(ev[1], ev[2], ev[3])
            elif ev[0] == "enable":
                syn.set_bus_enabled(ev[1], ev[2])
            else:
                raise AssertionError(ev[0])
        if k in pause_at:
            time.sleep(0.35)                                                    # longer than the kernel's idle timeout (200 ms)
        if k in edit_at:
            syn.set_clip_params(0, snapshot_clip(ref.clips[0]))                 # the same parameters again: an edit that changes nothing
        L, R = syn.process(N, sc.make_clocks(k, 1)[0])
        out[:, 0, k * N:(k + 1) * N] = L
        out[:, 1, k * N:(k + 1) * N] = R
    return out, syn.voice_reports(), syn


@pytest.mark.parametrize("name", golden_names())
def test_resident_kernel_renders_the_golden_vectors(built, rt_env, name):
    sc, ex = load_golden(name)
    bus, rep, syn = _play_blockwise(sc, pause_at=(3,), edit_at=(6,))
    assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32)), f"max diff {np.abs(bus - ex['bus']).max()}"
    for v in range(sc.num_buses * sc.voices_per_bus):
        assert bool(rep[v].playing) == bool(ex["state"][v, 0])
        if rep[v].playing:
            assert rep[v].source_sample_position == ex["state"][v, 1]
        assert rep[v].valid == int(ex["reports"][v, 0])
        if rep[v].valid:
            assert np.float32(rep[v].gain) == np.float32(ex["reports"][v, 1]) and np.float32(rep[v].progress) == np.float32(ex["reports"][v, 2])
    syn.close()


@pytest.mark.parametrize("seed,mode,nframes", [(300, 0, 256), (301, 4, 128), (302, 3, 64), (303, 0, 128),
                                               (304, 0, 32), (305, 4, 16), (306, 2, 48), (307, 0, 100), (308, 0, 240)])   # JACK periods that are no multiple of 64
def test_resident_kernel_matches_oracle_on_mixed_scenes(built, rt_env, seed, mode, nframes):
    """The reference's own shape -- 12 channels x 8 voices -- with commands and clip edits between cycles, and the levels."""
    from oracle import zl_oracle as zo
    lib = zo.load()
    sc = random_scene(seed, num_buses=12, voices_per_bus=8, nclips=20, mode=mode, nframes=nframes, nblocks=40)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn = _play_blockwise(sc, pause_at=(17,))
    assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32)), f"max diff {np.abs(bus - ref_bus).max()}"
    for v in range(96):
        assert bool(ref_syn.voices[v].isPlaying) == bool(rep[v].playing)
        assert ref_rep[v].valid == rep[v].valid
        if ref_rep[v].valid:
            assert ref_rep[v].gain == rep[v].gain and ref_rep[v].progress == rep[v].progress
    lv = syn.levels_tick(block_index=-1, with_hold_bus=1)                       # the meters read the resident kernel's last block
    N = nframes
    for b in range(12):
        ch = zo.LevelsChannel()
        L = np.ascontiguousarray(bus[b, 0, -N:]); R = np.ascontiguousarray(bus[b, 1, -N:])
        lib.zlo_levels_tick(C.byref(ch), L.ctypes.data, R.ctypes.data, N, 1 if b == 1 else 0)
        assert (lv[b].peak_a, lv[b].peak_b) == (ch.peakA, ch.peakB)
        assert lv[b].rms_a == lib.zlo_block_rms(L.ctypes.data, N, 0 if (mode & 2) else 1)
    syn.close()


@pytest.mark.parametrize("name", ["g5_adsr_commands", "g8_resampled_256"])
def test_launched_path_still_renders_the_golden_vectors(built, name):
    """ZL_RT_PERSISTENT=0: three launches and a completion event per block (the path wide buses always take)."""
    old = os.environ.get("ZL_RT_PERSISTENT")
    os.environ["ZL_RT_PERSISTENT"] = "0"
    try:
        sc, ex = load_golden(name)
        bus, rep, syn = _play_blockwise(sc)
        assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32))
        syn.close()
    finally:
        if old is None:
            os.environ.pop("ZL_RT_PERSISTENT", None)
        else:
            os.environ["ZL_RT_PERSISTENT"] = old


def test_resident_kernel_and_batches_interleave(built, rt_env):
    """Real-time cycles, then a batch on the same engine (the kernel is stopped), then cycles again: one continuous, exact stream."""
    from libzl_amd import SamplerSynth
    from oracle import zl_oracle as zo
    sc = random_scene(310, num_buses=4, voices_per_bus=8, nclips=10, nframes=128, nblocks=30, events=False)
    ref_bus, _, _ = run_oracle(sc)
    ref = zo.OracleSynth(1, 1, sc.fs, sc.mode, max_sounds=16)
    syn = SamplerSynth(num_buses=4, voices_per_bus=8, max_frames=128, max_batch_blocks=8, max_sounds=16, sound_arena_bytes=1 << 21)
    for i, (L, R, sr) in enumerate(sc.sounds):
        ref.register_clip(L, R, sr); syn.register_clip(L, R, sr)
        sc.clip_setup[i](ref.lib, ref.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    for ev in sc.events[0]:
        syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
    N = 128
    out = np.zeros((4, 2, 30 * N), dtype=np.float32)
    k = 0
    while k < 30:
        if k in (10, 21):                                                       # a batch of 8 blocks in the middle of the stream
            syn.render_batch(8, N, sc.make_clocks(k, 8))
            out[:, :, k * N:(k + 8) * N] = syn.read_bus()
            k += 8
            continue
        L, R = syn.process(N, sc.make_clocks(k, 1)[0])
        out[:, 0, k * N:(k + 1) * N] = L; out[:, 1, k * N:(k + 1) * N] = R
        k += 1
    assert np.array_equal(out.view(np.int32), ref_bus.view(np.int32))
    syn.close()


@pytest.fixture()
def rt_wide_env(rt_env):
    old = os.environ.get("ZL_RT_WIDE")
    os.environ["ZL_RT_WIDE"] = "1"
    yield
    if old is None:
        os.environ.pop("ZL_RT_WIDE", None)
    else:
        os.environ["ZL_RT_WIDE"] = old


@pytest.mark.parametrize("seed,mode,nframes,vpb,buses", [(320, 0, 256, 32, 3), (321, 4, 128, 64, 2), (322, 2, 64, 32, 2), (323, 0, 256, 128, 8)])
def test_resident_kernel_on_wide_buses(built, rt_wide_env, seed, mode, nframes, vpb, buses):
    """ZL_RT_WIDE=1 (opt-in: measured slower than launches).  Buses of 32 voices and more: one resident workgroup per voice (or few
    voices), the bus summed in voice order by the last workgroup of the bus to arrive.  Commands and clip edits between cycles, an
    idle spell, the levels; (323) is the bench shape."""
    from oracle import zl_oracle as zo
    lib = zo.load()
    V = vpb * buses
    sc = random_scene(seed, num_buses=buses, voices_per_bus=vpb, nclips=min(V, 160), mode=mode, nframes=nframes, nblocks=24 if V > 256 else 36)
    ref_bus, ref_rep, ref_syn = run_oracle(sc, threads=8)
    bus, rep, syn = _play_blockwise(sc, pause_at=(11,))
    assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32)), f"max diff {np.abs(bus - ref_bus).max()}"
    for v in range(V):
        assert bool(ref_syn.voices[v].isPlaying) == bool(rep[v].playing)
        assert ref_rep[v].valid == rep[v].valid
        if ref_rep[v].valid:
            assert ref_rep[v].gain == rep[v].gain and ref_rep[v].progress == rep[v].progress
    lv = syn.levels_tick(block_index=-1, with_hold_bus=1)
    N = nframes
    for b in range(buses):
        ch = zo.LevelsChannel()
        L = np.ascontiguousarray(bus[b, 0, -N:]); R = np.ascontiguousarray(bus[b, 1, -N:])
        lib.zlo_levels_tick(C.byref(ch), L.ctypes.data, R.ctypes.data, N, 1 if b == 1 else 0)
        assert (lv[b].peak_a, lv[b].peak_b) == (ch.peakA, ch.peakB)
        assert lv[b].rms_a == lib.zlo_block_rms(L.ctypes.data, N, 0 if (mode & 2) else 1)
    syn.close()


def test_wide_resident_equals_launched_with_a_command_storm(built, rt_wide_env):
    """The same wide engine through both real-time paths, every voice retriggered in one cycle (1024 operation ranges in one block)."""
    from libzl_amd import SamplerSynth, clip_command
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(77)
    src = [(rng.uniform(-1, 1, 3000 + 7 * i).astype(np.float32), rng.uniform(-1, 1, 3000 + 7 * i).astype(np.float32)) for i in range(16)]
    outs = []
    for resident in ("1", "0"):
        old = os.environ.get("ZL_RT_PERSISTENT")
        os.environ["ZL_RT_PERSISTENT"] = resident
        try:
            syn = SamplerSynth(num_buses=8, voices_per_bus=128, max_frames=256, max_batch_blocks=4, max_sounds=16, sound_arena_bytes=1 << 22)
            for L, R in src:
                syn.register_clip(L, R, 48000.0)
            rows = []
            for k in range(12):
                if k in (0, 5):                                                 # 1024 starts in one cycle
                    for v in range(1024):
                        syn.start_voice(v // 128, v % 128, clip_command(clip=(v + k) % 16, midi_note=55 + v % 11, midi_channel=v // 128 - 2, start_playback=1,
                                                                        looping=1, change_volume=1, volume=0.3 + 0.001 * (v % 97)), 0)
                if k == 8:
                    for v in range(0, 1024, 3):
                        syn.stop_voice(v // 128, v % 128, True)
                L, R = syn.process(256, synthetic_clocks(1, 256, 48000.0, start_block=k)[0])
                rows.append(np.stack([L, R], axis=1).copy())
            outs.append((np.concatenate(rows, axis=2), [(r.playing, r.valid, r.gain, r.progress) for r in syn.voice_reports()]))
            syn.close()
        finally:
            if old is None:
                os.environ.pop("ZL_RT_PERSISTENT", None)
            else:
                os.environ["ZL_RT_PERSISTENT"] = old
    assert np.array_equal(outs[0][0].view(np.int32), outs[1][0].view(np.int32)) and np.abs(outs[0][0]).max() > 1.0
    assert outs[0][1] == outs[1][1]


def test_parameter_edits_and_commands_keep_the_kernel_resident(built, rt_env):
    """zlhip_clip_set records the edit on the host and the resident kernel applies it at the next cycle boundary
    (SamplerSynthVoice.cpp:189-196 reads the parameters per block): a mixed scene with clip edits and commands between cycles
    is rendered by ONE launch of the kernel, bit-exact."""
    sc = None
    for seed in range(331, 360):              # a seed whose scene holds clip edits as well as commands between its cycles
        cand = random_scene(seed, num_buses=12, voices_per_bus=8, nclips=20, mode=0, nframes=128, nblocks=60)
        kinds = {ev[0] for k, evs in cand.events.items() if k > 0 for ev in evs}
        if {"clip", "cmd"} <= kinds:
            sc = cand
            break
    assert sc is not None
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn = _play_blockwise(sc, edit_at=(5, 6, 7, 30))
    assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32)), f"max diff {np.abs(bus - ref_bus).max()}"
    starts, cycles = syn.rt_stats()
    assert (starts, cycles) == (1, sc.nblocks)
    syn.close()


def test_an_idle_engine_sees_every_cycles_clock(built, rt_env):
    """one beat-locked voice, no commands for hundreds of cycles, a clock (and SyncTimer playhead) that changes every
    cycle: the resident kernel reads the mailbox behind a system-scope acquire, so a quiet cycle cannot render with the previous
    cycle's cached clock.  Against the launched path and the oracle."""
    from edge_scenes import _base
    from scenario import play_cmd
    sc = _base(55, nframes=64, nblocks=500, nsounds=1, length=20000)
    sc.num_buses, sc.voices_per_bus = 12, 8
    sc.bpm = 200; sc.moving_playhead = True; sc.block0 = 7000

    def setup(lib, clip):
        clip.lengthInBeats = 1.0
        clip.lengthInSeconds = float(np.float32(0.2))
    sc.clip_setup[0] = setup
    sc.events[0] = [("cmd", play_cmd(0, midi_channel=3, note=60, volume=0.9), sc.tick_at(0))]
    ref_bus, _, _ = run_oracle(sc)
    bus, rep, syn = _play_blockwise(sc)
    assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32))
    assert syn.rt_stats() == (1, 500)
    syn.close()
    os.environ["ZL_RT_PERSISTENT"] = "0"
    bus2, _, syn2 = _play_blockwise(sc)
    assert np.array_equal(bus2.view(np.int32), ref_bus.view(np.int32)) and syn2.rt_stats() == (0, 0)
    syn2.close()


def test_two_engines_device_wide_waits_do_not_wait_for_a_resident_kernel(built, rt_env):
    """engine A renders real-time cycles through its resident kernel while the same process creates, grows and destroys
    engine B (hipFree / hipHostFree / hipDeviceSynchronize inside: device-wide waits).  Those calls ask A's kernel to leave for
    their duration instead of waiting out its idle timeout (set to 2 s here), A renders the cycles in between with launches or a
    fresh residency, and its stream stays bit-exact."""
    import torch
    from libzl_amd import SamplerSynth
    sc = random_scene(340, num_buses=12, voices_per_bus=8, nclips=16, nframes=128, nblocks=40, events=False)
    ref_bus, _, _ = run_oracle(sc)
    from oracle import zl_oracle as zo
    ref = zo.OracleSynth(1, 1, sc.fs, sc.mode, max_sounds=32)
    A = SamplerSynth(num_buses=12, voices_per_bus=8, max_frames=128, max_batch_blocks=4, max_sounds=32, sound_arena_bytes=1 << 22, rt_idle_timeout_us=2_000_000)
    for i, (L, R, sr) in enumerate(sc.sounds):
        ref.register_clip(L, R, sr); A.register_clip(L, R, sr)
        sc.clip_setup[i](ref.lib, ref.clips[i])
        A.set_clip_params(i, snapshot_clip(ref.clips[i]))
    for ev in sc.events[0]:
        A.handle_clip_command(engine_cmd(**ev[1]), ev[2])
    N = 128
    out = np.zeros((12, 2, 40 * N), dtype=np.float32)
    t_foreign = 0.0
    for k in range(40):
        L, R = A.process(N, sc.make_clocks(k, 1)[0])
        out[:, 0, k * N:(k + 1) * N] = L; out[:, 1, k * N:(k + 1) * N] = R
        if k in (5, 15, 25):
            t0 = time.perf_counter()
            Bsyn = SamplerSynth(num_buses=2, voices_per_bus=8, max_frames=128, max_batch_blocks=4, max_sounds=8, sound_arena_bytes=1 << 20)
            src = torch.rand(4000, device="cuda")
            cid = Bsyn.register_clip_device(src.data_ptr(), None, 4000, 48000.0)      # hipDeviceSynchronize inside
            Bsyn.enable_trace(True)
            Bsyn.render_batch(2, N, sc.make_clocks(0, 2))                              # the trace buffer is allocated, then grown (hipFree)
            Bsyn.render_batch(4, N, sc.make_clocks(0, 4))
            Bsyn.synchronize()
            Bsyn.close()                                                               # hipFree / hipHostFree of everything
            t_foreign += time.perf_counter() - t0
    assert np.array_equal(out.view(np.int32), ref_bus.view(np.int32))
    starts, cycles = A.rt_stats()
    assert starts >= 2 and cycles >= 30, (starts, cycles)         # it was asked to leave and came back
    assert t_foreign < 3 * 1.5, t_foreign                         # three rounds; each would wait >= 2 s (often several times) behind A's kernel
    A.close()


def test_kernel_leaving_while_a_cycle_is_posted(built, rt_env):
    """The race at the idle timeout: with a timeout of 150 us and cycles 0-400 us apart, the resident kernel decides to leave again and
    again at the very moment the host posts a cycle -- sometimes it has seen the post (and renders it before it goes), sometimes not
    (the host finds it gone, starts it again, and the new kernel takes the cycle from the mailbox).  2500 cycles with commands and clip
    edits in between: every block the oracle's, none lost, none rendered twice; the kernel was started hundreds of times."""
    from libzl_amd import SamplerSynth
    from oracle import zl_oracle as zo
    sc = random_scene(411, num_buses=4, voices_per_bus=8, nclips=12, nframes=64, nblocks=2500)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    ref = zo.OracleSynth(1, 1, sc.fs, sc.mode, max_sounds=16)
    syn = SamplerSynth(num_buses=4, voices_per_bus=8, max_frames=64, max_batch_blocks=4, max_sounds=16, sound_arena_bytes=1 << 21, rt_idle_timeout_us=150)
    for i, (L, R, sr) in enumerate(sc.sounds):
        assert ref.register_clip(L, R, sr) == i and syn.register_clip(L, R, sr) == i
        if i in sc.clip_setup:
            sc.clip_setup[i](ref.lib, ref.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    rng = np.random.default_rng(5)
    gaps = rng.uniform(0.0, 400e-6, sc.nblocks)
    N = sc.nframes
    out = np.zeros((sc.num_buses, 2, sc.nblocks * N), dtype=np.float32)
    for k in range(sc.nblocks):
        for ev in sc.events.get(k, []):
            if ev[0] == "cmd":
                syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
            elif ev[0] == "clip":
                ev[2](ref.lib, ref.clips[ev[1]])
                syn.set_clip_params(ev[1], snapshot_clip(ref.clips[ev[1]]))
            elif ev[0] == "enable":
                syn.set_bus_enabled(ev[1], ev[2])
            else:
                raise AssertionError(ev[0])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < gaps[k]:                                  # (a spin: sleep() is far coarser than the timeout)
            pass
        L, R = syn.process(N, sc.make_clocks(k, 1)[0])
        out[:, 0, k * N:(k + 1) * N] = L
        out[:, 1, k * N:(k + 1) * N] = R
    starts, cycles = syn.rt_stats()
    rep = syn.voice_reports()
    syn.close()
    assert np.array_equal(out.view(np.int32), ref_bus.view(np.int32)), int(np.argmax(np.abs(out - ref_bus).max(axis=(0, 1)) > 0)) // N
    for v in range(32):
        assert bool(ref_syn.voices[v].isPlaying) == bool(rep[v].playing)
    assert cycles == sc.nblocks and starts > 100, (starts, cycles)


def test_more_voice_operations_in_a_cycle_than_the_mapped_buffers_hold(built, rt_env):
    """96 voices, and a cycle that patches every one of them three times (288 operations; the engine's host-mapped operation buffers
    start at voices + 32 entries): the buffers grow while the resident kernel is up -- a device-synchronising step, so the kernel is
    asked to yield first and is started again -- and the cycle is still the oracle's, as are the cycles around it."""
    from scenario import Scene, play_cmd, rand_source
    rng = np.random.default_rng(612)
    sc = Scene(num_buses=12, voices_per_bus=8, fs=48000.0, nframes=128, nblocks=16)
    for i in range(12):
        L, R = rand_source(rng, 3000 + 100 * i, stereo=bool(i % 3))
        sc.sounds.append((L, R, 48000.0))
        sc.clip_setup[i] = (lambda lib, clip, i=i: (setattr(clip, "lengthInBeats", 0.37), setattr(clip, "lengthInSeconds", float(np.float32(0.03 + 0.002 * i)))))
    notes = [52, 55, 58, 60, 62, 65, 67, 70]
    sc.events[0] = [("cmd", play_cmd(b, midi_channel=b - 2, loop=True, note=n, volume=0.2 + 0.05 * j), 0) for b in range(12) for j, n in enumerate(notes)]
    sc.events[5] = [("cmd", dict(clip=b, midiChannel=b - 2, midiNote=n, changeVolume=1, volume=v), 0)
                    for v in (0.9, 0.1, 0.45) for b in range(12) for n in notes]
    sc.events[9] = [("cmd", dict(clip=b, midiChannel=b - 2, midiNote=n, stopPlayback=1), 0) for b in range(12) for n in notes[::2]]
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn = _play_blockwise(sc)
    starts, cycles = syn.rt_stats()
    syn.close()
    assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32))
    for v in range(96):
        assert bool(ref_syn.voices[v].isPlaying) == bool(rep[v].playing)
    assert cycles == 16 and starts >= 2                            # (the kernel was restarted by the growth)


def test_the_period_changes_between_cycles(built, rt_env):
    """JACK renegotiates its buffer size: cycles of 256, 100, 32, 17 ... frames on ONE engine, in any order.  The resident kernel is started
    again whenever the period changes (its workgroup size is the period rounded up to whole waves); every cycle is the oracle's."""
    from libzl_amd import SamplerSynth
    from libzl_amd._abi import Clock
    from oracle import zl_oracle as zo
    sc = random_scene(431, num_buses=4, voices_per_bus=8, nclips=12, nframes=64, nblocks=1, events=False)
    ref = zo.OracleSynth(1, 1, sc.fs, sc.mode, max_sounds=16)
    osyn = zo.OracleSynth(4, 8, sc.fs, sc.mode, max_sounds=16)
    syn = SamplerSynth(num_buses=4, voices_per_bus=8, max_frames=256, max_batch_blocks=4, max_sounds=16, sound_arena_bytes=1 << 21)
    for i, (L, R, sr) in enumerate(sc.sounds):
        assert ref.register_clip(L, R, sr) == i and syn.register_clip(L, R, sr) == i and osyn.register_clip(L, R, sr) == i
        sc.clip_setup[i](ref.lib, ref.clips[i]); sc.clip_setup[i](osyn.lib, osyn.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    from scenario import oracle_cmd
    for ev in sc.events[0]:
        syn.handle_clip_command(engine_cmd(**ev[1]), ev[2]); osyn.handle_clip_command(oracle_cmd(**ev[1]), ev[2])
    periods = [256, 100, 100, 32, 17, 256, 64, 1, 200, 32, 32, 255, 128, 100, 256, 48] * 3
    t = 0
    for k, N in enumerate(periods):
        per = int(round(1e6 * N / sc.fs))
        clk = Clock(); clk.current_usecs = t; clk.next_usecs = t + per; clk.jack_playhead = 0; clk.jack_playhead_usecs = 0
        clk.jack_subbeat_length_usecs = ((60000000000) // (sc.bpm * 96)) // 1000
        t += per
        L, R = syn.process(N, clk)
        bus, _ = osyn.render_batch(1, N, [clk])
        assert np.array_equal(L.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(R.view(np.int32), bus[:, 1].view(np.int32)), (k, N)
    starts, cycles = syn.rt_stats()
    syn.close()
    assert cycles == len(periods) and starts >= 30                 # one start per change of period


def test_wide_buses_with_one_voice_per_workgroup_are_resident_by_default(built, rt_env):
    """No ZL_RT_WIDE: an engine whose voices each get a resident workgroup of their own (4 buses x 32 voices here) renders its cycles through
    the resident kernel -- same median as the launched path, a far shorter tail --; one that would need several voices per workgroup
    (8 x 128) keeps the launched path.  Exact either way."""
    old = os.environ.pop("ZL_RT_WIDE", None)
    try:
        for buses, vpb, resident in ((4, 32, True), (8, 128, False)):
            sc = random_scene(440 + vpb, num_buses=buses, voices_per_bus=vpb, nclips=min(buses * vpb, 120), nframes=128, nblocks=14)
            ref_bus, ref_rep, ref_syn = run_oracle(sc, threads=8)
            bus, rep, syn = _play_blockwise(sc)
            starts, cycles = syn.rt_stats()
            syn.close()
            assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32)), (buses, vpb)
            assert (cycles == sc.nblocks and starts >= 1) if resident else (cycles == 0 and starts == 0), (buses, vpb, starts, cycles)
    finally:
        if old is not None:
            os.environ["ZL_RT_WIDE"] = old
