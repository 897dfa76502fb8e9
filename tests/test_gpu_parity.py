"""GPU tier (MI355X): the HIP engine, called through the C-ABI of include/zlhip.h, against the CPU oracle and the
golden vectors.  Integer / index work and fp32 audio are compared BIT-EXACTLY (the kernels are built with
-ffp-contract=off and follow the oracle's operation order); the tolerance north_star allows (1e-6 abs per sample)
is only used where the summation order differs on purpose (mix groups vs the strictly sequential reference)."""
import ctypes as C

import numpy as np
import pytest

from golden_util import golden_names, load_golden
from scenario import (Scene, compare_runs, oracle_trace, play_cmd, rand_source, random_scene, run_backend, run_oracle,
                      stop_cmd)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the engine has no CPU path")
    from libzl_amd import SamplerSynth
    return SamplerSynth


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("batch", [1, 4, 1 << 30])
def test_golden_vectors(Engine, name, batch):
    sc, ex = load_golden(name)
    bus, rep, syn, trace = run_backend(sc, Engine, batch=batch, trace=True)
    assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32)), f"max diff {np.abs(bus - ex['bus']).max()}"
    assert np.array_equal(trace, ex["trace"]), "per-frame source index differs"
    for v in range(sc.num_buses * sc.voices_per_bus):
        assert bool(rep[v].playing) == bool(ex["state"][v, 0])
        if rep[v].playing:
            assert rep[v].source_sample_position == ex["state"][v, 1]
        assert rep[v].valid == int(ex["reports"][v, 0])
        if rep[v].valid:
            assert np.float32(rep[v].gain) == np.float32(ex["reports"][v, 1]) and np.float32(rep[v].progress) == np.float32(ex["reports"][v, 2])
    syn.close()


@pytest.mark.parametrize("name", golden_names())
def test_golden_vectors_pipelined_path(Engine, name):
    """Without the trace (which the debug store adds to the gather loop) and with the per-frame control path forced."""
    sc, ex = load_golden(name)
    bus, _, syn, _ = run_backend(sc, Engine, batch=6)
    assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32))
    syn.close()
    bus, _, syn, _ = run_backend(sc, Engine, batch=6, force_slow=True)
    assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32))
    syn.close()


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("mode", [0, 3, 4])
def test_mixed_scenes_bit_exact(Engine, seed, mode):
    sc = random_scene(100 + seed, mode=mode, nframes=[64, 128, 256][seed % 3], nblocks=20)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=[1, 3, 7, 1 << 30][seed % 4])
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()


@pytest.mark.parametrize("seed", range(3))
def test_source_indices_and_batch_split(Engine, seed):
    sc = random_scene(200 + seed, nframes=128, nblocks=16, events=False)
    a, _, s1, ta = run_backend(sc, Engine, batch=1, trace=True)
    b, _, s2, tb = run_backend(sc, Engine, batch=16, trace=True)
    assert np.array_equal(a.view(np.int32), b.view(np.int32)) and np.array_equal(ta, tb)
    tr, _ = oracle_trace(sc)
    assert np.array_equal(ta, tr)          # bit-exact loop-index / wrap arithmetic
    s1.close(); s2.close()


@pytest.mark.parametrize("group", [1, 2, 4])
def test_mix_groups(Engine, group):
    sc = random_scene(300 + group, mix_group=group, num_buses=2, voices_per_bus=8, nclips=14, nblocks=10)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=5)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 16)                  # same two-level order: bit-exact
    sc.mix_group = 0
    seq_bus, _, _ = run_oracle(sc)
    tol = 1e-6                                                             # north_star: 1e-6 fp32 per sample
    assert np.abs(seq_bus - bus).max() <= tol * max(1.0, float(np.abs(seq_bus).max()))
    syn.close()


@pytest.mark.parametrize("window", [1, 3, 8])
def test_plan_windows_overlap_planning_and_rendering(Engine, window):
    """One call that spans several plan windows (K1 of window i+1 runs on the planning stream while K2 of window i
    renders from the other buffer set) gives the same bits as the oracle, including events, tails and reports."""
    sc = random_scene(1000 + window, nframes=128, nblocks=26, events=False)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, trace = run_backend(sc, Engine, batch=26, trace=True, plan_window_blocks=window)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    tr, _ = oracle_trace(sc)
    assert np.array_equal(trace, tr)
    syn.close()
    bus2, rep2, syn2, _ = run_backend(sc, Engine, batch=26, plan_window_blocks=window)      # untraced (pipelined) path
    compare_runs(ref_bus, ref_rep, ref_syn, bus2, rep2, sc.num_buses * sc.voices_per_bus)
    syn2.close()


@pytest.mark.parametrize("window", [0, 512])
def test_segment_table_overflow(Engine, window):
    """Short pitched loops over a long batch: thousands of segments per voice; a full segment table makes K1
    simulate the rest of the window (one window of 1500 blocks; windows of 256/512/512...)."""
    sc = random_scene(510, nclips=8, min_len=700, max_len=1500, nblocks=1500, nframes=64, events=False)
    for ev in sc.events[0]:
        ev[1]["looping"] = 1
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1500, plan_window_blocks=window, no_periodic=True)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1500, plan_window_blocks=window)          # periodic loops: one pass planned per window
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()


@pytest.mark.parametrize("nframes", [64, 128, 192, 256, 512, 1024, 4096,        # 4096 = the largest block the engine accepts
                                     16, 32, 48, 100, 200, 300, 441, 480, 1000, 1])  # ... and any other JACK period: a block runs on whole waves
@pytest.mark.parametrize("batch", [1, 5, 1 << 30])
def test_block_sizes(Engine, nframes, batch):
    """Every block size: 64 / 128 frames render 4 / 2 blocks per workgroup in batches, 192 and 256 one, 512 and
    1024 use several frame tiles per block and the separate level scan (K3); single blocks keep their own workgroup.  A length that
    is no multiple of 64 leaves lanes behind the block's end in its last wave: they recompute the last frame and store nothing."""
    sc = random_scene(5000 + nframes, nframes=nframes, nblocks=11, events=True, min_len=900, max_len=9000)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=batch)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    peaks = syn.block_peaks()                                        # integer peaks of the last call's blocks
    k_last = peaks.shape[0]
    tail = ref_bus[:, :, ref_bus.shape[2] - k_last * nframes:].reshape(sc.num_buses, 2, k_last, nframes)
    want = np.abs(np.float32(131072.0) * tail).astype(np.int64).max(axis=3).transpose(2, 0, 1)
    assert np.array_equal(peaks.astype(np.int64), want)
    syn.close()


@pytest.mark.parametrize("seed,events", [(4100, False), (4101, True), (4102, False)])
def test_back_to_back_calls_pipeline(Engine, seed, events):
    """Consecutive zlhip_render_batch calls queued without host synchronisation in between (call i+1 is planned while
    call i renders; per-call clocks / reports / statistics are double buffered) give the oracle's audio and reports.
    With events, the commands between calls synchronise (they need the voice table) -- also the oracle's result."""
    sc = random_scene(seed, nframes=128, nblocks=40, events=events)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=8, pipelined=True, plan_window_blocks=3)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.set_profiling(True)
    syn.close()


def test_single_window_calls_pipeline(Engine):
    """Calls of one plan window each, queued back to back: the first is planned on the caller's stream, the following
    ones on the planning stream while their predecessor renders (the voice state is handed over between the streams)."""
    sc = random_scene(4200, nframes=128, nblocks=2400, nclips=10, min_len=3000, max_len=40000, events=False)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=600, pipelined=True)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()


def test_profile_totals_sum_over_pipelined_calls(Engine):
    from libzl_amd.engine import synthetic_clocks
    sc = random_scene(4103, nframes=128, nblocks=8, events=False)
    _, _, syn, _ = run_backend(sc, Engine, batch=8, plan_window_blocks=3)
    syn.set_profiling(True)
    syn.profile_totals(reset=True)
    for i in range(5):
        syn.render_batch(8, 128, synthetic_clocks(8, 128, sc.fs, start_block=8 * (i + 1)))
    tot, ncalls = syn.profile_totals(reset=True)
    assert ncalls == 5 and tot.render_launches == 15 and tot.render_ms > 0.0 and tot.total_ms >= tot.render_ms
    assert syn.profile_totals()[1] == 0
    syn.close()


def test_realtime_process_equals_batch(Engine):
    """zlhip_render (one JACK cycle, host buffers) gives the same bits as the batched path."""
    from libzl_amd.engine import synthetic_clocks
    sc = random_scene(700, nframes=256, nblocks=6, events=False)
    ref_bus, _, _ = run_oracle(sc)
    from scenario import snapshot_clip, engine_cmd
    from oracle import zl_oracle as zo
    ref = zo.OracleSynth(1, 1, sc.fs, 0, max_sounds=16)
    syn = Engine(sc.num_buses, sc.voices_per_bus, max_frames=256, max_batch_blocks=4, max_sounds=16, playback_sample_rate=sc.fs)
    for i, (L, R, sr) in enumerate(sc.sounds):
        ref.register_clip(L, R, sr); syn.register_clip(L, R, sr)
        sc.clip_setup[i](ref.lib, ref.clips[i]); syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    for ev in sc.events[0]:
        syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
    for k in range(sc.nblocks):
        L, R = syn.process(256, sc.make_clocks(k, 1)[0])
        assert np.array_equal(L.view(np.int32), ref_bus[:, 0, k * 256:(k + 1) * 256].view(np.int32))
        assert np.array_equal(R.view(np.int32), ref_bus[:, 1, k * 256:(k + 1) * 256].view(np.int32))
    syn.close()


def _big_scene(V=1024, B=8, nframes=256, nblocks=24, loop=3000, seed=0x5A19, hermite=False, ratios=False):
    """BASELINE-sized voice count on short sources so the oracle finishes in seconds."""
    rng = np.random.default_rng(seed)
    sc = Scene(num_buses=B, voices_per_bus=V // B, fs=48000.0, nframes=nframes, nblocks=nblocks, mode=4 if hermite else 0)
    ev = []
    for v in range(V):
        L, R = rand_source(rng, loop + int(rng.integers(0, 500)), stereo=True)
        sc.sounds.append((L, R, 48000.0))
        vol, pan = float(np.float32(rng.uniform(0.25, 1.0))), float(np.float32(rng.uniform(-1, 1)))
        ln = float(np.float32((loop - 40 - (v % 17)) / 48000.0))

        def setup(lib, clip, vol=vol, pan=pan, ln=ln):
            clip.lengthInBeats = 3.5
            clip.lengthInSeconds = ln
            clip.volumeAbsolute = vol
            clip.pan = pan
        sc.clip_setup[v] = setup
        note = int(rng.integers(48, 73)) if ratios else 60
        ev.append(("start", v // (V // B), v % (V // B), play_cmd(v, midi_channel=v // (V // B) - 2, note=note,
                                                                    volume=float(np.float32(rng.uniform(0.1, 1.0)))), 0))
    sc.events[0] = ev
    return sc


def test_full_voice_count_bit_exact_against_oracle(Engine):
    """1024 stereo voices x 256-frame blocks (the BASELINE metric's shape), whole buses summed in voice order."""
    sc = _big_scene()
    ref_bus, ref_rep, ref_syn = run_oracle(sc, threads=8)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=24)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 1024)
    peaks = syn.block_peaks()
    exp = np.stack([np.abs(np.float32(131072.0) * bus[:, c].reshape(8, 24, 256)).astype(np.int64).max(axis=2) for c in (0, 1)], axis=-1)
    assert np.array_equal(peaks, exp.transpose(1, 0, 2))                   # AudioLevels integer peaks, every block
    # RMS extension at the full voice count: exact (defined summation order)
    from oracle import zl_oracle as zo
    lib = zo.load()
    for k in (0, 11, 23):
        lv = syn.levels_tick(block_index=k)
        for b in range(8):
            for c, got in ((0, lv[b].rms_a), (1, lv[b].rms_b)):
                row = np.ascontiguousarray(bus[b, c, k * 256:(k + 1) * 256])
                assert got == lib.zlo_block_rms(row.ctypes.data, 256, 1), (k, b, c)
    syn.close()


def test_baseline_configs_1_64_voices_resampled(Engine):
    """BASELINE configs[1]: 64 stereo voices, linear-interp resample (44.1 kHz and 48 kHz sources, notes +-12),
    48 kHz playback, 256-frame blocks; attack/decay envelopes as SURVEY.md section 8d cfg 2."""
    rng = np.random.default_rng(0x5A17 + 1)
    sc = Scene(num_buses=8, voices_per_bus=8, fs=48000.0, nframes=256, nblocks=40)
    ev = []
    for v in range(64):
        sr = [44100.0, 48000.0][v % 2]
        L, R = rand_source(rng, int(rng.integers(int(0.05 * sr), int(0.12 * sr))), stereo=True)
        sc.sounds.append((L, R, sr))
        vol, pan, A, S = float(rng.uniform(0.25, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(0, 0.05)), float(rng.uniform(0.5, 1))
        beats = float(rng.uniform(0.04, 0.09))

        def setup(lib, clip, vol=vol, pan=pan, A=A, S=S, beats=beats):
            lib.zlo_clip_set_length(clip, C.c_float(beats), 120)
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(vol))
            lib.zlo_clip_set_pan(clip, C.c_float(pan))
            clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = (A, 0.1, S, 0.05)
        sc.clip_setup[v] = setup
        ev.append(("cmd", play_cmd(v, midi_channel=v // 8 - 2, note=60 + int(rng.integers(-12, 13)), volume=float(np.float32(rng.uniform(0.1, 1)))), 0))
    sc.events[0] = ev
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=40)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 64)
    syn.close()


def test_baseline_configs_2_1024_loops_128_frames_levels(Engine):
    """BASELINE configs[2]: 1024 stereo clip loops, per-clip gain/pan, AudioLevels peaks every block, 128-frame blocks."""
    sc = _big_scene(V=1024, B=16, nframes=128, nblocks=30, seed=0x5A17 + 2)
    ref_bus, ref_rep, ref_syn = run_oracle(sc, threads=8)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=30)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 1024)
    peaks = syn.block_peaks()
    exp = np.stack([np.abs(np.float32(131072.0) * bus[:, c].reshape(16, 30, 128)).astype(np.int64).max(axis=2) for c in (0, 1)], axis=-1)
    assert np.array_equal(peaks, exp.transpose(1, 0, 2))
    from oracle import zl_oracle as zo                                      # "AudioLevels RMS/peak": the RMS extension, exact
    lib = zo.load()
    for k in (0, 17, 29):
        lv = syn.levels_tick(block_index=k)
        for b in range(16):
            for c, got in ((0, lv[b].rms_a), (1, lv[b].rms_b)):
                row = np.ascontiguousarray(bus[b, c, k * 128:(k + 1) * 128])
                assert got == lib.zlo_block_rms(row.ctypes.data, 128, 1), (k, b, c)
    syn.close()


def test_baseline_configs_4_96k_batched_bounce(Engine):
    """BASELINE configs[4] in miniature: 96 kHz sources and playback, many voices, long batched render that spans
    several plan windows; throughput-only in the benchmark, parity here."""
    rng = np.random.default_rng(0x5A17 + 5)
    V, B = 512, 4
    sc = Scene(num_buses=B, voices_per_bus=V // B, fs=96000.0, nframes=256, nblocks=48)
    ev = []
    for v in range(V):
        L, R = rand_source(rng, 5000 + int(rng.integers(0, 900)), stereo=True)
        sc.sounds.append((L, R, 96000.0))

        def setup(lib, clip, v=v):
            clip.lengthInBeats = 2.5
            clip.lengthInSeconds = float(np.float32((4800 - v % 23) / 96000.0))
        sc.clip_setup[v] = setup
        ev.append(("start", v // (V // B), v % (V // B), play_cmd(v, midi_channel=v // (V // B) - 2, note=60, volume=0.5), 0))
    sc.events[0] = ev
    ref_bus, ref_rep, ref_syn = run_oracle(sc, threads=8)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=48, plan_window_blocks=16)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, V)
    syn.close()
    # the bounce proper (zlhip_bounce): the same blocks delivered to host memory in sub-batches of 10 (the last one ragged) while the
    # next sub-batch renders; as floats, and in the recorder's 16-bit format converted on the GPU
    bus, rep, syn, _ = run_backend(sc, Engine, bounce=("f32", 10), plan_window_blocks=4)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, V)
    syn.close()
    pcm, rep, syn, _ = run_backend(sc, Engine, bounce=("pcm16", 10))
    from oracle import zl_oracle as zo
    lib = zo.load()
    for b in range(B):
        want = np.empty((sc.nblocks * sc.nframes, 2), dtype=np.int16)
        Lr, Rr = np.ascontiguousarray(ref_bus[b, 0]), np.ascontiguousarray(ref_bus[b, 1])
        lib.zlo_pcm16_stereo(Lr.ctypes.data, Rr.ctypes.data, len(Lr), want.ctypes.data)
        assert np.array_equal(pcm[b], want)
    assert len(np.unique(pcm)) > 1000                              # (a real signal, not a constant)
    syn.close()


def test_baseline_configs_3_pitched_hermite_shard(Engine):
    """BASELINE.json configs[3], per-GPU shape: 1024 voices on one bus, pitch 0.5-2x, 4-tap Hermite (build-defined extension)."""
    sc = _big_scene(V=1024, B=1, nblocks=8, hermite=True, ratios=True, seed=0x5A1B)
    sc.mix_group = 64
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=8)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 1024)
    syn.close()


def test_size_independent_properties_at_full_size(Engine):
    """Linearity in the command volume (second tap only, Q1) and determinism across batch splits at 1024 voices."""
    sc = _big_scene(nblocks=32, seed=0x5A1C)
    a, _, s1, _ = run_backend(sc, Engine, batch=32)
    b, _, s2, _ = run_backend(sc, Engine, batch=5)
    assert np.array_equal(a.view(np.int32), b.view(np.int32))
    # Q2: first frame of every block is silent, in every bus
    assert not a.reshape(8, 2, 32, 256)[:, :, :, 0].any()
    s1.close(); s2.close()


def test_command_batches_equal_single_commands(Engine):
    """zlhip_handle_commands (a block's worth of commands in one call, one K0 update) == the same commands one by one."""
    from scenario import engine_cmd
    sc = random_scene(6100, num_buses=4, voices_per_bus=8, nclips=12, nframes=128, nblocks=12, events=True)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)

    class Batched(Engine):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self._pending, self._tick = [], 0

        def handle_clip_command(self, cmd, current_tick=0):
            if self._pending and current_tick != self._tick:
                self._flush()
            self._pending.append(cmd); self._tick = current_tick
            return 1

        def _flush(self):
            if self._pending:
                self.handle_clip_commands(self._pending, self._tick)
                self._pending = []

        def set_clip_params(self, clip, params):
            self._flush()
            return super().set_clip_params(clip, params)

        def render_batch(self, *a, **k):
            self._flush()
            return super().render_batch(*a, **k)

    bus, rep, syn, _ = run_backend(sc, Batched, batch=4)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()


def test_voice_stealing_and_slot_reuse(Engine):
    """One-shots end on the device; their slots must become allocatable again exactly as in the oracle."""
    sc = random_scene(800, num_buses=1, voices_per_bus=3, nclips=4, nblocks=30, nframes=128, events=False, min_len=900, max_len=1400)
    for ev in sc.events[0]:
        ev[1].update(midiChannel=-2, looping=0); ev[1].pop("stopPlayback", None)
    for k in (6, 12, 18, 24):
        sc.events[k] = [("cmd", play_cmd(k % 4, midi_channel=-2, loop=False, note=58 + k % 5, volume=0.7), k)]
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=4)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 3)
    syn.close()


def test_levels_tick_matches_oracle(Engine):
    from oracle import zl_oracle as zo
    sc = random_scene(900, num_buses=4, nblocks=6, nframes=128, events=False)
    bus, _, syn, _ = run_backend(sc, Engine, batch=6)
    lib = zo.load()
    chans = [zo.LevelsChannel() for _ in range(4)]
    for k in (2, 5, -2, 5):
        lv = syn.levels_tick(block_index=k, with_hold_bus=1)
        for b in range(4):
            if k == -2:
                lib.zlo_levels_tick(C.byref(chans[b]), None, None, 0, 1 if b == 1 else 0)
            else:
                L = np.ascontiguousarray(bus[b, 0, k * 128:(k + 1) * 128]); R = np.ascontiguousarray(bus[b, 1, k * 128:(k + 1) * 128])
                lib.zlo_levels_tick(C.byref(chans[b]), L.ctypes.data, R.ctypes.data, 128, 1 if b == 1 else 0)
            assert (lv[b].peak_a, lv[b].peak_b) == (chans[b].peakA, chans[b].peakB)
            assert lv[b].peak_db_a == chans[b].peakDbA and lv[b].peak_db_b == chans[b].peakDbB and lv[b].combined_db == chans[b].combinedDb
            if b == 1:
                assert lv[b].peak_a_hold_signal == chans[b].peakAHoldSignal and lv[b].hold_db_b == chans[b].holdDbB
            if k != -2:
                # build-defined RMS extension: the summation order is part of its definition (oracle zlo_block_sumsq): exact
                assert lv[b].rms_a == lib.zlo_block_rms(L.ctypes.data, 128, 1) and lv[b].rms_b == lib.zlo_block_rms(R.ctypes.data, 128, 1)
    syn.close()


@pytest.mark.parametrize("shape", [dict(nframes=128), dict(nframes=256), dict(nframes=64), dict(nframes=512),
                                   dict(nframes=256, mode=2), dict(nframes=128, mode=3), dict(nframes=256, mix_group=4),
                                   dict(nframes=256, batch=1), dict(nframes=1024, mode=2),
                                   dict(nframes=32), dict(nframes=100, mode=2), dict(nframes=441), dict(nframes=200, mix_group=4), dict(nframes=48, batch=1)])
def test_rms_extension_is_bit_exact_in_every_kernel_shape(Engine, shape):
    """The sums of squares come from K2's fused scan (N <= 256), from K3's wave-per-block scan (N > 256, single blocks)
    and from K3 behind the mix-group sum: all three follow the order the oracle defines, in the modes with and without
    the one-frame delay (tile offset 1 / 0)."""
    from oracle import zl_oracle as zo
    lib = zo.load()
    shape = dict(shape)
    batch = shape.pop("batch", 5)
    N = shape["nframes"]
    sc = random_scene(910 + N, num_buses=4, nblocks=5, events=False, **shape)
    bus, _, syn, _ = run_backend(sc, Engine, batch=batch)
    off = 0 if (sc.mode & 2) else 1
    assert np.abs(bus).max() > 0
    for k in ([-1] if batch == 1 else [0, 3, 4]):
        lv = syn.levels_tick(block_index=k, with_hold_bus=-1)
        kk = 4 if batch == 1 else k
        for b in range(4):
            L = np.ascontiguousarray(bus[b, 0, kk * N:(kk + 1) * N]); R = np.ascontiguousarray(bus[b, 1, kk * N:(kk + 1) * N])
            assert lv[b].rms_a == lib.zlo_block_rms(L.ctypes.data, N, off), (k, b)
            assert lv[b].rms_b == lib.zlo_block_rms(R.ctypes.data, N, off), (k, b)
    syn.close()


@pytest.mark.parametrize("n", [1000, 1001, 64])      # 16-byte accesses (frames % 4 == 0) and the scalar form
def test_passthrough_matches_oracle(Engine, n):
    import torch
    from oracle import zl_oracle as zo
    from libzl_amd import PassthroughParams
    lib = zo.load()
    B = 3
    x = torch.rand((B, 2, n), device="cuda") * 2 - 1
    out = torch.full((B, 6, n), 7.0, device="cuda")
    syn = Engine(B, 2, max_frames=64, max_batch_blocks=1, max_sounds=4)
    params = [PassthroughParams(1.0, 0.0, 0.5, 0.0, 0), PassthroughParams(0.8, 1.0, 1.0, -0.3, 0), PassthroughParams(1.0, 1.0, 1.0, 0.0, 1)]
    syn.passthrough(params, x.data_ptr(), out.data_ptr(), n)
    syn.synchronize(); torch.cuda.synchronize()
    xh, oh = x.cpu().numpy(), out.cpu().numpy()
    for b in range(B):
        outs = [np.zeros(n, dtype=np.float32) for _ in range(6)]
        arr = (C.c_void_p * 6)(*[o.ctypes.data for o in outs])
        p = zo.Passthrough(params[b].dry_amount, params[b].wet_fx1_amount, params[b].wet_fx2_amount, params[b].pan_amount, params[b].muted)
        L = np.ascontiguousarray(xh[b, 0]); R = np.ascontiguousarray(xh[b, 1])
        lib.zlo_passthrough_process(C.byref(p), L.ctypes.data, R.ctypes.data, arr, n)
        for c in range(6):
            assert np.array_equal(oh[b, c].view(np.int32), outs[c].view(np.int32)), (b, c)
    syn.close()


def _oracle_fanout(bus, params):
    """JackPassthrough (oracle) applied to every bus of [B][2][frames]."""
    from oracle import zl_oracle as zo
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


FANOUT_CASES = {
    # name: (scene kwargs, batch, pipelined)
    "several_buses_per_workgroup": (dict(num_buses=5, voices_per_bus=8, nframes=128, nblocks=12), 1 << 30, False),
    "four_blocks_per_workgroup":   (dict(num_buses=3, voices_per_bus=16, nframes=64, nblocks=22), 1 << 30, False),
    "odd_bus_width":               (dict(num_buses=4, voices_per_bus=12, nframes=256, nblocks=9), 4, True),
    "two_frame_tiles":             (dict(num_buses=3, voices_per_bus=8, nframes=512, nblocks=6), 1 << 30, False),
    "mix_groups":                  (dict(num_buses=3, voices_per_bus=8, nframes=128, nblocks=10, mix_group=4), 1 << 30, False),
    "delay_fixed":                 (dict(num_buses=3, voices_per_bus=8, nframes=128, nblocks=10, mode=2), 1 << 30, False),
    "single_blocks":               (dict(num_buses=4, voices_per_bus=8, nframes=128, nblocks=5), 1, False),
    "period_of_100_frames":        (dict(num_buses=5, voices_per_bus=8, nframes=100, nblocks=11), 1 << 30, False),   # lanes behind the block's end store no fan-out
    "period_of_300_frames_groups": (dict(num_buses=3, voices_per_bus=8, nframes=300, nblocks=5, mix_group=4), 1 << 30, False),
    "period_of_33_frames_single":  (dict(num_buses=2, voices_per_bus=8, nframes=33, nblocks=7), 1, False),
}


@pytest.mark.parametrize("case", sorted(FANOUT_CASES))
def test_fused_fanout_is_the_passthrough_of_the_bus(Engine, case):
    """zlhip_render_batch_fanout: bus identical to the plain call (the oracle's), fan-out identical to the oracle's
    JackPassthrough applied to that bus -- every fast path, negative amounts, pan beyond +-1, in every K2 / K3 shape."""
    from libzl_amd import PassthroughParams
    kw, batch, pipelined = FANOUT_CASES[case]
    sc = random_scene(4242, **kw)
    zoo = [PassthroughParams(1.0, 0.0, 0.5, 0.0, 0), PassthroughParams(0.8, 1.0, -1.25, -0.3, 0), PassthroughParams(1.0, 1.0, 1.0, 0.0, 1),
           PassthroughParams(-0.5, 2.0, 0.0, 1.5, 0), PassthroughParams(1.0, 1.0, 1.0, 0.0, 0)]
    params = [zoo[b % len(zoo)] for b in range(sc.num_buses)]
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=batch, pipelined=pipelined, fanout=params)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    want = _oracle_fanout(ref_bus, params)
    got = syn.fan_result
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), np.argwhere(got.view(np.int32) != want.view(np.int32))[:4].tolist()
    syn.close()


@pytest.mark.parametrize("batch", [1, 1 << 30])
def test_one_voice_per_task_keeps_the_reference_order(Engine, batch):
    """voices_per_task = 1: every voice is rendered by its own workgroup and K3 adds the voices in voice order -- the
    reference's summation order exactly (0 + v0 + v1 + ...), with the parallelism a wide bus lacks in real time.  Same bits
    as the oracle's default (whole-bus) order."""
    sc = random_scene(7300, num_buses=2, voices_per_bus=24, nclips=30, nframes=128, nblocks=14, events=True)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)                       # mix_group = 0: sequential over the whole bus
    sc.mix_group = 1
    bus, rep, syn, _ = run_backend(sc, Engine, batch=batch)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()


@pytest.mark.parametrize("nframes", [128, 512])
def test_real_time_blocks_of_wide_buses_are_split_per_voice(Engine, nframes):
    """Single-block calls on buses of 32 voices and more: the engine renders one voice per workgroup and adds them in
    voice order (pick_group) -- no configuration, same bits as the whole-bus walk of the oracle; zlhip_render too."""
    sc = random_scene(7400, num_buses=2, voices_per_bus=40, nclips=60, nframes=nframes, nblocks=10, events=True)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    peaks = syn.block_peaks()                                      # levels of the last block come from K3 on this path
    exp = np.abs(np.float32(131072.0) * ref_bus[:, :, -nframes:]).astype(np.int64).max(axis=2)
    assert np.array_equal(peaks[-1].astype(np.int64), exp)
    syn.close()


def test_errors_are_reported(Engine):
    from libzl_amd import ZlHipError
    from libzl_amd.engine import synthetic_clocks
    syn = Engine(2, 4, max_frames=128, max_batch_blocks=2, max_sounds=2, sound_arena_bytes=1 << 16, sound_arena_max_bytes=1 << 16)   # a fixed arena
    with pytest.raises(ZlHipError):
        syn.render_batch(3, 128, synthetic_clocks(3, 128, 48000.0))        # more blocks than max_batch_blocks
    with pytest.raises(ZlHipError):
        syn.render_batch(1, 192, synthetic_clocks(1, 192, 48000.0))        # nframes beyond max_frames
    with pytest.raises(ZlHipError):
        syn.register_clip(np.zeros(1 << 20, dtype=np.float32), None, 48000.0)   # arena full
    syn.close()


@pytest.mark.parametrize("window", [7, 64])
def test_moving_playhead_across_plan_windows(Engine, window):
    """beat-locked loops against a MOVING SyncTimer playhead (golden g4b: a timer that has been running for 12 000
    cycles, start ticks that are not 0, a nextLoopTick behind the playhead -- the u64 wrap of SamplerSynthVoice.cpp:180-181,236-237 -- and
    the edge scenes) rendered in ONE batch cut into several plan windows: every window's planner reads the playhead of its own blocks'
    clocks (the restart block is found by bisection over clocks that each carry a different playhead).  Traced and pipelined."""
    from golden_util import load_golden
    from edge_scenes import SCENES
    sc, ex = load_golden("g4b_beat_locked_moving_playhead")
    bus, rep, syn, trace = run_backend(sc, Engine, batch=1 << 30, trace=True, plan_window_blocks=window)
    assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32)), f"max diff {np.abs(bus - ex['bus']).max()}"
    assert np.array_equal(trace, ex["trace"])
    syn.close()
    bus, rep, syn, _ = run_backend(sc, Engine, batch=90, pipelined=True, plan_window_blocks=window)
    assert np.array_equal(bus.view(np.int32), ex["bus"].view(np.int32))
    syn.close()
    for name in ("beat_locked_moving_playhead", "beat_locked_moving_playhead_long"):
        es = SCENES[name]()
        ref_bus, ref_rep, ref_syn = run_oracle(es)
        bus, rep, syn, _ = run_backend(es, Engine, batch=1 << 30, plan_window_blocks=window * 8)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, es.num_buses * es.voices_per_bus)
        syn.close()


@pytest.mark.parametrize("buses,vpb", [(1, 1), (5, 3), (3, 5), (2, 7), (7, 12), (2, 20), (1, 9), (13, 2), (3, 33)])
@pytest.mark.parametrize("batch", [1, 1 << 30])
def test_engines_of_any_shape(Engine, buses, vpb, batch):
    """Bus widths that are no multiple of the kernel's chunk of 8 voices, a single bus, a single voice, 13 buses of 2: the shapes the
    narrow-bus packing (whole buses of a multiple of 8 voices per workgroup) and the wide-bus split do not take."""
    sc = random_scene(7600 + 10 * buses + vpb, num_buses=buses, voices_per_bus=vpb, nclips=max(3, buses * vpb // 2), nframes=128, nblocks=12, events=True)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=batch)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, buses * vpb)
    syn.close()
