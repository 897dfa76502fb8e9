"""CPU tier: the engine's planning / per-frame code (zl_plan.h, zl_render.h, zl_host.h) executed on the host by
tests/cpu_harness versus the oracle on seeded mixed scenes -- bit-exact audio, positions, voice state and reports."""
import numpy as np
import pytest

from scenario import compare_runs, oracle_trace, random_scene, run_backend, run_oracle


@pytest.fixture(scope="module")
def Sim(built):
    from cpu_harness.sim import SimSynth
    return SimSynth


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("mode", [0, 3, 4])
def test_mixed_scenes_bit_exact(Sim, seed, mode):
    sc = random_scene(100 + seed, mode=mode, nframes=[64, 128, 256][seed % 3], nblocks=20)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Sim, batch=[1, 3, 7, 1 << 30][seed % 4])
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)


@pytest.mark.parametrize("seed", range(4))
def test_batch_split_invariance_and_trace(Sim, seed):
    """The same scene rendered block by block and in batches gives identical bits; per-frame source indices match."""
    sc = random_scene(200 + seed, nframes=128, nblocks=16, events=False)
    a, _, _, ta = run_backend(sc, Sim, batch=1, trace=True)
    b, _, _, tb = run_backend(sc, Sim, batch=16, trace=True)
    assert np.array_equal(a.view(np.int32), b.view(np.int32)) and np.array_equal(ta, tb)
    tr, _ = oracle_trace(sc)
    assert np.array_equal(ta, tr)


@pytest.mark.parametrize("group", [1, 2, 4])
def test_mix_group_order_matches_oracle_grouped_order(Sim, group):
    """voices_per_task < voices_per_bus: two-level summation order, reproduced by the oracle's mix_group."""
    sc = random_scene(300 + group, mix_group=group, num_buses=2, voices_per_bus=8, nclips=14, nblocks=10)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, _, _ = run_backend(sc, Sim, batch=5)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 16)
    # and within 1e-6 (relative to the mix magnitude) of the reference's strictly sequential order
    sc.mix_group = 0
    seq_bus, _, _ = run_oracle(sc)
    assert np.abs(seq_bus - bus).max() <= 1e-6 * max(1.0, float(np.abs(seq_bus).max()))


def test_forced_per_frame_control_equals_planned(Sim):
    sc = random_scene(400, nframes=128, nblocks=12)
    a, ra, _, _ = run_backend(sc, Sim, batch=4)
    b, rb, _, _ = run_backend(sc, Sim, batch=4, force_slow=True)
    assert np.array_equal(a.view(np.int32), b.view(np.int32))
    for v in range(sc.num_buses * sc.voices_per_bus):
        assert (ra[v].playing, ra[v].source_sample_position, ra[v].gain) == (rb[v].playing, rb[v].source_sample_position, rb[v].gain)


def test_steady_state_uses_runs_not_per_block_plans(Sim):
    """Long loops at ratio 1: after the first block the planner records O(1) runs per voice and no slow blocks."""
    sc = random_scene(500, nclips=6, min_len=40000, max_len=50000, nblocks=64, nframes=256, events=False)
    for i in list(sc.clip_setup):
        def setup(lib, clip):
            clip.lengthInBeats = 0.5
            clip.lengthInSeconds = float(np.float32(0.8))
        sc.clip_setup[i] = setup
    for ev in sc.events[0]:
        ev[1]["looping"] = 1                      # every voice loops (one-shots would end in a release tail)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Sim, batch=64)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    assert syn.slow_blocks() == 0
    playing = [v for v in range(syn.num_voices) if rep[v].playing]
    assert playing and all(1 <= syn.l.zlsim_num_runs(syn.s, v) <= 6 for v in playing)      # inline runs


def _short_pitched_loops():
    sc = random_scene(510, nclips=8, min_len=700, max_len=1500, nblocks=1500, nframes=64, events=False)
    for ev in sc.events[0]:
        ev[1]["looping"] = 1
    return sc


def test_segment_table_overflow_falls_back_to_simulation(Sim):
    """Short pitched loops over one long window, planned pass by pass (test hook): thousands of linear segments per
    voice.  When a voice's segment table (ZL_MAXTSEG) is full the rest of its window is simulated per frame -- same
    audio as the oracle."""
    sc = _short_pitched_loops()
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Sim, batch=1500, no_periodic=True)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    assert syn.slow_blocks() > 100                                # the fallback was exercised


def test_periodic_loops_are_planned_once_per_window(Sim):
    """A sample-space loop in sustain repeats exactly: K1 plans one pass and K1c replays it (same scene as above:
    no table overflow, a few dozen segments per voice instead of thousands, same audio)."""
    sc = _short_pitched_loops()
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Sim, batch=1500)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    playing = [v for v in range(syn.num_voices) if rep[v].playing]
    periodic = [v for v in playing if syn.l.zlsim_periodic_segments(syn.s, v) > 0]
    assert len(periodic) >= len(playing) // 2                     # beat-locked loops are clock-driven, not periodic
    assert all(syn.l.zlsim_num_tsegs(syn.s, v) < 200 for v in periodic)
    # the same in several windows / calls: the window-end state of a periodic voice is the exact recurrence value
    bus2, rep2, syn2, _ = run_backend(sc, Sim, batch=97)
    compare_runs(ref_bus, ref_rep, ref_syn, bus2, rep2, sc.num_buses * sc.voices_per_bus)


def test_steady_loops_replay_their_recorded_pass(Sim):
    """From the second window on, a looping voice in sustain is not planned at all: K1 finds it on the pass recorded for
    it (same position, bit for bit) and hands K1c that pass shifted back by the voice's offset into it (per_t0 < 0).  Same
    audio as the oracle for every split of the scene into calls; parameter changes fall back to the planner."""
    sc = _short_pitched_loops()
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    for batch in (97, 300, 13):
        bus, rep, syn, _ = run_backend(sc, Sim, batch=batch)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
        playing = [v for v in range(syn.num_voices) if rep[v].playing]
        replayed = [v for v in playing if syn.l.zlsim_periodic_segments(syn.s, v) > 0 and syn.l.zlsim_period_start(syn.s, v) < 0]
        assert len(replayed) >= len(playing) // 2, (batch, len(replayed), len(playing))     # (beat-locked loops are clock-driven)
        assert all(syn.l.zlsim_num_tsegs(syn.s, v) <= 64 for v in replayed)
    # with events (volume changes, restarts, stops, clip edits between blocks): still the oracle's audio
    sc2 = random_scene(511, nclips=8, min_len=700, max_len=1500, nblocks=600, nframes=64, events=True)
    for ev in sc2.events[0]:
        ev[1]["looping"] = 1
    ref2 = run_oracle(sc2)
    for batch in (50, 1 << 30):
        bus, rep, _, _ = run_backend(sc2, Sim, batch=batch)
        compare_runs(ref2[0], ref2[1], ref2[2], bus, rep, sc2.num_buses * sc2.voices_per_bus)


def test_clip_edits_drop_the_recorded_pass(Sim):
    """edge_scenes.loop_edits_while_playing in windows of 40 blocks: passes are recorded and replayed, and every clip edit
    (loop length, start, beat-locking, a volume command) sends the voice back to the planner -- the oracle's audio."""
    from edge_scenes import loop_edits_while_playing
    sc = loop_edits_while_playing()
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    hits = []
    orig = Sim.render_batch

    def counting(self, *a, **k):
        orig(self, *a, **k)
        hits.append(sum(1 for v in range(self.num_voices)
                        if self.l.zlsim_periodic_segments(self.s, v) > 0 and self.l.zlsim_period_start(self.s, v) < 0))
    Sim.render_batch = counting
    try:
        bus, rep, syn, _ = run_backend(sc, Sim, batch=40)
    finally:
        Sim.render_batch = orig
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    assert sum(hits) >= 2 * len(hits) and min(hits[1:]) < 4 <= max(hits)      # replayed most of the time, not right after an edit


def test_unit_step_loops_over_many_passes_are_planned_once(Sim):
    """Playback at the source rate from an integer start: a pass is ONE exact linear run.  A few passes per window are
    described by inline runs; a window with more passes than the run list holds is finished by the periodic descriptor
    after two passes (K1 stays O(1) per window however long the window is)."""
    from edge_scenes import unit_step_loops_many_passes
    sc = unit_step_loops_many_passes()
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Sim, batch=1500)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 8)
    playing = [v for v in range(syn.num_voices) if rep[v].playing]
    assert len(playing) == 6
    assert all(1 <= syn.l.zlsim_periodic_segments(syn.s, v) <= 2 for v in playing)
    assert all(syn.l.zlsim_num_tsegs(syn.s, v) < 12 for v in playing)
    bus2, rep2, _, _ = run_backend(sc, Sim, batch=211)            # several windows: fewer passes each, inline runs + descriptor
    compare_runs(ref_bus, ref_rep, ref_syn, bus2, rep2, 8)


def test_no_free_voice_drops_command_like_reference(Sim):
    """More starts than voices on one channel: the surplus commands are dropped (SamplerSynth.cpp:204-215)."""
    sc = random_scene(600, num_buses=1, voices_per_bus=2, nclips=5, nblocks=6, events=False)
    for ev in sc.events[0]:
        ev[1]["midiChannel"] = -2
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, _, _ = run_backend(sc, Sim, batch=3)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 2)


@pytest.mark.parametrize("nframes", [1, 16, 32, 48, 100, 300, 441, 480, 1000])
@pytest.mark.parametrize("batch", [1, 5, 1 << 30])
def test_block_lengths_that_are_no_multiple_of_64(Sim, nframes, batch):
    """JACK periods of 16 or 32 frames, 441, 480 ...: the planner, the assembler and the per-frame code take any block length (the
    engine's kernels then run a block on whole 64-lane waves and mask the lanes behind its end: tests/test_gpu_parity.py)."""
    sc = random_scene(5000 + nframes, nframes=nframes, nblocks=11, events=True, min_len=900, max_len=9000)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Sim, batch=batch)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
    syn.close()
