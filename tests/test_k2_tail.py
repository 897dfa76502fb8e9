"""K2's split tail (zl_launch_render / zl_k2_body): the last blocks of a narrow-bus window are rendered by several workgroups per block, a few
buses each.  Parity against the oracle for every way the buses divide, at the lowered threshold of the test tier (conftest) and at the shipped one
(windows of 2048 blocks and more), with events, fan-out and the offline bounce crossing the seam between the two parts of the launch."""
import numpy as np
import pytest

from scenario import compare_runs, random_scene, run_backend, run_oracle


@pytest.fixture(scope="module")
def Engine(built):
    from libzl_amd import SamplerSynth
    return SamplerSynth


@pytest.mark.gpu
@pytest.mark.parametrize("buses,width", [(8, 8), (12, 8), (10, 8), (4, 8), (2, 16), (6, 16), (16, 8), (7, 8)])
@pytest.mark.parametrize("mode", [0, 3])
def test_split_tail_every_division_of_the_buses(Engine, buses, width, mode):
    """8 / 12 / 16 buses: four workgroups per tail block; 10 / 6 / 2: two (one bus each for 2 x 16); 4 x 8: four of one bus; 7: no split"""
    sc = random_scene(0x7A11 + buses * 31 + width + mode, nframes=256, nblocks=131, nclips=10, min_len=300, max_len=30000, events=True, mode=mode,
                      num_buses=buses, voices_per_bus=width)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1 << 30)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, buses * width)
    syn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nframes", [256, 192, 100])
def test_split_tail_block_lengths(Engine, nframes):
    sc = random_scene(0x7A12 + nframes, nframes=nframes, nblocks=97, nclips=8, min_len=1500, max_len=9000, events=True, mode=0, num_buses=8, voices_per_bus=8)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1 << 30, pipelined=True)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 64)
    syn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("events,buses", [(False, 8), (True, 12)])
def test_split_tail_at_the_shipped_threshold(Engine, events, buses):
    """2100 blocks: steady voices are one window (2100 blocks, the last 525 split); with events the windows are 2048 + 52 blocks"""
    sc = random_scene(0x7A13 + buses, nframes=256, nblocks=2100, nclips=9, min_len=20000, max_len=60000, events=events, mode=0, num_buses=buses, voices_per_bus=8)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1 << 30)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, buses * 8)
    syn.close()


@pytest.mark.gpu
def test_split_tail_with_fanout_and_bounce(Engine):
    from libzl_amd import PassthroughParams
    sc = random_scene(0x7A14, nframes=256, nblocks=120, nclips=8, min_len=1500, max_len=20000, events=True, mode=0, num_buses=8, voices_per_bus=8)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    fan = [PassthroughParams(0.8, 1.0, -1.25, -0.3 + 0.1 * b, 0) for b in range(8)]
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1 << 30, fanout=fan)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 64)
    from test_rt_fanout import _oracle_fanout
    assert np.array_equal(syn.fan_result.view(np.int32), _oracle_fanout(ref_bus, fan).view(np.int32))
    syn.close()
    bus, rep, syn, _ = run_backend(sc, Engine, bounce=("f32", 50))
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 64)
    syn.close()


@pytest.mark.gpu
def test_very_long_call_stays_inside_the_grid_limit(Engine):
    """66000 blocks of a 16-voice engine in ONE call: its plan windows would be 65536 blocks long (few voices: long windows) -- one launch slot per
    block plus the split tail's extra workgroups must stay below the 65536 slots of a grid's y dimension (windows are capped at 60000 blocks)"""
    sc = random_scene(0x7A20, nframes=256, nblocks=66000, nclips=4, min_len=20000, max_len=60000, events=False, mode=0, num_buses=2, voices_per_bus=8)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, batch=1 << 30)
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 16)
    syn.close()
