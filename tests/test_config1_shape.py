"""BASELINE configs[0] at its STATED shape (SURVEY 8d cfg 1): one mono 44.1 kHz source of 176 400 frames, 256-frame blocks,
looping with lengthInBeats = 8 (clock-driven restart, synthetic clock k * 5805 us) and 7.5 (sample-space wrap), 1400 blocks =
two passes.  Fixture: tests/golden/c1_config1_shape.npz, made by the numpy restatement (tests/golden/make_golden.py --config1)."""
import numpy as np
import pytest

from golden_util import check_config1, load_config1
from scenario import oracle_trace, run_backend, run_oracle


def test_fixture_holds_both_wrap_branches():
    sc, ex = load_config1()
    assert (sc.nframes, sc.nblocks, sc.fs) == (256, 1400, 44100.0) and sc.sounds[0][0].shape[0] == 176400 and sc.sounds[0][1] is None
    # 8 beats: restarts decided by the JACK clock (not at the sample-space boundary 176400 = 689 * 256 + 16)
    assert ex["restarts"][0] == [(689, 6), (1378, 10)]
    # 7.5 beats = 165375 frames = 645 * 256 + 255: the sample-space wrap
    assert ex["restarts"][1] == [(645, 255), (1291, 254)]
    for v, rs in ex["restarts"].items():
        for k, _ in rs:
            assert k in ex["keep"]


def test_oracle_reproduces_config1(built):
    sc, ex = load_config1()
    bus, _, osyn = run_oracle(sc)
    tr, _ = oracle_trace(sc)
    check_config1(bus, tr, ex)
    for b in range(2):
        assert bool(osyn.voices[b].isPlaying) == bool(ex["state"][b, 0]) and osyn.voices[b].sourceSamplePosition == ex["state"][b, 1]


def test_kernel_code_on_host_reproduces_config1(built):
    from cpu_harness.sim import SimSynth
    sc, ex = load_config1()
    bus, rep, _, trace = run_backend(sc, SimSynth, batch=350, trace=True)
    check_config1(bus, trace, ex)
    for b in range(2):
        assert bool(rep[b].playing) == bool(ex["state"][b, 0]) and rep[b].source_sample_position == ex["state"][b, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1400, 256, 1])
def test_engine_reproduces_config1(built, batch):
    """The HIP engine through the C-ABI: one call of 1400 blocks, calls of 256, and 1400 real-time blocks."""
    from libzl_amd import SamplerSynth
    sc, ex = load_config1()
    bus, rep, syn, trace = run_backend(sc, SamplerSynth, batch=batch, trace=(batch != 1))
    check_config1(bus, trace, ex)
    for b in range(2):
        assert bool(rep[b].playing) == bool(ex["state"][b, 0]) and rep[b].source_sample_position == ex["state"][b, 1]
    syn.close()
