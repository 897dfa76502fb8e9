"""Unusual parameter combinations (pitch ratios of 1/16 and 16, loops of a few frames, loops longer than the file,
lengthInBeats = -1 (Q10), slices, degenerate envelopes, mono next to stereo, resampled sources): the oracle defines the
behaviour, the engine's code must match it bit for bit -- on the host harness here, on the GPU in the gpu tier."""
import pytest

from edge_scenes import SCENES
from scenario import compare_runs, run_backend, run_oracle


@pytest.fixture(scope="module")
def Sim(built):
    from cpu_harness.sim import SimSynth
    return SimSynth


@pytest.fixture(scope="module")
def Engine(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the engine has no CPU path")
    from libzl_amd import SamplerSynth
    return SamplerSynth


@pytest.mark.parametrize("mode", [0, 4])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_edge_scene_on_the_host_harness(Sim, name, mode):
    sc = SCENES[name]()
    sc.mode = mode
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    for batch in (1, 1 << 30):
        bus, rep, syn, _ = run_backend(sc, Sim, batch=batch)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 3, 4])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_edge_scene_on_the_gpu(Engine, name, mode):
    sc = SCENES[name]()
    sc.mode = mode
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    for batch in (1, 5, 1 << 30):
        bus, rep, syn, _ = run_backend(sc, Engine, batch=batch)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
        syn.close()
