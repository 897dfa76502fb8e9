"""GPU tier: BASELINE configs[3] and configs[4] at their PER-GPU size (SURVEY 8d cfg 4-5), checked the way bench.py checks its
own output -- a few random (bus, block) rows re-rendered by the oracle from the positions the engine reported -- plus the
size-independent properties of the domain: a bounce cut into sub-batches equals the single call, frame 0 of every block of a bus
is silent (quirk Q2), the integer peaks equal those computed from the bus, the 16-bit delivery equals the oracle's conversion."""
import os
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the engine has no CPU path")
    return torch


def _restart_all(syn, P, V, vpb):
    from libzl_amd import clip_command
    for v in range(V):
        syn.stop_voice(v // vpb, v % vpb, allow_tail_off=False)
    for v in range(V):
        bus, slot = divmod(v, vpb)
        assert syn.start_voice(bus, slot, clip_command(clip=v, midi_note=P["note"][v], midi_channel=bus - 2, start_playback=1, looping=1,
                                                       change_volume=1, volume=P["velocity"][v]), 0) == 1


def test_configs_4_per_gpu_share_4096_voices_3750_blocks_96k(torch_cuda):
    """BASELINE configs[4], one GPU's share: 4096 voices (32 buses x 128) x 10 s @ 96 kHz = 3750 blocks of 256 frames, distinct
    2 s loops per voice (6.3 GB of sources), through zlhip_render_batch (one call) AND zlhip_bounce (16-bit delivery, six
    sub-batches)."""
    import bench
    from libzl_amd import SamplerSynth
    from libzl_amd.engine import synthetic_clocks
    from oracle import zl_oracle as zo
    torch = torch_cuda
    V, B, N, K, fs = 4096, 32, 256, 3750, 96000.0
    vpb = V // B
    loop = int(2.0 * fs)
    rng = np.random.default_rng(0x5A17 + 5)
    picks = [(int(rng.integers(0, B)), int(rng.integers(0, K))) for _ in range(3)]
    syn = SamplerSynth(B, vpb, max_frames=N, max_batch_blocks=K, max_sounds=V, mode=0, playback_sample_rate=fs,
                       sound_arena_bytes=(loop + 16) * 8 * V + (1 << 20))
    P, kept = bench.build_scene(syn, torch, torch.device("cuda", 0), vpb, B, fs, loop, 0x5A17 + 5, keep_buses={b for b, _ in picks})
    # (one block of pre-roll: the engine reports positions after a rendered block; the oracle is then set to them, as bench.py does)
    syn.render_batch(1, N, synthetic_clocks(1, N, fs))
    syn.synchronize()
    clocks = synthetic_clocks(K, N, fs, start_block=1)
    before = syn.voice_reports()
    out = torch.empty((B, 2, K * N), device="cuda", dtype=torch.float32)
    syn.render_batch(K, N, clocks, bus_out_dev=out.data_ptr())
    syn.synchronize()
    rows = {(b, k): out[b, :, k * N:(k + 1) * N].cpu().numpy() for b, k in picks}
    args = types.SimpleNamespace(frames=N, fs=fs)
    res = bench.spot_check(syn, rows, before, P, kept, picks, args, clocks, vpb, loop, fs, 0)
    assert all(r["bit_exact"] for r in res) and all(r["peak"] > 1.0 for r in res), res
    # quirk Q2: frame 0 of every block of a bus is the constant 0; AudioLevels' integer peaks are those of the bus
    blocks = out.view(B, 2, K, N)
    assert bool((blocks[..., 0] == 0).all())
    want_peaks = (blocks * 131072.0).abs().to(torch.int32).amax(dim=3).permute(2, 0, 1).cpu().numpy()      # [K][B][2]
    assert np.array_equal(syn.block_peaks(), want_peaks)
    # the same 10 s again as a bounce to host memory in the recorder's 16-bit format: six sub-batches of 625 blocks, equal to the
    # oracle's conversion (zlo_pcm16_stereo) of the single call's floats -- batch-split determinism and the delivery in one
    _restart_all(syn, P, V, vpb)
    syn.render_batch(1, N, synthetic_clocks(1, N, fs))
    pcm = syn.bounce(K, N, clocks, fmt="pcm16")
    host = out.cpu().numpy()
    lib = zo.load()
    want = np.empty((K * N, 2), dtype=np.int16)
    for b in range(B):
        Lr, Rr = np.ascontiguousarray(host[b, 0]), np.ascontiguousarray(host[b, 1])
        lib.zlo_pcm16_stereo(Lr.ctypes.data, Rr.ctypes.data, K * N, want.ctypes.data)
        assert np.array_equal(pcm[b], want), b
    syn.close()


def test_configs_3_per_gpu_share_1024_pitched_hermite_voices_8192_blocks(torch_cuda):
    """BASELINE configs[3], one GPU's share: 1024 voices of ONE stereo bus, pitch ratio 0.5-2x, 4-tap Hermite, 8192 blocks of 256
    frames (43.7 s).  One call of 8192 blocks == 8128 + 64 blocks (determinism across batch splits, on the device); rows of the
    first blocks against the oracle from the start, rows of the last 64 blocks against the oracle from the positions the engine
    reported after block 8127 (every voice is a loop in sustain: the position is its whole state)."""
    import bench
    from libzl_amd import SamplerSynth
    from libzl_amd.engine import synthetic_clocks
    torch = torch_cuda
    V, B, N, K, fs = 1024, 1, 256, 8192, 48000.0
    tail = 64
    loop = int(2.0 * fs)
    syn = SamplerSynth(B, V, max_frames=N, max_batch_blocks=K, max_sounds=V, mode=4, playback_sample_rate=fs,
                       sound_arena_bytes=(loop + 16) * 8 * V + (1 << 20))
    P, kept = bench.build_scene(syn, torch, torch.device("cuda", 0), V, B, fs, loop, 0x5A17 + 4, notes=(48, 72), keep_buses={0})
    assert min(P["note"]) == 48 and max(P["note"]) == 72
    syn.render_batch(1, N, synthetic_clocks(1, N, fs))          # one block of pre-roll (see above)
    syn.synchronize()
    clocks = synthetic_clocks(K, N, fs, start_block=1)
    args = types.SimpleNamespace(frames=N, fs=fs)
    before = syn.voice_reports()
    one = torch.empty((B, 2, K * N), device="cuda", dtype=torch.float32)
    syn.render_batch(K, N, clocks, bus_out_dev=one.data_ptr())
    syn.synchronize()
    head = [(0, 0), (0, 5), (0, 31)]
    rows = {(b, k): one[b, :, k * N:(k + 1) * N].cpu().numpy() for b, k in head}
    res = bench.spot_check(syn, rows, before, P, kept, head, args, clocks, V, loop, fs, 4)
    assert all(r["bit_exact"] for r in res) and all(r["peak"] > 3.0 for r in res), res
    # again, split 8128 + 64
    _restart_all(syn, P, V, V)
    syn.render_batch(1, N, synthetic_clocks(1, N, fs))
    two = torch.empty((B, 2, K * N), device="cuda", dtype=torch.float32)
    syn.render_batch(K - tail, N, clocks, bus_out_dev=two.data_ptr())
    # (the second call's bus stride is its own number of blocks: render it into a buffer of its own, compare piecewise)
    syn.synchronize()
    mid = syn.voice_reports()
    last = torch.empty((B, 2, tail * N), device="cuda", dtype=torch.float32)
    tail_clocks = synthetic_clocks(tail, N, fs, start_block=1 + K - tail)
    syn.render_batch(tail, N, tail_clocks, bus_out_dev=last.data_ptr())
    syn.synchronize()
    # layout of a call: [B][2][nblocks * N] with nblocks = the call's own block count
    a = one.view(B, 2, K, N)
    b1 = two.flatten()[: B * 2 * (K - tail) * N].view(B, 2, K - tail, N)
    assert torch.equal(a[:, :, :K - tail], b1)
    assert torch.equal(a[:, :, K - tail:], last.view(B, 2, tail, N))
    picks = [(0, 3), (0, 40), (0, 63)]
    rows = {(b, k): last[b, :, k * N:(k + 1) * N].cpu().numpy() for b, k in picks}
    res = bench.spot_check(syn, rows, mid, P, kept, picks, args, tail_clocks, V, loop, fs, 4)
    assert all(r["bit_exact"] for r in res), res
    assert bool((a[..., 0] == 0).all())
    syn.close()
