"""Offline bounce (zlhip_bounce, BASELINE configs[4]) and the recorder's 16-bit sample format (AudioLevels.cpp:53-58)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import np_restatement as npr
from oracle import zl_oracle as zo

f32 = np.float32


@pytest.fixture(scope="module")
def Engine(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the engine has no CPU path")
    from libzl_amd import SamplerSynth
    return SamplerSynth


def test_pcm16_known_answers_and_restatements_agree(built):
    lib = zo.load()
    # hand-derived: q = rint(0x7fffffff * x) (half to even), sample = q >> 16 (arithmetic: towards minus infinity)
    kat = [(0.0, 0), (1.0, 32767), (-1.0, -32768), (2.0, 32767), (-3.0, -32768), (0.5, 16384), (-0.5, -16384),
           (2.0 ** -15, 1), (2.0 ** -16, 0), (-1e-9, -1), (1e-9, 0), (float("inf"), 32767), (float("-inf"), -32768), (float("nan"), 0),
           (0.999969482421875, 32766), (np.nextafter(f32(1.0), f32(0.0)), 32767)]
    for x, want in kat:
        assert lib.zlo_pcm16_sample(float(f32(x))) == want, (x, want)
        assert int(npr.pcm16(np.array([x], dtype=f32))[0]) == want, (x, want)
    rng = np.random.default_rng(16)
    x = np.concatenate([rng.uniform(-1.2, 1.2, 100000), rng.normal(0, 1e-4, 20000), (np.arange(-40000, 40000) + 0.5) / 32768.0]).astype(f32)
    out = np.empty((len(x), 2), dtype=np.int16)
    lib.zlo_pcm16_stereo(x.ctypes.data, x[::-1].copy().ctypes.data, len(x), out.ctypes.data)
    assert np.array_equal(out[:, 0], npr.pcm16(x)) and np.array_equal(out[:, 1], npr.pcm16(x[::-1]))


def test_wav_16_bit_files_hold_the_recorder_format(built, tmp_path):
    from libzl_amd import libzl
    zl = libzl.load()
    rng = np.random.default_rng(17)
    L = rng.uniform(-1.1, 1.1, 999).astype(f32); R = rng.uniform(-1.1, 1.1, 999).astype(f32)
    a, b = str(tmp_path / "a.wav").encode(), str(tmp_path / "b.wav").encode()
    assert zl.libzl_wav_write(a, L.ctypes.data, R.ctypes.data, len(L), 48000.0, 16) == 0
    frames = np.stack([npr.pcm16(L), npr.pcm16(R)], axis=1).copy()
    assert zl.libzl_wav_write_interleaved(b, frames.ctypes.data, len(L), 2, 48000.0, 16) == 0
    blob = open(a, "rb").read()
    assert blob == open(b, "rb").read()
    assert blob[:4] == b"RIFF" and blob[36:40] == b"data" and np.array_equal(np.frombuffer(blob[44:], dtype="<i2").reshape(-1, 2), frames)
    assert zl.libzl_wav_write_interleaved(b, frames.ctypes.data, len(L), 3, 48000.0, 16) != 0      # mono or stereo only
    assert zl.libzl_wav_write_interleaved(b, None, 4, 2, 48000.0, 16) != 0


@pytest.mark.gpu
def test_bounce_equals_consecutive_batches_and_the_oracle(Engine):
    """A scene with commands in the middle: every stretch between events is one zlhip_bounce call cut into ragged sub-batches;
    the host buffers hold the oracle's mix bit for bit, and reports / levels carry on as after zlhip_render_batch."""
    from scenario import compare_runs, random_scene, run_backend, run_oracle
    sc = random_scene(0xB0, num_buses=5, voices_per_bus=8, nclips=12, nframes=128, nblocks=61)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, bounce=("f32", 7))
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 40)
    # the engine's own last sub-batch is still readable on the device side, with its levels
    last = syn.read_bus()
    K = last.shape[2] // sc.nframes
    assert np.array_equal(last.view(np.int32), bus[:, :, -last.shape[2]:].view(np.int32))
    exp = np.abs(f32(131072.0) * last.reshape(5, 2, K, sc.nframes)).astype(np.int64).max(axis=3).transpose(2, 0, 1)
    assert np.array_equal(syn.block_peaks(), exp)
    syn.close()
    pcm, _, syn, _ = run_backend(sc, Engine, bounce=("pcm16", 7))
    want = np.stack([npr.pcm16(ref_bus[:, 0]), npr.pcm16(ref_bus[:, 1])], axis=2)
    assert np.array_equal(pcm, want)
    syn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nframes,sub", [(100, 7), (441, 3), (33, 16)])
def test_bounce_with_periods_that_are_no_multiple_of_64(Engine, nframes, sub):
    """the render kernel's own stores into the caller's page-locked buffer (fp32 planar and the recorder's 16-bit format) when a block's
    last wave has lanes behind the block's end: they store nothing, the rows stay contiguous"""
    from scenario import compare_runs, random_scene, run_backend, run_oracle
    sc = random_scene(0xB7 + nframes, num_buses=5, voices_per_bus=8, nclips=12, nframes=nframes, nblocks=37)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    bus, rep, syn, _ = run_backend(sc, Engine, bounce=("f32", sub))
    compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 40)
    syn.close()
    pcm, _, syn, _ = run_backend(sc, Engine, bounce=("pcm16", sub))
    want = np.stack([npr.pcm16(ref_bus[:, 0]), npr.pcm16(ref_bus[:, 1])], axis=2)
    assert np.array_equal(pcm, want)
    syn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nframes,sub", [(441, 3), (33, 7), (100, 5)])
@pytest.mark.parametrize("how", ["pageable", "copy_engine"])
def test_bounce_through_the_copy_engine_with_odd_periods(Engine, nframes, sub, how):
    """The copy-engine delivery (pageable destination, or ZL_BOUNCE_DIRECT=0) converts every plan window with zl_k_deliver first: window
    lengths K * nframes that are no multiple of four frames -- 441 x 3, 33 x 7 -- have a per-frame tail, and rows / offsets that are no
    multiple of four take the per-frame form (the last 1-3 frames of a window used to stay unconverted)."""
    import os
    from scenario import compare_runs, random_scene, run_backend, run_oracle
    sc = random_scene(0xC1 + nframes, num_buses=3, voices_per_bus=8, nclips=10, nframes=nframes, nblocks=23)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    want = np.stack([npr.pcm16(ref_bus[:, 0]), npr.pcm16(ref_bus[:, 1])], axis=2)
    old = os.environ.get("ZL_BOUNCE_DIRECT")
    if how == "copy_engine":
        os.environ["ZL_BOUNCE_DIRECT"] = "0"
    try:
        extra = ("pageable",) if how == "pageable" else ()
        pcm, _, syn, _ = run_backend(sc, Engine, bounce=("pcm16", sub) + extra)
        assert np.array_equal(pcm, want), np.argwhere(pcm != want)[:4].tolist()
        syn.close()
        bus, rep, syn, _ = run_backend(sc, Engine, bounce=("f32", sub) + extra)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, 24)
        syn.close()
    finally:
        if old is None:
            os.environ.pop("ZL_BOUNCE_DIRECT", None)
        else:
            os.environ["ZL_BOUNCE_DIRECT"] = old


@pytest.mark.gpu
def test_bounce_into_caller_memory_and_argument_checks(Engine):
    from libzl_amd.engine import ZlHipError, synthetic_clocks
    syn = Engine(num_buses=2, voices_per_bus=4, max_frames=256, max_batch_blocks=8, max_sounds=4)
    rng = np.random.default_rng(3)
    L = rng.uniform(-1, 1, 5000).astype(f32)
    cid = syn.register_clip(L, None, 48000.0)
    from scenario import engine_cmd, play_cmd
    syn.handle_clip_command(engine_cmd(**play_cmd(cid)), 0)
    clocks = synthetic_clocks(20, 256, 48000.0)
    out = np.full((2, 2, 20 * 256), 9.0, dtype=f32)               # pageable memory of the caller: works, through staging copies
    got = syn.bounce(20, 256, clocks, out=out)
    assert got is out and np.abs(out[0]).max() > 0.1 and not (out == 9.0).any()
    pinned = syn.bounce(20, 256, synthetic_clocks(20, 256, 48000.0, start_block=20))
    assert pinned.shape == out.shape and np.isfinite(pinned).all()
    # both delivery paths (kernel stores into mapped page-locked memory / copy commands into pageable memory) give the same bytes
    syn2 = Engine(num_buses=2, voices_per_bus=4, max_frames=256, max_batch_blocks=8, max_sounds=4)
    syn2.register_clip(L, None, 48000.0)
    syn2.handle_clip_command(engine_cmd(**play_cmd(cid)), 0)
    first = syn2.bounce(20, 256, clocks)
    assert np.array_equal(first.view(np.int32), out.view(np.int32))
    pcm_pageable = syn.bounce(20, 256, synthetic_clocks(20, 256, 48000.0, start_block=40), fmt="pcm16", out=np.zeros((2, 20 * 256, 2), dtype=np.int16))
    syn2.bounce(20, 256, synthetic_clocks(20, 256, 48000.0, start_block=20))
    pcm_pinned = syn2.bounce(20, 256, synthetic_clocks(20, 256, 48000.0, start_block=40), fmt="pcm16")
    assert np.array_equal(pcm_pageable, pcm_pinned) and np.abs(pcm_pinned.astype(np.int32)).max() > 3000
    syn2.close()
    with pytest.raises(ValueError):
        syn.bounce(20, 256, clocks, out=np.zeros((2, 2, 5), dtype=f32))
    with pytest.raises(ZlHipError):
        syn.bounce(20, 100000, clocks)                             # nframes beyond max_frames
    with pytest.raises(ZlHipError):
        syn.bounce(0, 256, clocks)
    syn.close()


@pytest.mark.gpu
def test_bounce_to_wav_files_and_back(Engine, tmp_path):
    """The whole n3 chain on real files: clips loaded from WAV files (decode side), bounced on the GPU in the recorder's 16-bit format,
    one stereo WAV per bus written from the bounce buffer, read back: every sample is the oracle's mix in the oracle's 16-bit format."""
    from libzl_amd import libzl
    from scenario import compare_runs, random_scene, run_backend, run_oracle
    zl = libzl.load()
    sc = random_scene(0xB1, num_buses=3, voices_per_bus=8, nclips=8, nframes=256, nblocks=20, events=False)
    # the scene's sources go through 32-bit float WAV files first (bit-preserving), as ClipAudioSource_new(path) would load them
    for i, (L, R, sr) in enumerate(sc.sounds):
        p = str(tmp_path / f"clip{i}.wav").encode()
        assert zl.libzl_wav_write(p, L.ctypes.data, None if R is None else R.ctypes.data, len(L), sr, 32) == 0
        Lp, Rp, n, rate = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_int(), C.c_double()
        assert zl.libzl_wav_read(p, C.byref(Lp), C.byref(Rp), C.byref(n), C.byref(rate)) == 0 and n.value == len(L) and rate.value == sr
        L2 = np.ctypeslib.as_array(Lp, (n.value,)).copy(); R2 = np.ctypeslib.as_array(Rp, (n.value,)).copy() if Rp else None
        zl.libzl_wav_free(Lp); zl.libzl_wav_free(Rp)
        assert np.array_equal(L2, L) and (R is None) == (R2 is None) and (R is None or np.array_equal(R2, R))
        sc.sounds[i] = (L2, R2, sr)
    ref_bus, _, _ = run_oracle(sc)
    pcm, _, syn, _ = run_backend(sc, Engine, bounce=("pcm16", 6))
    frames = sc.nblocks * sc.nframes
    scale = f32(1.0 / 2147483648.0)
    for b in range(sc.num_buses):
        p = str(tmp_path / f"bus{b}.wav").encode()
        row = np.ascontiguousarray(pcm[b])
        assert zl.libzl_wav_write_interleaved(p, row.ctypes.data, frames, 2, sc.fs, 16) == 0
        Lp, Rp, n, rate = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_int(), C.c_double()
        assert zl.libzl_wav_read(p, C.byref(Lp), C.byref(Rp), C.byref(n), C.byref(rate)) == 0 and n.value == frames and rate.value == sc.fs
        got = np.stack([np.ctypeslib.as_array(Lp, (frames,)).copy(), np.ctypeslib.as_array(Rp, (frames,)).copy()])
        zl.libzl_wav_free(Lp); zl.libzl_wav_free(Rp)
        want = (npr.pcm16(ref_bus[b]).astype(np.int32) << 16).astype(np.float32) * scale       # the reader's int -> float convention
        assert np.array_equal(got, want)
    assert np.abs(pcm.astype(np.int32)).max() > 2000
    syn.close()
