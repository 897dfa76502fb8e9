"""GPU tier: engine life-cycle behaviour behind the C-ABI that the audio parity tests do not reach -- the source arena
under load / release cycles (SamplerSynth::registerClip / unregisterClip, reference lib/SamplerSynth.cpp:285-312) and the
ordering of engine-stream work behind batches queued on a caller's stream."""
import ctypes as C

import numpy as np
import pytest

from scenario import Scene, compare_runs, play_cmd, rand_source, random_scene, run_backend, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the engine has no CPU path")
    from libzl_amd import SamplerSynth
    return SamplerSynth


def test_arena_space_is_reused_after_release(Engine):
    """Cumulative uploads of 12x the arena: every release returns its extent (coalesced), so loading never fails, and a
    clip loaded into recycled space renders the same bits as the oracle."""
    from libzl_amd import ZlHipError, clip_command
    from libzl_amd.engine import synthetic_clocks
    from oracle import zl_oracle as zo
    rng = np.random.default_rng(3)
    arena = 1 << 20                                                       # 1 MiB = 131072 stereo frames
    syn = Engine(2, 4, max_frames=128, max_batch_blocks=4, max_sounds=16, sound_arena_bytes=arena, sound_arena_max_bytes=arena)   # a FIXED arena
    live = []
    total = 0
    for i in range(60):
        n = int(rng.integers(20000, 50000))                               # 160-400 KB each
        L, R = rand_source(rng, n, stereo=(i % 3 != 0))
        while True:
            try:
                cid = syn.register_clip(L, R, 48000.0)
                break
            except ZlHipError:
                assert live, "the arena is empty and a clip that fits was refused"
                syn.unregister_clip(live.pop(int(rng.integers(0, len(live))))[0])     # full: the host frees a clip, as zynthbox does
        live.append((cid, L, R))
        total += n * (8 if R is not None else 4)
    assert total > 12 * arena
    # fragmentation check: free everything, then one clip that needs almost the whole arena must fit (extents coalesced)
    for cid, _, _ in live:
        syn.unregister_clip(cid)
    big = (arena // 8) - 64
    L, R = rand_source(rng, big, stereo=True)
    cid = syn.register_clip(L, R, 48000.0)
    # and it plays correctly from the recycled space
    osyn = zo.OracleSynth(2, 4, 48000.0, 0)
    oid = osyn.register_clip(L, R, 48000.0)
    p = syn.default_clip_params(big / 48000.0); p.length_in_beats = 0.3; p.length_seconds = 0.011
    syn.set_clip_params(cid, p)
    osyn.clips[oid].lengthInBeats = 0.3; osyn.clips[oid].lengthInSeconds = float(np.float32(0.011))
    syn.handle_clip_command(clip_command(clip=cid, midi_note=62, midi_channel=-1, start_playback=1, looping=1, change_volume=1, volume=0.8), 0)
    osyn.handle_clip_command(zo.clip_command(clip=oid, midiNote=62, midiChannel=-1, startPlayback=1, looping=1, changeVolume=1, volume=0.8), 0)
    clk = synthetic_clocks(4, 128, 48000.0)
    syn.render_batch(4, 128, clk)
    ref, _ = osyn.render_batch(4, 128, clk)
    assert np.array_equal(syn.read_bus().view(np.int32), ref.view(np.int32))
    syn.close()


def test_failed_upload_keeps_neither_slot_nor_space(Engine):
    from libzl_amd import ZlHipError
    rng = np.random.default_rng(4)
    syn = Engine(1, 2, max_frames=64, max_batch_blocks=1, max_sounds=4, sound_arena_bytes=1 << 16, sound_arena_max_bytes=1 << 16)
    L, R = rand_source(rng, 100000, stereo=True)                          # 800 KB into a 64 KB arena
    for _ in range(8):                                                    # more failures than sound slots
        with pytest.raises(ZlHipError):
            syn.register_clip(L, R, 48000.0)
    ids = [syn.register_clip(L[:1000], R[:1000], 48000.0) for _ in range(4)]
    assert sorted(ids) == [0, 1, 2, 3]
    syn.close()


def test_levels_tick_is_ordered_behind_a_batch_on_the_callers_stream(Engine):
    """zlhip_render_batch(stream = caller's) followed at once by zlhip_levels_tick: the tick kernel runs on the engine's own
    (non-blocking) stream and must still see the block levels the batch produces."""
    import torch
    from oracle import zl_oracle as zo
    from scenario import snapshot_clip
    lib = zo.load()
    sc = random_scene(77, num_buses=8, voices_per_bus=32, nblocks=256, nframes=256, nclips=40, events=False, min_len=4000, max_len=9000)
    ref_bus, _, _ = run_oracle(sc, threads=8)
    ref = zo.OracleSynth(1, 1, sc.fs, sc.mode, max_sounds=64)
    syn = Engine(num_buses=8, voices_per_bus=32, max_frames=256, max_batch_blocks=256, max_sounds=64,
                 sound_arena_bytes=sum((s[0].shape[0] + 16) * 8 for s in sc.sounds) + (1 << 16))
    from scenario import engine_cmd
    for i, (L, R, sr) in enumerate(sc.sounds):
        ref.register_clip(L, R, sr); syn.register_clip(L, R, sr)
        sc.clip_setup[i](ref.lib, ref.clips[i])
        syn.set_clip_params(i, snapshot_clip(ref.clips[i]))
    for ev in sc.events[0]:
        syn.handle_clip_command(engine_cmd(**ev[1]), ev[2])
    stream = torch.cuda.Stream()
    out = torch.zeros((8, 2, 256 * 256), device="cuda", dtype=torch.float32)
    # a long-running kernel in front of the batch on the caller's stream widens the window in which an unordered tick
    # would read the previous contents of the level buffer
    with torch.cuda.stream(stream):
        junk = torch.rand((1 << 27,), device="cuda")
        for _ in range(4):
            junk = junk * 1.0001 + 0.5
    syn.render_batch(256, 256, sc.make_clocks(0, 256), bus_out_dev=out.data_ptr(), stream=stream.cuda_stream)
    lv = syn.levels_tick(block_index=255, with_hold_bus=-1)              # no synchronisation in between
    bus = out.cpu().numpy()
    assert np.array_equal(bus.view(np.int32), ref_bus.view(np.int32))
    for b in range(8):
        ch = zo.LevelsChannel()
        L = np.ascontiguousarray(bus[b, 0, 255 * 256:]); R = np.ascontiguousarray(bus[b, 1, 255 * 256:])
        lib.zlo_levels_tick(C.byref(ch), L.ctypes.data, R.ctypes.data, 256, 0)
        assert (lv[b].peak_a, lv[b].peak_b) == (ch.peakA, ch.peakB), b
    syn.close()


def test_device_upload_is_ordered_behind_its_producer(Engine):
    """zlhip_sound_upload_device right after the kernels that fill the planes, no synchronisation by the caller."""
    import torch
    from libzl_amd import clip_command
    from libzl_amd.engine import synthetic_clocks
    from oracle import zl_oracle as zo
    n = 1 << 22
    syn = Engine(1, 2, max_frames=128, max_batch_blocks=2, max_sounds=2, sound_arena_bytes=(n + 64) * 8)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        src = torch.rand((2, n), generator=g, device="cuda") * 2 - 1
        for _ in range(6):
            src = src * 0.999                                           # a chain of kernels still running when upload is called
        cid = syn.register_clip_device(src[0].data_ptr(), src[1].data_ptr(), n, 48000.0)
    host = src.cpu().numpy()
    osyn = zo.OracleSynth(1, 2, 48000.0, 0)
    oid = osyn.register_clip(np.ascontiguousarray(host[0]), np.ascontiguousarray(host[1]), 48000.0)
    syn.handle_clip_command(clip_command(clip=cid, midi_note=60, midi_channel=-2, start_playback=1, looping=1, change_volume=1, volume=1.0), 0)
    osyn.handle_clip_command(zo.clip_command(clip=oid, midiNote=60, midiChannel=-2, startPlayback=1, looping=1, changeVolume=1, volume=1.0), 0)
    clk = synthetic_clocks(2, 128, 48000.0)
    syn.render_batch(2, 128, clk)
    ref, _ = osyn.render_batch(2, 128, clk)
    assert np.array_equal(syn.read_bus().view(np.int32), ref.view(np.int32))
    syn.close()


@pytest.mark.parametrize("npieces,nframes,mode", [(2, 128, 0), (8, 256, 0), (3, 64, 2), (11, 512, 0), (4, 100, 0), (3, 33, 2), (2, 441, 0)])
def test_bus_reduce_sum_scan_kernel(Engine, npieces, nframes, mode):
    """zlhip_bus_reduce_sum_scan: the rank-order sum of received pieces and the level scan of every unit in one kernel, and
    zlhip_levels_import_units on the root -- against the defined arithmetic: ((0 + p0) + p1) + ... in fp32, integer peaks,
    the oracle's sum-of-squares order (tile offset 1 / 0 by mode)."""
    import torch
    from oracle import zl_oracle as zo
    lib = zo.load()
    B, K = 3, 5
    units = B * 2 * K                                                # one rank holding the whole (reduced) bus: every unit
    g = torch.Generator(device="cuda"); g.manual_seed(100 + npieces)
    stride = units * nframes + 192                                   # pieces need not be packed
    pieces = (torch.rand((npieces, stride), generator=g, device="cuda") * 2 - 1) * 3.0
    pieces[1, 5 * nframes:6 * nframes] = -0.0                       # a unit of negative zeros in one piece: 0 + (-0) = +0
    out = torch.full((units * nframes,), 7.0, device="cuda")
    lv = torch.zeros((units, 2), dtype=torch.int32, device="cuda")
    syn = Engine(B, 2, max_frames=max(64, nframes), max_batch_blocks=K, max_sounds=2, mode=mode)
    syn.bus_reduce_sum_scan(pieces.data_ptr(), npieces, stride, units, nframes, out.data_ptr(), lv.data_ptr())
    syn.levels_import_units(lv.data_ptr(), K, nframes)
    peaks = syn.block_peaks()                                        # waits for the engine
    ph = pieces.cpu().numpy()
    acc = np.zeros(units * nframes, dtype=np.float32)
    for r in range(npieces):
        acc = acc + ph[r, :units * nframes]
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.int32), acc.view(np.int32))
    rows = np.ascontiguousarray(acc.reshape(B, 2, K, nframes))
    assert np.array_equal(peaks, np.abs(np.float32(131072.0) * rows).astype(np.int64).max(axis=3).transpose(2, 0, 1))
    lvh = lv.cpu().numpy()
    off = 0 if (mode & 2) else 1
    for u in range(units):
        row = np.ascontiguousarray(acc[u * nframes:(u + 1) * nframes])
        assert lvh[u, 1:].view(np.float32)[0] == lib.zlo_block_sumsq(row.ctypes.data, nframes, off), u
    t = syn.levels_tick(block_index=2)
    for b in range(B):
        assert t[b].rms_a == lib.zlo_block_rms(np.ascontiguousarray(rows[b, 0, 2]).ctypes.data, nframes, off)
        assert t[b].rms_b == lib.zlo_block_rms(np.ascontiguousarray(rows[b, 1, 2]).ctypes.data, nframes, off)
    syn.close()


def test_setters_race_free_next_to_the_process_thread(built, tmp_path):
    """the ClipAudioSource_set* entry points run on the host's UI thread while the JACK thread calls
    libzl_hotpath_process (reference: setters on the message thread, Helper.h:8-26).  Every entry point that reaches the
    engine or the clip list takes the bridge's mutex: two threads hammering both sides for 300 cycles must neither crash nor
    produce a non-finite sample, and the final parameter values must be the last ones written."""
    import threading
    from libzl_amd import libzl
    from libzl_amd.engine import synthetic_clocks
    zl = libzl.load()
    zl.initJuce()
    assert zl.libzl_hotpath_status() == 0
    rng = np.random.default_rng(9)
    clips = []
    for i in range(4):
        L = rng.uniform(-1, 1, 6000).astype(np.float32); R = rng.uniform(-1, 1, 6000).astype(np.float32)
        c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, R.ctypes.data, 6000, 48000.0, f"r{i}".encode())
        zl.ClipAudioSource_setLength(c, 0.2, 120)
        zl.ClipAudioSource_playOnChannel(c, True, i)
        clips.append(c)
    stop = threading.Event()
    errors = []

    def ui():
        k = 0
        while not stop.is_set():
            c = clips[k % 4]
            zl.ClipAudioSource_setPan(c, ((k % 21) - 10) / 10.0)
            zl.ClipAudioSource_setVolumeAbsolute(c, 0.1 + (k % 9) / 10.0)
            zl.ClipAudioSource_setStartPosition(c, (k % 5) * 0.001)
            zl.ClipAudioSource_setADSRRelease(c, 0.01 + (k % 3) * 0.01)
            zl.ClipAudioSource_setSlices(c, 4 + k % 13)
            zl.ClipAudioSource_peakGain(c); zl.ClipAudioSource_firstProgress(c); zl.ClipAudioSource_byID(zl.ClipAudioSource_id(c))
            k += 1

    th = threading.Thread(target=ui)
    th.start()
    N = 128
    outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
    try:
        for k in range(300):
            if zl.libzl_hotpath_process(N, synthetic_clocks(1, N, 48000.0, start_block=k), outL.ctypes.data, outR.ctypes.data) != 0:
                errors.append(k)
            if not (np.isfinite(outL).all() and np.isfinite(outR).all()):
                errors.append(("nan", k))
    finally:
        stop.set(); th.join()
    assert not errors
    zl.ClipAudioSource_setPan(clips[0], 0.25)
    for c in clips:
        zl.ClipAudioSource_destroy(c)
    zl.shutdownJuce()


def test_peak_conversion_of_non_finite_and_huge_samples(Engine):
    """(int) abs(131072.f * x) on the device for the cases C leaves undefined -- NaN, infinities, values beyond int -- and at the
    edges around them: the oracle's definition (NaN -> 0, saturation) bit for bit, through the scan of a device bus (K3; K2's fused
    scan uses the same function)."""
    import torch
    from oracle import zl_oracle as zo
    lib = zo.load()
    special = np.array([np.nan, np.inf, -np.inf, 16383.99, 16384.0, -16384.0, 16384.004, 3.0e38, -3.0e38, 0.0, -0.0, 1e-45, -1e-39,
                        7.62939453125e-06, 7.6293e-06, -1.0, 0.99999994, 16383.998046875], dtype=np.float32)
    rng = np.random.default_rng(9)
    N, K, B = 64, 6, 2
    bus = rng.uniform(-0.5, 0.5, (B, 2, K * N)).astype(np.float32)
    # one special value per (bus, channel, block) row at most, so that every one of them decides a peak or leaves it alone
    rows = [(b, c, k) for b in range(B) for c in range(2) for k in range(K)]
    for (b, c, k), v in zip(rows, special):
        bus[b, c, k * N + 1 + int(rng.integers(0, N - 1))] = v
    syn = Engine(num_buses=B, voices_per_bus=8, max_frames=N, max_batch_blocks=K, max_sounds=4, mode=2)     # FIX_DELAY: frame 0 counts like the others
    dev = torch.from_numpy(bus).cuda()
    syn.levels_scan_device(dev.data_ptr(), K, N)
    got = syn.block_peaks()
    want = np.zeros((K, B, 2), dtype=np.int64)
    for b in range(B):
        for c in range(2):
            for k in range(K):
                want[k, b, c] = max(lib.zlo_sample_to_peak_int(float(x)) for x in bus[b, c, k * N:(k + 1) * N])
    assert np.array_equal(got, want)
    assert want.max() == 2147483647 and (want == 0).sum() == 0
    syn.close()



def test_the_source_arena_grows_in_segments(Engine):
    """Clips are loaded freely (SamplerSynth::registerClip has no budget, SamplerSynth.cpp:285-295): when a source does not fit the
    arena reserved at creation the engine allocates another segment; sources in later segments (wherever the allocator put them in
    the address space) render the same bits as the oracle; released extents of every segment are reused; a cap makes loading fail."""
    from libzl_amd import ZlHipError, clip_command
    from libzl_amd.engine import synthetic_clocks
    from oracle import zl_oracle as zo
    rng = np.random.default_rng(11)
    arena = 1 << 20
    syn = Engine(2, 8, max_frames=128, max_batch_blocks=8, max_sounds=32, sound_arena_bytes=arena)
    osyn = zo.OracleSynth(2, 8, 48000.0, 0, max_sounds=32)
    total0, a0 = syn.memory_bytes()
    assert a0 == arena
    clips = []
    for i in range(12):
        n = int(rng.integers(50000, 70000)) if i != 7 else 400000          # 400-560 KB each; one source of 3.2 MB (larger than a segment)
        L, R = rand_source(rng, n, stereo=(i % 4 != 1))
        cid = syn.register_clip(L, R, [48000.0, 44100.0][i % 2])
        oid = osyn.register_clip(L, R, [48000.0, 44100.0][i % 2])
        assert cid == oid == i
        p = syn.default_clip_params(n / [48000.0, 44100.0][i % 2]); p.length_in_beats = 0.41; p.length_seconds = float(np.float32(0.02 + 0.003 * i)); p.pan = 0.1 * i - 0.5
        syn.set_clip_params(cid, p)
        oc = osyn.clips[oid]; oc.lengthInBeats = 0.41; oc.lengthInSeconds = float(np.float32(0.02 + 0.003 * i)); oc.pan = float(np.float32(0.1 * i - 0.5))
        clips.append((L, R))
    total1, a1 = syn.memory_bytes()
    assert a1 >= 6 * arena and total1 - total0 >= a1 - a0
    for i in range(12):
        f = dict(clip=i, midi_note=57 + i, midi_channel=(i % 2) - 2, start_playback=1, looping=1, change_volume=1, volume=0.5 + 0.03 * i)
        assert syn.handle_clip_command(clip_command(**f), 0) == 1
        osyn.handle_clip_command(zo.clip_command(clip=i, midiNote=57 + i, midiChannel=(i % 2) - 2, startPlayback=1, looping=1, changeVolume=1, volume=0.5 + 0.03 * i), 0)
    clk = synthetic_clocks(8, 128, 48000.0)
    syn.render_batch(8, 128, clk)
    ref, _ = osyn.render_batch(8, 128, clk)
    assert np.array_equal(syn.read_bus().view(np.int32), ref.view(np.int32)) and np.abs(ref).max() > 0.5
    # real-time cycles from the same sources (the resident kernel addresses them the same way)
    L1, R1 = syn.process(128, synthetic_clocks(1, 128, 48000.0, start_block=8)[0])
    ref1, _ = osyn.render_batch(1, 128, synthetic_clocks(1, 128, 48000.0, start_block=8))
    assert np.array_equal(L1.view(np.int32), ref1[:, 0].view(np.int32)) and np.array_equal(R1.view(np.int32), ref1[:, 1].view(np.int32))
    # a segment all of whose sources are released goes back to the device (memory only ever grew): the voices are stopped,
    # the large source -- a segment of its own -- and then everything else is released; what stays is the arena reserved at creation
    for i in range(12):
        syn.stop_voice(i % 2, i // 2, False)
    syn.process(128, synthetic_clocks(1, 128, 48000.0, start_block=9)[0])
    syn.unregister_clip(7)
    _, a2 = syn.memory_bytes()
    assert a2 < a1 and a2 >= arena
    for i in range(12):
        if i != 7:
            syn.unregister_clip(i)
    total3, a3 = syn.memory_bytes()
    assert a3 == arena and total3 == total0
    # ... and the engine goes on: new sources, same bits as the oracle
    for i in range(3):
        cid = syn.register_clip(*clips[i], 48000.0)
        p = syn.default_clip_params(len(clips[i][0]) / 48000.0); p.length_in_beats = 0.41; p.length_seconds = float(np.float32(0.03)); syn.set_clip_params(cid, p)
    o2 = zo.OracleSynth(2, 8, 48000.0, 0, max_sounds=32)
    for i in range(3):
        oid = o2.register_clip(*clips[i], 48000.0)
        oc = o2.clips[oid]; oc.lengthInBeats = 0.41; oc.lengthInSeconds = float(np.float32(0.03))
        assert syn.handle_clip_command(clip_command(clip=i, midi_note=60, midi_channel=-2, start_playback=1, looping=1, change_volume=1, volume=0.7), 0) == 1
        o2.handle_clip_command(zo.clip_command(clip=oid, midiNote=60, midiChannel=-2, startPlayback=1, looping=1, changeVolume=1, volume=0.7), 0)
    clk2 = synthetic_clocks(8, 128, 48000.0, start_block=10)
    syn.render_batch(8, 128, clk2)
    ref2, _ = o2.render_batch(8, 128, clk2)
    assert np.array_equal(syn.read_bus().view(np.int32), ref2.view(np.int32)) and np.abs(ref2).max() > 0.3
    syn.close()
    # capped: the same uploads stop fitting
    cap = Engine(2, 8, max_frames=128, max_batch_blocks=8, max_sounds=32, sound_arena_bytes=arena, sound_arena_max_bytes=3 * arena)
    ok = 0
    with pytest.raises(ZlHipError):
        for L, R in clips:
            cap.register_clip(L, R, 48000.0)
            ok += 1
    assert 2 <= ok <= 7
    cap.close()



def test_device_upload_behind_the_producers_stream(Engine):
    """zlhip_sound_upload_device_on: source planes produced on a side stream (a long chain of kernels) are uploaded with that stream
    as the only thing waited for -- no device-wide wait -- and the engine reads the finished planes: the render equals the oracle's
    on the same data."""
    import torch
    from libzl_amd import clip_command
    from libzl_amd.engine import synthetic_clocks
    from oracle import zl_oracle as zo
    n = 200_000
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        x = torch.rand((2, n), generator=g, device="cuda") * 2 - 1
        for _ in range(200):                       # keep the producer busy: the planes are final only at the end of the chain
            x = torch.roll(x, 1, dims=1) * 0.999
    syn = Engine(1, 2, max_frames=128, max_batch_blocks=4, max_sounds=2, sound_arena_bytes=(n + 64) * 8)
    cid = syn.register_clip_device_on(x[0].data_ptr(), x[1].data_ptr(), n, 48000.0, side.cuda_stream)
    side.synchronize()
    h = x.cpu().numpy()
    osyn = zo.OracleSynth(1, 2, 48000.0, 0, max_sounds=2)
    oid = osyn.register_clip(np.ascontiguousarray(h[0]), np.ascontiguousarray(h[1]), 48000.0)
    syn.handle_clip_command(clip_command(clip=cid, midi_note=60, midi_channel=-2, start_playback=1, looping=1, change_volume=1, volume=0.9), 0)
    osyn.handle_clip_command(zo.clip_command(clip=oid, midiNote=60, midiChannel=-2, startPlayback=1, looping=1, changeVolume=1, volume=0.9), 0)
    clk = synthetic_clocks(4, 128, 48000.0)
    syn.render_batch(4, 128, clk)
    ref, _ = osyn.render_batch(4, 128, clk)
    assert np.array_equal(syn.read_bus().view(np.int32), ref.view(np.int32)) and np.abs(ref).max() > 0.1
    syn.close()
