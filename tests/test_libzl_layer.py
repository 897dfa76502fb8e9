"""The libzl.h-named C-ABI (include/libzl_hotpath.h): parameter-setter semantics against the oracle's restated
setters and the RIFF/WAVE IO on the CPU; play/stop through ClipAudioSource_* against the oracle on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import zl_oracle as zo

f32 = np.float32


@pytest.fixture(scope="module")
def zl(built):
    from libzl_amd import libzl
    return libzl.load()


def _wav(tmp_path, zl, L, R, sr, bits, name="a.wav"):
    p = str(tmp_path / name).encode()
    assert zl.libzl_wav_write(p, L.ctypes.data, None if R is None else R.ctypes.data, len(L), sr, bits) == 0
    return p


def _read(zl, path):
    Lp, Rp = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    n, sr = C.c_int(), C.c_double()
    assert zl.libzl_wav_read(path, C.byref(Lp), C.byref(Rp), C.byref(n), C.byref(sr)) == 0
    L = np.ctypeslib.as_array(Lp, (n.value,)).copy()
    R = np.ctypeslib.as_array(Rp, (n.value,)).copy() if Rp else None
    zl.libzl_wav_free(Lp); zl.libzl_wav_free(Rp)
    return L, R, sr.value


def test_wav_roundtrip_float_and_pcm16(zl, tmp_path):
    rng = np.random.default_rng(5)
    L = rng.uniform(-1, 1, 777).astype(np.float32); R = rng.uniform(-1, 1, 777).astype(np.float32)
    l2, r2, sr = _read(zl, _wav(tmp_path, zl, L, R, 44100.0, 32))
    assert sr == 44100.0 and np.array_equal(l2, L) and np.array_equal(r2, R)
    l3, r3, _ = _read(zl, _wav(tmp_path, zl, L, None, 48000.0, 16, "b.wav"))
    assert r3 is None
    from oracle import np_restatement as npr
    q = npr.pcm16(L).astype(np.int32)                              # the recorder's format (tests/test_bounce.py)
    np.testing.assert_array_equal(l3, (q << 16).astype(np.float32) * f32(1.0 / 2147483648.0))   # JUCE int->float convention
    assert zl.libzl_wav_read(b"/nonexistent.wav", C.byref(C.POINTER(C.c_float)()), C.byref(C.POINTER(C.c_float)()), C.byref(C.c_int()), C.byref(C.c_double())) != 0
    assert zl.ClipAudioSource_new(b"/nonexistent.wav", False) is None       # reference: failures are logged, not raised


def test_wav_reader_formats_and_malformed_files(zl, tmp_path):
    """Hand-built RIFF files: PCM 8 / 24 / 32, float64, WAVE_FORMAT_EXTENSIBLE, three channels (the first two are kept,
    SamplerSynthSound.cpp:45) -- and files the reader must refuse without reading past their end."""
    import struct

    def riff(fmt_tag, channels, rate, bits, payload, ext_sub=None, fmt_extra=b""):
        block = channels * bits // 8
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * block, block, bits)
        if ext_sub is not None:
            fmt += struct.pack("<HHIH", 22, bits, 0, ext_sub) + b"\x00" * 14
        fmt += fmt_extra
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
        return b"RIFF" + struct.pack("<I", len(body)) + body

    def read(blob, name):
        p = tmp_path / name
        p.write_bytes(blob)
        Lp, Rp = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        n, sr = C.c_int(), C.c_double()
        rc = zl.libzl_wav_read(str(p).encode(), C.byref(Lp), C.byref(Rp), C.byref(n), C.byref(sr))
        if rc != 0:
            return rc, None, None
        L = np.ctypeslib.as_array(Lp, (n.value,)).copy()
        R = np.ctypeslib.as_array(Rp, (n.value,)).copy() if Rp else None
        zl.libzl_wav_free(Lp); zl.libzl_wav_free(Rp)
        return 0, L, R

    scale = f32(1.0 / 2147483648.0)
    rc, L, R = read(riff(1, 1, 8000, 8, bytes([0, 128, 255, 64])), "u8.wav")
    assert rc == 0 and R is None
    np.testing.assert_array_equal(L, (np.array([-128, 0, 127, -64], dtype=np.int32) << 24).astype(np.float32) * scale)
    s24 = [0x123456, -0x123456, 0x7fffff, -0x800000]
    rc, L, R = read(riff(1, 1, 48000, 24, b"".join(struct.pack("<i", v)[:3] for v in s24)), "s24.wav")
    np.testing.assert_array_equal(L, (np.array(s24, dtype=np.int32) << 8).astype(np.float32) * scale)
    rc, L, R = read(riff(1, 2, 48000, 32, struct.pack("<4i", 1 << 30, -(1 << 30), 5, -5)), "s32.wav")
    np.testing.assert_array_equal(L, np.array([1 << 30, 5], dtype=np.float32) * scale)
    np.testing.assert_array_equal(R, np.array([-(1 << 30), -5], dtype=np.float32) * scale)
    rc, L, R = read(riff(3, 1, 44100, 64, struct.pack("<3d", 0.25, -1.5, 1e-3)), "f64.wav")
    np.testing.assert_array_equal(L, np.array([0.25, -1.5, 1e-3], dtype=np.float64).astype(np.float32))
    rc, L, R = read(riff(0xFFFE, 3, 48000, 16, struct.pack("<6h", 100, 200, 300, -100, -200, -300), ext_sub=1), "ext3ch.wav")
    assert rc == 0
    np.testing.assert_array_equal(L, (np.array([100, -100], dtype=np.int32) << 16).astype(np.float32) * scale)
    np.testing.assert_array_equal(R, (np.array([200, -200], dtype=np.int32) << 16).astype(np.float32) * scale)
    # refused: float with 16 bits, PCM with 12 bits, a compressed format tag, a truncated header, an extensible header cut short
    assert read(riff(3, 1, 48000, 16, b"\x00" * 8), "f16.wav")[0] != 0
    assert read(riff(1, 1, 48000, 12, b"\x00" * 8), "s12.wav")[0] != 0
    assert read(riff(0x55, 2, 44100, 16, b"\x00" * 64), "mp3.wav")[0] != 0
    assert read(riff(1, 1, 48000, 16, b"\x00" * 8)[:30], "cut.wav")[0] != 0
    ext = riff(0xFFFE, 1, 48000, 16, b"", ext_sub=1)
    assert read(ext[:12 + 8 + 20], "extcut.wav")[0] != 0
    # a data chunk that claims more bytes than the file holds is clamped to the file
    blob = bytearray(riff(1, 1, 48000, 16, struct.pack("<4h", 1, 2, 3, 4)))
    blob[-12:-8] = struct.pack("<I", 4000)
    rc, L, _ = read(bytes(blob), "long.wav")
    assert rc == 0 and len(L) == 4


def test_setters_follow_the_reference_semantics(zl, tmp_path):
    """Every setter of the bridge against the oracle's restatement of ClipAudioSource.cpp (no GPU needed: without
    initJuce the clips only hold parameters)."""
    lib = zo.load()
    L = np.zeros(96000, dtype=np.float32)
    c = zl.ClipAudioSource_new(_wav(tmp_path, zl, L, None, 48000.0, 16, "c.wav"), False)
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(96000 / 48000.0), 48000.0)
    assert zl.ClipAudioSource_getDuration(c) == oc.duration == 2.0
    assert zl.ClipAudioSource_getFileName(c) == b"c.wav"
    assert zl.ClipAudioSource_id(c) >= 1 and zl.ClipAudioSource_byID(zl.ClipAudioSource_id(c)) == c
    assert zl.ClipAudioSource_byID(987654) is None
    # ADSR: ctor defaults, then quirk Q13 on every setter
    assert (zl.ClipAudioSource_adsrAttack(c), zl.ClipAudioSource_adsrRelease(c)) == (oc.adsr.p.attack, oc.adsr.p.release) == (0.0, f32(0.05))
    for name, fn in (("Release", lib.zlo_clip_set_adsr_release), ("Attack", lib.zlo_clip_set_adsr_attack),
                     ("Sustain", lib.zlo_clip_set_adsr_sustain), ("Decay", lib.zlo_clip_set_adsr_decay)):
        getattr(zl, f"ClipAudioSource_setADSR{name}")(c, 0.37)
        fn(C.byref(oc), C.c_float(0.37))
        got = tuple(getattr(zl, f"ClipAudioSource_adsr{n}")(c) for n in ("Attack", "Decay", "Sustain", "Release"))
        assert got == (oc.adsr.p.attack, oc.adsr.p.decay, oc.adsr.p.sustain, oc.adsr.p.release)
    # root note / key zone
    zl.ClipAudioSource_setRootNote(c, 64); zl.ClipAudioSource_setKeyZoneStart(c, 12); zl.ClipAudioSource_setKeyZoneEnd(c, 100)
    assert (zl.ClipAudioSource_rootNote(c), zl.ClipAudioSource_keyZoneStart(c), zl.ClipAudioSource_keyZoneEnd(c)) == (64, 12, 100)
    # volume: clamp of setVolumeAbsolute, -40 dB floor of setVolume
    zl.ClipAudioSource_setVolumeAbsolute(c, 1.5); assert zl.ClipAudioSource_volumeAbsolute(c) == 1.0
    zl.ClipAudioSource_setVolume(c, -40.0); assert zl.ClipAudioSource_volumeAbsolute(c) == 0.0
    zl.ClipAudioSource_setVolume(c, 6.0); assert abs(zl.ClipAudioSource_volumeAbsolute(c) - 1.0) < 1e-6
    assert abs(zl.dBFromVolume(1.0) - 6.0) < 1e-6
    assert zl.SyncTimer_getMultiplier() == 96
    zl.ClipAudioSource_destroy(c)


def test_passthrough_parameter_bridge(zl):
    from libzl_amd import PassthroughParams
    assert zl.JackPassthrough_getDryAmount(3) == 1.0 and zl.JackPassthrough_getPanAmount(-1) == 0.0
    zl.JackPassthrough_setPanAmount(3, -0.25); zl.JackPassthrough_setWetFx1Amount(3, 0.5); zl.JackPassthrough_setMuted(-1, True)
    assert zl.JackPassthrough_getPanAmount(3) == -0.25 and zl.JackPassthrough_getWetFx1Amount(3) == 0.5 and zl.JackPassthrough_getMuted(-1) == 1.0
    zl.JackPassthrough_setDryAmount(10, 0.1)            # out of range: ignored, getters return 0 (libzl.cpp:476-575)
    assert zl.JackPassthrough_getDryAmount(10) == 0.0 and zl.JackPassthrough_getDryAmount(-2) == 0.0
    p = PassthroughParams()
    assert zl.JackPassthrough_getParams(3, C.byref(p)) == 0 and (p.pan_amount, p.wet_fx1_amount, p.muted) == (-0.25, 0.5, 0)
    zl.JackPassthrough_setMuted(-1, False); zl.JackPassthrough_setPanAmount(3, 0.0); zl.JackPassthrough_setWetFx1Amount(3, 1.0)


@pytest.mark.gpu
def test_play_stop_through_the_libzl_symbols_matches_oracle(zl, tmp_path):
    """The calling pattern of the reference's test/playtest.py: initJuce, ClipAudioSource_new, setLength, play;
    audio pulled per JACK cycle.  Compared bit for bit with the oracle driven through its restated setters."""
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(11)
    lib = zo.load()
    zl.initJuce()
    assert zl.libzl_hotpath_status() == 0
    osyn = zo.OracleSynth(12, 8, 48000.0, 0)
    clips = []
    for i in range(3):
        n = 5000 + 700 * i
        L = rng.uniform(-1, 1, n).astype(np.float32); R = rng.uniform(-1, 1, n).astype(np.float32) if i != 1 else None
        path = _wav(tmp_path, zl, L, R, 44100.0, 32, f"clip{i}.wav")
        c = zl.ClipAudioSource_new(path, False)
        oid = osyn.register_clip(L, R, 44100.0)
        oc = osyn.clips[oid]
        zl.ClipAudioSource_setLength(c, 0.13 + 0.02 * i, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(0.13 + 0.02 * i), 120)
        zl.ClipAudioSource_setPan(c, -0.5 + 0.4 * i); lib.zlo_clip_set_pan(C.byref(oc), C.c_float(-0.5 + 0.4 * i))
        zl.ClipAudioSource_setVolumeAbsolute(c, 0.5 + 0.2 * i); lib.zlo_clip_set_volume_absolute(C.byref(oc), C.c_float(0.5 + 0.2 * i))
        zl.ClipAudioSource_setStartPosition(c, 0.004 * i); lib.zlo_clip_set_start_position(C.byref(oc), C.c_float(0.004 * i))
        zl.ClipAudioSource_setADSRRelease(c, 0.01); lib.zlo_clip_set_adsr_release(C.byref(oc), C.c_float(0.01))
        clips.append((c, oid))
    levels = []
    cb = __import__("libzl_amd.libzl", fromlist=["CB"]).CB(lambda db: levels.append(db))
    zl.ClipAudioSource_setAudioLevelChangedCallback(clips[0][0], cb)

    def ocmd(oid, ch, loop, stop_only=False):
        f = dict(clip=oid, midiChannel=ch, midiNote=60, stopPlayback=1) if stop_only else \
            dict(clip=oid, midiChannel=ch, midiNote=60, changeVolume=1, volume=1.0, looping=1 if loop else 0, startPlayback=1, **({"stopPlayback": 1} if loop else {}))
        osyn.handle_clip_command(zo.clip_command(**f), 0)

    N = 128
    outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
    for k in range(30):
        if k == 0:
            zl.ClipAudioSource_play(clips[0][0], True); ocmd(clips[0][1], -2, True)
            zl.ClipAudioSource_playOnChannel(clips[1][0], True, 3); ocmd(clips[1][1], 3, True)
        if k == 5:
            zl.ClipAudioSource_playOnChannel(clips[2][0], False, 0); ocmd(clips[2][1], 0, False)
        if k == 12:
            zl.ClipAudioSource_stopOnChannel(clips[1][0], 3); ocmd(clips[1][1], 3, False, stop_only=True)
        if k == 20:
            zl.ClipAudioSource_stop(clips[0][0])
            for ch in [-2, -1] + list(range(10)):
                ocmd(clips[0][1], ch, False, stop_only=True)
        clk = synthetic_clocks(1, N, 48000.0, start_block=k)
        assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
        bus, _ = osyn.render_batch(1, N, clk)
        assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
        if k == 3:
            assert zl.ClipAudioSource_peakGain(clips[0][0]) == lib.zlo_positions_peak_gain(C.byref(osyn.clips[clips[0][1]].positions))
            assert zl.ClipAudioSource_firstProgress(clips[0][0]) == lib.zlo_positions_first_progress(C.byref(osyn.clips[clips[0][1]].positions))
    assert levels and all(np.isfinite(levels))
    for c, _ in clips:
        zl.ClipAudioSource_destroy(c)
    zl.shutdownJuce()


@pytest.mark.gpu
def test_level_and_progress_callbacks_match_oracle_block_by_block(zl, tmp_path):
    """the per-clip level / progress chain (ClipAudioSource.cpp:88-113,225-240) under an injected
    millisecond clock.  120 JACK cycles of 128 frames (2.67 ms each); the clock advances 3 ms per cycle so the 30 ms and
    100 ms rate limits open and close many times; a one-shot starts, fades into its release tail and ends; a loop is
    stopped with a tail.  Every callback value AND the cycle in which it fired must equal the oracle's restatement."""
    from libzl_amd import libzl
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(23)
    lib = zo.load()
    now = [1_000_000]
    clock_cb = libzl.CLOCK_MS(lambda: now[0])
    zl.libzl_hotpath_set_clock_ms(clock_cb)
    try:
        zl.initJuce()
        assert zl.libzl_hotpath_status() == 0
        osyn = zo.OracleSynth(12, 8, 48000.0, 0)
        clips, meters, got_lvl, got_prog = [], [], [], []
        cbs = []
        cycle = [0]
        for i in range(3):
            n = 9000 + 1300 * i
            env = np.linspace(1.0, 0.05, n).astype(np.float32)               # a decaying clip: the meter has something to follow
            L = (rng.uniform(-1, 1, n).astype(np.float32) * env); R = (rng.uniform(-1, 1, n).astype(np.float32) * env) if i != 2 else None
            c = zl.ClipAudioSource_new(_wav(tmp_path, zl, L, R, 48000.0, 32, f"m{i}.wav"), False)
            oid = osyn.register_clip(L, R, 48000.0)
            oc = osyn.clips[oid]
            zl.ClipAudioSource_setLength(c, 0.3 + 0.05 * i, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(0.3 + 0.05 * i), 120)
            zl.ClipAudioSource_setStartPosition(c, 0.01 * i); lib.zlo_clip_set_start_position(C.byref(oc), C.c_float(0.01 * i))
            zl.ClipAudioSource_setADSRRelease(c, 0.02); lib.zlo_clip_set_adsr_release(C.byref(oc), C.c_float(0.02))
            m = zo.ClipMeter(); lib.zlo_clip_meter_init(C.byref(m))
            lv = libzl.CB(lambda db, i=i: got_lvl.append((cycle[0], i, db)))
            pg = libzl.CB(lambda s, i=i: got_prog.append((cycle[0], i, s)))
            zl.ClipAudioSource_setAudioLevelChangedCallback(c, lv)
            if i != 1:
                zl.ClipAudioSource_setProgressCallback(c, pg)                  # clip 1 has no progress callback (:228)
            cbs += [lv, pg]
            clips.append((c, oid)); meters.append(m)

        def ocmd(oid, ch, loop, stop_only=False):
            f = dict(clip=oid, midiChannel=ch, midiNote=60, stopPlayback=1) if stop_only else \
                dict(clip=oid, midiChannel=ch, midiNote=60, changeVolume=1, volume=1.0, looping=1 if loop else 0, startPlayback=1, **({"stopPlayback": 1} if loop else {}))
            osyn.handle_clip_command(zo.clip_command(**f), 0)

        N = 128
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        want_lvl, want_prog = [], []
        val = C.c_float()
        for k in range(120):
            cycle[0] = k
            now[0] += 3
            osyn.now_ms = now[0]
            if k == 2:
                zl.ClipAudioSource_play(clips[0][0], True); ocmd(clips[0][1], -2, True)
            if k == 10:
                zl.ClipAudioSource_playOnChannel(clips[1][0], False, 1); ocmd(clips[1][1], 1, False)      # one-shot: ends by itself
                zl.ClipAudioSource_playOnChannel(clips[2][0], True, 4); ocmd(clips[2][1], 4, True)
            if k == 70:
                zl.ClipAudioSource_stopOnChannel(clips[2][0], 4); ocmd(clips[2][1], 4, False, stop_only=True)
            if k == 90:
                zl.ClipAudioSource_stop(clips[0][0])
                for ch in [-2, -1] + list(range(10)):
                    ocmd(clips[0][1], ch, False, stop_only=True)
            clk = synthetic_clocks(1, N, 48000.0, start_block=k)
            assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
            bus, _ = osyn.render_batch(1, N, clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            for i, (c, oid) in enumerate(clips):
                oc = osyn.clips[oid]
                if lib.zlo_sync_audio_level(C.byref(meters[i]), C.byref(oc), now[0], C.byref(val)):
                    want_lvl.append((k, i, val.value))
                if lib.zlo_sync_progress(C.byref(meters[i]), C.byref(oc), 1 if i != 1 else 0, now[0], C.byref(val)):
                    want_prog.append((k, i, val.value))
                assert zl.ClipAudioSource_peakGain(c) == lib.zlo_positions_peak_gain(C.byref(oc.positions)), (k, i)
                assert zl.ClipAudioSource_firstProgress(c) == lib.zlo_positions_first_progress(C.byref(oc.positions)), (k, i)
        assert len(want_lvl) > 20 and len(want_prog) > 5            # the scene exercises both chains
        assert {i for _, i, _ in want_lvl} == {0, 1, 2}
        assert got_lvl == want_lvl
        assert got_prog == want_prog
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
        zl.shutdownJuce()
    finally:
        zl.libzl_hotpath_set_clock_ms(libzl.CLOCK_MS())


def test_clip_meter_restatement_known_answers():
    """Hand-derived values of the level chain (ClipAudioSource.cpp:88-113): first tick from -400 dB, the x0.94 fade
    (20 log10 0.94 = -0.5374 dB per tick), the 30 ms gate and the 0.1 dB notification threshold."""
    lib = zo.load()
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(1.0), 48000.0)
    m = zo.ClipMeter(); lib.zlo_clip_meter_init(C.byref(m))
    v = C.c_float()
    pid = lib.zlo_positions_create(C.byref(oc.positions), 0.0, 100)
    lib.zlo_positions_set_gain_and_progress(C.byref(oc.positions), pid, 0.5, 0.25, 100)
    assert lib.zlo_sync_audio_level(C.byref(m), C.byref(oc), 100, C.byref(v)) == 1
    assert abs(v.value - 20 * np.log10(0.5)) < 1e-5                     # -6.0206 dB
    assert lib.zlo_sync_audio_level(C.byref(m), C.byref(oc), 129, C.byref(v)) == 0   # inside the 30 ms gate
    lib.zlo_positions_set_gain_and_progress(C.byref(oc.positions), pid, 0.0, 0.5, 131)
    assert lib.zlo_sync_audio_level(C.byref(m), C.byref(oc), 131, C.byref(v)) == 1   # the bar fades instead of dropping
    assert abs(v.value - (20 * np.log10(0.5) + 20 * np.log10(0.94))) < 1e-4
    assert lib.zlo_sync_progress(C.byref(m), C.byref(oc), 1, 131, C.byref(v)) == 1 and v.value == 0.5   # progress 0.5 x 1 s
    assert lib.zlo_sync_progress(C.byref(m), C.byref(oc), 1, 200, C.byref(v)) == 0                       # 100 ms gate
    lib.zlo_positions_set_gain_and_progress(C.byref(oc.positions), pid, 0.0, 0.5004, 240)
    assert lib.zlo_sync_progress(C.byref(m), C.byref(oc), 1, 240, C.byref(v)) == 0                       # moved by < 0.001


# ---- the libzl-named command path: SyncTimer::scheduleClipCommand's merge and the dispatch tick ----------------------
class _OracleHost:
    """The reference side of a libzl-named session: ClipAudioSource::play / stop build their ClipCommands
    (ClipAudioSource.cpp:415-455), SyncTimer merges and dispatches them, SamplerSynth renders."""

    def __init__(self, nframes, fs=48000.0):
        self.lib = zo.load()
        self.osyn = zo.OracleSynth(12, 8, fs, 0)
        self.N, self.fs = nframes, fs

    def play(self, oid, loop, ch=-2):
        f = dict(clip=oid, midiChannel=ch, midiNote=60, changeVolume=1, volume=1.0, looping=1 if loop else 0, startPlayback=1)
        if loop:
            f["stopPlayback"] = 1
        return zo.clip_command(**f)

    def stop(self, oid, ch=-3):
        chans = [ch] if ch > -3 else [-2, -1] + list(range(10))
        return [zo.clip_command(clip=oid, midiChannel=c, midiNote=60, stopPlayback=1) for c in chans]

    def render(self, clk):
        bus, _ = self.osyn.render_batch(1, self.N, clk)
        return bus

    def voices_playing(self, oid):
        return sum(1 for v in self.osyn.voices if v.isPlaying and v.clip == oid)


def _mk_clips(zl, host, rng, n, setup):
    out = []
    for i in range(n):
        ln = 7000 + 900 * i
        L = rng.uniform(-1, 1, ln).astype(np.float32); R = rng.uniform(-1, 1, ln).astype(np.float32) if i % 2 == 0 else None
        c = zl.ClipAudioSource_newFromBuffer(L.ctypes.data, None if R is None else R.ctypes.data, ln, 48000.0, f"c{i}".encode())
        assert c
        oid = host.osyn.register_clip(L, R, 48000.0)
        setup(i, c, host.osyn.clips[oid])
        out.append((c, oid))
    return out


@pytest.mark.gpu
def test_play_and_stop_through_the_libzl_names_merge_per_step_and_carry_the_playhead(zl):
    """VERDICT r2 items 1-2.  Host-owned transport (libzl_hotpath_process; the host's SyncTimer getters in the clock, here a timer
    that has been running for 20 000 cycles: playhead ~ 10 000 and moving).  Through the libzl symbols, bit-exact against the
    oracle's merge (zlo_step_schedule) + SamplerSynth, cycle by cycle:
      * play + stop inside one cycle: the stop folds into the play (stopPlayback is not copied): the clip KEEPS playing;
      * play twice inside one cycle: ONE voice;
      * play(loop) over the playing loop in a later cycle: stop (tail) + restart;
      * an integer-beat clip played at playhead ~10 000 restarts on its beat (nextLoopTick = dispatch tick + 96 beats...)."""
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(41)
    lib = zo.load()
    N, bpm, block0 = 128, 200, 20_000
    zl.initJuce()
    try:
        host = _OracleHost(N)

        def setup(i, c, oc):
            beats = [1.0, 2.0, 0.37][i]
            zl.ClipAudioSource_setLength(c, beats, bpm); lib.zlo_clip_set_length(C.byref(oc), C.c_float(beats), bpm)
            zl.ClipAudioSource_setPan(c, 0.3 - 0.3 * i); lib.zlo_clip_set_pan(C.byref(oc), C.c_float(0.3 - 0.3 * i))
            zl.ClipAudioSource_setVolumeAbsolute(c, 0.8); lib.zlo_clip_set_volume_absolute(C.byref(oc), C.c_float(0.8))
        clips = _mk_clips(zl, host, rng, 3, setup)
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        pend = (zo.ClipCommand * 64)()
        npend = C.c_int32(0)

        def sched(cmds):
            for cm in (cmds if isinstance(cmds, list) else [cmds]):
                lib.zlo_step_schedule(pend, C.byref(npend), C.byref(cm))
        for k in range(300):
            if k == 1:      # play + stop in one cycle
                zl.ClipAudioSource_play(clips[0][0], True); sched(host.play(clips[0][1], True))
                zl.ClipAudioSource_stopOnChannel(clips[0][0], -2); sched(host.stop(clips[0][1], -2))
            if k == 3:      # play twice in one cycle
                zl.ClipAudioSource_playOnChannel(clips[1][0], True, 0); sched(host.play(clips[1][1], True, 0))
                zl.ClipAudioSource_playOnChannel(clips[1][0], True, 0); sched(host.play(clips[1][1], True, 0))
            if k == 40:     # play(loop) over the playing loop
                zl.ClipAudioSource_play(clips[0][0], True); sched(host.play(clips[0][1], True))
            if k == 60:     # a one-shot on a fractional-beat clip + a full stop of clip 1 in the same cycle
                zl.ClipAudioSource_playOnChannel(clips[2][0], False, 3); sched(host.play(clips[2][1], False, 3))
                zl.ClipAudioSource_stop(clips[1][0]); sched(host.stop(clips[1][1]))
            clk = synthetic_clocks(1, N, 48000.0, start_block=block0 + k, bpm=bpm, moving_playhead=True)
            tick = clk[0].jack_playhead
            assert tick > 10_000
            for i in range(npend.value):
                host.osyn.handle_clip_command(pend[i], tick)
            npend.value = 0
            assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
            bus = host.render(clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            if k == 2:
                assert host.voices_playing(clips[0][1]) == 1 and np.abs(outL[0]).max() > 0        # still playing after play + stop
            if k == 5:
                assert host.voices_playing(clips[1][1]) == 1                                     # one voice after play + play
            if k == 41:
                assert host.voices_playing(clips[0][1]) == 2                                     # the old voice tails off next to the new one
        # 1 beat at 200 bpm = 0.3 s = 112.5 blocks: the voice of clip 0 started at k = 40 with the playhead of that cycle as its start
        # tick has restarted on its beat twice by k = 300, each time one beat (96 ticks) after the last (SamplerSynthVoice.cpp:232-237)
        v = [x for x in host.osyn.voices if x.isPlaying and x.clip == clips[0][1]]
        assert len(v) == 1 and v[0].startTick > 10_000
        assert v[0].nextLoopTick - v[0].startTick == 3 * 96
        assert v[0].nextLoopTick > synthetic_clocks(1, N, 48000.0, start_block=block0 + 299, bpm=bpm, moving_playhead=True)[0].jack_playhead
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()


@pytest.mark.gpu
def test_the_librarys_own_transport_matches_the_oracle_sync_timer(zl):
    """libzl_hotpath_cycle: SyncTimerPrivate::process + every SamplerChannel per JACK cycle, against zlo_sync_timer_* + the oracle's
    SamplerSynth.  Paused timer first (delay 0 = the step behind the read head, the getters follow the read head), then
    SyncTimer_startTimer: the playhead counts steps, commands carry it, integer-beat loops restart against it; a bpm change through
    the SetBpm timer command; queueClipToStart (next bar) / queueClipToStop; SyncTimer_stopTimer."""
    from libzl_amd import Clock
    rng = np.random.default_rng(43)
    lib = zo.load()
    N, fs = 256, 48000.0
    per = int(round(1e6 * N / fs))
    zl.initJuce()
    try:
        host = _OracleHost(N)
        st = zo.OracleSyncTimer()
        st.set_latency(N, fs)

        def setup(i, c, oc):
            beats = [1.0, 0.41, 2.0, 1.0][i]
            zl.ClipAudioSource_setLength(c, beats, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(beats), 120)
            zl.ClipAudioSource_setVolumeAbsolute(c, 0.7); lib.zlo_clip_set_volume_absolute(C.byref(oc), C.c_float(0.7))
        clips = _mk_clips(zl, host, rng, 4, setup)
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        t0 = 3_000_017
        ndisp = 0
        from libzl_amd import libzl
        got_beats, want_beats = [], []
        tcb = libzl.TIMER_CB(lambda b: got_beats.append(b))               # SyncTimer_registerTimerCallback: the host sequencer's tick
        zl.SyncTimer_registerTimerCallback(tcb)
        for k in range(420):
            if k == 2:
                zl.ClipAudioSource_play(clips[0][0], True); st.schedule(host.play(clips[0][1], True), 0)
                zl.ClipAudioSource_stop(clips[0][0]); [st.schedule(c, 0) for c in host.stop(clips[0][1])]     # folds: keeps playing
            if k == 6:
                zl.SyncTimer_startTimer(120); st.start(120)
            if k == 8:
                zl.ClipAudioSource_playOnChannel(clips[1][0], True, 1); st.schedule(host.play(clips[1][1], True, 1), 0)
                zl.ClipAudioSource_playOnChannel(clips[2][0], True, 2); st.schedule(host.play(clips[2][1], True, 2), 0)
            if k == 30:
                zl.SyncTimer_queueClipToStartOnChannel(clips[3][0], 4); st.queue_start(clips[3][1], 4)        # at the next bar of the timer
            if k == 120:
                zl.SyncTimer_setBpm(174); st.set_bpm(174)
            if k == 250:
                zl.SyncTimer_queueClipToStopOnChannel(clips[3][0], 4); st.queue_stop(clips[3][1], 4)
                zl.ClipAudioSource_play(clips[0][0], True); st.schedule(host.play(clips[0][1], True), 0)
            if k == 60:                                               # one more tick of the timer thread (it has caught up already: no new beats)
                zl.libzl_hotpath_timer_tick(); st.timer_callback()
            if k == 300:                                              # the wrappers without a channel: channel -1 (SyncTimer.cpp:862-868)
                zl.SyncTimer_queueClipToStart(clips[1][0]); st.queue_start(clips[1][1], -1)
            if k == 322:
                zl.SyncTimer_queueClipToStop(clips[1][0]); st.queue_stop(clips[1][1], -1)
            if k == 330:
                zl.SyncTimer_stopTimer(); st.stop()
            if k == 400:                                              # libzl.h stopClips: ClipAudioSource::stop on each (libzl.cpp:87-94)
                arr = (C.c_void_p * 2)(clips[0][0], clips[2][0])
                zl.stopClips(2, arr)
                for _, oid in (clips[0], clips[2]):
                    [st.schedule(c, 0) for c in host.stop(oid)]
            cu, nx = t0 + k * per, t0 + (k + 1) * per
            for cm, tick in st.process(N, cu, nx):
                host.osyn.handle_clip_command(cm, tick)
                ndisp += 1
            oclk = st.clock(cu, nx)
            assert zl.libzl_hotpath_cycle(N, cu, nx, float(nx - cu), outL.ctypes.data, outR.ctypes.data) == 0
            got = Clock()
            assert zl.libzl_hotpath_transport(C.byref(got)) == 0
            assert (got.jack_playhead, got.jack_playhead_usecs, got.jack_subbeat_length_usecs) == (oclk.jackPlayhead, oclk.jackPlayheadUsecs, oclk.jackSubbeatLengthInMicroseconds), k
            bus = host.render([oclk])
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
            if not st.t.contents.threadPaused:
                b0, c0 = st.t.contents.beat, st.t.contents.cumulativeBeat
                st.timer_callback()
                want_beats += [(b0 + i) % 384 for i in range(st.t.contents.cumulativeBeat - c0)]      # callbacks[i](beat) per tick, SyncTimer.cpp:395-405
            if k == 200:
                zl.SyncTimer_deregisterTimerCallback(tcb)
                n200 = len(want_beats)
        assert ndisp >= 6
        assert got_beats == want_beats[:n200] and len(got_beats) > 150
        st.close()
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()


@pytest.mark.gpu
def test_callbacks_may_call_back_into_the_api_and_setters_do_not_evict_the_resident_kernel(zl):
    """the progress / level callbacks fire after the cycle's lock is released -- one that reads peakGain, looks a clip up
    by id and turns a knob must not deadlock.  VERDICT r2 item 4: parameter setters while the engine runs real-time cycles leave
    the resident kernel resident (one launch for the whole session) and land on the next cycle, bit-exact against the oracle."""
    from libzl_amd import libzl, _abi
    from libzl_amd.engine import synthetic_clocks
    rng = np.random.default_rng(47)
    lib = zo.load()
    N = 128
    now = [5_000_000]
    clock_cb = libzl.CLOCK_MS(lambda: now[0])
    zl.libzl_hotpath_set_clock_ms(clock_cb)
    zl.initJuce()
    try:
        host = _OracleHost(N)

        def setup(i, c, oc):
            zl.ClipAudioSource_setLength(c, 0.33, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(0.33), 120)
        clips = _mk_clips(zl, host, rng, 2, setup)
        seen = []

        def on_progress(sec):
            c = zl.ClipAudioSource_byID(zl.ClipAudioSource_id(clips[0][0]))          # takes the cycle's mutex
            seen.append((zl.ClipAudioSource_peakGain(c), sec))
            zl.ClipAudioSource_setKeyZoneEnd(c, 99)
        pg = libzl.CB(on_progress)
        zl.ClipAudioSource_setProgressCallback(clips[0][0], pg)
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        eng = zl.libzl_hotpath_engine()
        starts, cyc = C.c_uint64(), C.c_uint64()
        for k in range(200):
            now[0] += 3
            if k == 1:
                for c, oid in clips:
                    zl.ClipAudioSource_play(c, True); host.osyn.handle_clip_command(host.play(oid, True), 0)
            if k >= 10 and k % 3 == 0:                 # a pan / volume knob turned every third cycle while both loops play
                pan, vol = float(np.float32(np.sin(k * 0.1))), float(np.float32(0.5 + 0.4 * np.cos(k * 0.07)))
                c, oid = clips[(k // 3) % 2]
                zl.ClipAudioSource_setPan(c, pan); lib.zlo_clip_set_pan(C.byref(host.osyn.clips[oid]), C.c_float(pan))
                zl.ClipAudioSource_setVolumeAbsolute(c, vol); lib.zlo_clip_set_volume_absolute(C.byref(host.osyn.clips[oid]), C.c_float(vol))
            if k == 100:                               # a loop-length edit lands on the next block too
                zl.ClipAudioSource_setLength(clips[1][0], 0.21, 120); lib.zlo_clip_set_length(C.byref(host.osyn.clips[clips[1][1]]), C.c_float(0.21), 120)
            clk = synthetic_clocks(1, N, 48000.0, start_block=k)
            assert zl.libzl_hotpath_process(N, clk, outL.ctypes.data, outR.ctypes.data) == 0
            bus = host.render(clk)
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), k
        assert len(seen) >= 3 and all(g >= 0.0 for g, _ in seen) and zl.ClipAudioSource_keyZoneEnd(clips[0][0]) == 99
        _abi.bind(zl)
        assert zl.zlhip_rt_stats(eng, C.byref(starts), C.byref(cyc)) == 0
        assert cyc.value == 200 and starts.value == 1, (starts.value, cyc.value)       # ~130 setter calls, one launch
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()
        zl.libzl_hotpath_set_clock_ms(libzl.CLOCK_MS())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [61, 62, 63])
def test_random_sessions_through_the_libzl_names(zl, seed):
    """A seeded session of everything a host does through the libzl names while audio runs -- play / stop (also twice in a cycle),
    playOnChannel / stopOnChannel, SyncTimer_startTimer / _setBpm / _stopTimer / _queueClipToStart / _queueClipToStop, and the setters
    (length in beats, pan, volume, start position, slices, root note, ADSR attack / release) -- against the oracle's SyncTimer + ClipAudioSource
    setters + SamplerSynth, cycle by cycle: audio bit for bit, the transport triple, the positions-model read-outs."""
    from libzl_amd import Clock
    rng = np.random.default_rng(seed)
    lib = zo.load()
    N = int(rng.choice([64, 128, 256])); fs = 48000.0
    per = int(round(1e6 * N / fs))
    zl.initJuce()
    try:
        host = _OracleHost(N)
        st = zo.OracleSyncTimer()
        st.set_latency(N, fs)
        nclips = 6

        def setup(i, c, oc):
            beats = float(rng.choice([1.0, 2.0, 0.37, 0.61]))
            zl.ClipAudioSource_setLength(c, beats, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(beats), 120)
        clips = _mk_clips(zl, host, rng, nclips, setup)
        outL = np.zeros((12, N), dtype=np.float32); outR = np.zeros((12, N), dtype=np.float32)
        t0 = int(rng.integers(1, 5)) * 1_000_003
        ndisp = nsound = 0
        for k in range(260):
            for _ in range(int(rng.integers(0, 4)) if rng.random() < 0.45 else 0):
                i = int(rng.integers(0, nclips)); c, oid = clips[i]; oc = host.osyn.clips[oid]
                a = int(rng.integers(0, 16))
                if a < 4:
                    loop = bool(rng.integers(0, 2)); ch = int(rng.choice([-2, -2, -1, 0, 3]))
                    if ch == -2: zl.ClipAudioSource_play(c, loop)
                    else: zl.ClipAudioSource_playOnChannel(c, loop, ch)
                    st.schedule(host.play(oid, loop, ch), 0)
                elif a < 6:
                    if rng.random() < 0.5:
                        zl.ClipAudioSource_stop(c); [st.schedule(x, 0) for x in host.stop(oid)]
                    else:
                        ch = int(rng.choice([-2, -1, 0, 3])); zl.ClipAudioSource_stopOnChannel(c, ch); [st.schedule(x, 0) for x in host.stop(oid, ch)]
                elif a == 6:
                    b = int(rng.choice([90, 120, 174])); zl.SyncTimer_startTimer(b); st.start(b)
                elif a == 7:
                    zl.SyncTimer_stopTimer(); st.stop()
                elif a == 8:
                    b = int(rng.choice([60, 120, 150, 240])); zl.SyncTimer_setBpm(b); st.set_bpm(b)
                elif a == 9:
                    ch = int(rng.choice([-1, 2])); zl.SyncTimer_queueClipToStartOnChannel(c, ch); st.queue_start(oid, ch)
                elif a == 10:
                    ch = int(rng.choice([-1, 2])); zl.SyncTimer_queueClipToStopOnChannel(c, ch); st.queue_stop(oid, ch)
                elif a == 11:
                    v = float(np.float32(rng.uniform(-1, 1))); zl.ClipAudioSource_setPan(c, v); lib.zlo_clip_set_pan(C.byref(oc), C.c_float(v))
                elif a == 12:
                    v = float(np.float32(rng.uniform(0, 1.2))); zl.ClipAudioSource_setVolumeAbsolute(c, v); lib.zlo_clip_set_volume_absolute(C.byref(oc), C.c_float(v))
                elif a == 13:
                    beats = float(rng.choice([1.0, 0.29, 0.53, 2.0])); zl.ClipAudioSource_setLength(c, beats, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(beats), 120)
                elif a == 14:
                    v = float(np.float32(rng.uniform(0, 0.02))); zl.ClipAudioSource_setStartPosition(c, v); lib.zlo_clip_set_start_position(C.byref(oc), C.c_float(v))
                else:
                    if rng.random() < 0.5:
                        v = float(np.float32(rng.choice([0.0, 0.003, 0.02]))); zl.ClipAudioSource_setADSRAttack(c, v); lib.zlo_clip_set_adsr_attack(C.byref(oc), C.c_float(v))
                    else:
                        v = float(np.float32(rng.choice([0.0, 0.01, 0.08]))); zl.ClipAudioSource_setADSRRelease(c, v); lib.zlo_clip_set_adsr_release(C.byref(oc), C.c_float(v))
            cu, nx = t0 + k * per, t0 + (k + 1) * per
            for cm, tick in st.process(N, cu, nx):
                host.osyn.handle_clip_command(cm, tick)
                ndisp += 1
            oclk = st.clock(cu, nx)
            assert zl.libzl_hotpath_cycle(N, cu, nx, float(nx - cu), outL.ctypes.data, outR.ctypes.data) == 0
            got = Clock()
            zl.libzl_hotpath_transport(C.byref(got))
            assert (got.jack_playhead, got.jack_playhead_usecs, got.jack_subbeat_length_usecs) == (oclk.jackPlayhead, oclk.jackPlayheadUsecs, oclk.jackSubbeatLengthInMicroseconds), k
            bus = host.render([oclk])
            assert np.array_equal(outL.view(np.int32), bus[:, 0].view(np.int32)) and np.array_equal(outR.view(np.int32), bus[:, 1].view(np.int32)), (seed, k)
            nsound += int(np.abs(bus).max() > 0)
            if not st.t.contents.threadPaused:
                st.timer_callback()
        assert ndisp > 20 and nsound > 100, (ndisp, nsound)
        st.close()
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [16, 32])
def test_offline_bounce_of_a_session_to_wav_files(zl, tmp_path, bits):
    """SURVEY 8f n3, BASELINE configs[4] 'from / to real files': clips loaded from WAV files, a running SyncTimer, commands that fall
    due in the middle of the bounce (one scheduled 40 steps ahead, a queued bar-aligned start), then libzl_hotpath_bounce_to_wav --
    one stereo WAV per sampler channel.  Every file equals the oracle's cycle-by-cycle rendering of the same session: as floats bit
    for bit, as 16 bit through the oracle's recorder conversion (zlo_pcm16_stereo)."""
    from libzl_amd import _abi
    from scenario import engine_cmd
    rng = np.random.default_rng(71)
    lib = zo.load()
    N, fs, K = 128, 48000.0, 800        # 2.13 s: the bar-aligned start (384 steps of 5208 us ahead) falls inside
    per = int(round(1e6 * N / fs))
    cfg = _abi.Config()
    _abi.bind(zl)
    zl.zlhip_config_default(C.byref(cfg))
    cfg.max_batch_blocks = 64; cfg.max_frames = 256
    zl.libzl_hotpath_configure(C.byref(cfg))
    zl.initJuce()
    try:
        host = _OracleHost(N)
        st = zo.OracleSyncTimer()
        st.set_latency(N, fs)
        clips = []
        for i in range(4):
            ln = 6000 + 700 * i
            # 16-bit files: what a host loads; the oracle gets the decoded planes
            L = (rng.uniform(-1, 1, ln) * 0.9).astype(np.float32); R = (rng.uniform(-1, 1, ln) * 0.9).astype(np.float32) if i % 2 == 0 else None
            path = _wav(tmp_path, zl, L, R, fs, 16, f"src{i}.wav")
            dL, dR, _ = _read(zl, path)
            c = zl.ClipAudioSource_new(path, False)
            assert c
            oid = host.osyn.register_clip(dL, dR, fs)
            oc = host.osyn.clips[oid]
            beats = [1.0, 0.43, 2.0, 0.31][i]
            zl.ClipAudioSource_setLength(c, beats, 120); lib.zlo_clip_set_length(C.byref(oc), C.c_float(beats), 120)
            zl.ClipAudioSource_setPan(c, 0.2 * i - 0.3); lib.zlo_clip_set_pan(C.byref(oc), C.c_float(0.2 * i - 0.3))
            clips.append((c, oid))
        t0 = 9_000_011
        zl.SyncTimer_startTimer(120); st.start(120)
        zl.ClipAudioSource_play(clips[0][0], True); st.schedule(host.play(clips[0][1], True), 0)
        zl.ClipAudioSource_playOnChannel(clips[1][0], True, 1); st.schedule(host.play(clips[1][1], True, 1), 0)
        late = host.play(clips[2][1], False, 2)                                       # a one-shot 40 steps from now
        zl.libzl_hotpath_schedule_clip_command(C.byref(engine_cmd(**{f: getattr(late, f) for f in zo.CMD_FIELDS})), 40); st.schedule(late, 40)
        zl.SyncTimer_queueClipToStartOnChannel(clips[3][0], 0); st.queue_start(clips[3][1], 0)      # at the next bar
        prefix = str(tmp_path / f"bounce{bits}").encode()
        assert zl.libzl_hotpath_bounce_to_wav(prefix, K, N, t0, bits) == 0
        want = np.zeros((12, 2, K * N), dtype=np.float32)
        ndisp = 0
        for k in range(K):
            cu, nx = t0 + k * per, t0 + (k + 1) * per
            for cm, tick in st.process(N, cu, nx):
                host.osyn.handle_clip_command(cm, tick)
                ndisp += 1
            want[:, :, k * N:(k + 1) * N] = host.render([st.clock(cu, nx)])
            if not st.t.contents.threadPaused:
                st.timer_callback()
        assert ndisp == 4
        heard = 0
        for b in range(12):
            raw = open(prefix.decode() + f"-channel_{b}.wav", "rb").read()
            assert raw[:4] == b"RIFF" and len(raw) == 44 + K * N * 2 * (2 if bits == 16 else 4)
            if bits == 16:
                got = np.frombuffer(raw[44:], dtype=np.int16).reshape(-1, 2)
                exp = np.empty((K * N, 2), dtype=np.int16)
                Lr, Rr = np.ascontiguousarray(want[b, 0]), np.ascontiguousarray(want[b, 1])
                lib.zlo_pcm16_stereo(Lr.ctypes.data, Rr.ctypes.data, K * N, exp.ctypes.data)
                assert np.array_equal(got, exp), b
            else:
                got = np.frombuffer(raw[44:], dtype=np.float32).reshape(-1, 2)
                assert np.array_equal(got[:, 0].view(np.int32), want[b, 0].view(np.int32)) and np.array_equal(got[:, 1].view(np.int32), want[b, 1].view(np.int32)), b
            heard += int(np.abs(want[b]).max() > 0)
        assert heard == 4                                                             # channels -2, 1, 2, 0 carry sound
        st.close()
        for c, _ in clips:
            zl.ClipAudioSource_destroy(c)
    finally:
        zl.shutdownJuce()
        zl.libzl_hotpath_configure(None)
