"""Known-answer tests of the CPU oracle, derived by hand from the reference source text
(/root/reference/lib/SamplerSynthVoice.cpp:174-270 and friends).
The reference itself holds no golden vectors ("parity unpinned"); these pin the restatement."""
import ctypes as C

import numpy as np
import pytest

from oracle import zl_oracle as zo
from scenario import Scene, play_cmd, run_oracle, oracle_trace
from libzl_amd.engine import synthetic_clocks

f32 = np.float32


@pytest.fixture(scope="module")
def lib(built):
    return zo.load()


def one_voice_scene(L, R, sr=48000.0, fs=48000.0, *, beats=0.37, volume=1.0, pan=0.0, note=60, loop=True, nframes=64, nblocks=4,
                    cmd_volume=1.0, adsr=None, mode=0):
    sc = Scene(num_buses=1, voices_per_bus=1, fs=fs, nframes=nframes, nblocks=nblocks, mode=mode)
    sc.sounds.append((np.asarray(L, dtype=np.float32), None if R is None else np.asarray(R, dtype=np.float32), sr))

    def setup(lib, clip):
        lib.zlo_clip_set_length(clip, C.c_float(beats), 120)
        lib.zlo_clip_set_volume_absolute(clip, C.c_float(volume))
        lib.zlo_clip_set_pan(clip, C.c_float(pan))
        if adsr is not None:
            clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = adsr
    sc.clip_setup[0] = setup
    sc.events[0] = [("cmd", play_cmd(0, loop=loop, note=note, volume=cmd_volume), 0)]
    return sc


def test_kat_i_alpha_zero_raw_source_delayed_one_frame(lib):
    """(i) ratio 1 => alpha = 0 => only the un-gained first tap survives (Q1); Q2 delays it by one frame and drops
    the last frame of every block; M/S pan at pan = 0 of a mono source gives 0.5 * x on both channels (Q3/Q4)."""
    x = np.linspace(-0.9, 0.9, 400, dtype=np.float32)
    sc = one_voice_scene(x, None, cmd_volume=0.25, volume=0.5, nframes=64, nblocks=2, beats=3.7)
    bus, rep, _ = run_oracle(sc)
    L, R = bus[0, 0], bus[0, 1]
    for k in range(2):
        assert L[k * 64] == 0.0 and R[k * 64] == 0.0                       # out[0] stays 0
        exp = (f32(0.5) * (x[k * 64:k * 64 + 63] + x[k * 64:k * 64 + 63])) * f32(0.5)   # lPan * m + s with r = l: 0.5 * (0.5 * 2x) + 0
        np.testing.assert_array_equal(L[k * 64 + 1:k * 64 + 64], f32(0.5) * x[k * 64:k * 64 + 63])
        np.testing.assert_array_equal(R[k * 64 + 1:k * 64 + 64], f32(0.5) * x[k * 64:k * 64 + 63])
        del exp


def test_kat_ii_mono_pan_law(lib):
    """(ii) mono source: l' = lPan * l, r' = rPan * l with lPan = 0.5(1+pan), rPan = 0.5(1-pan) (s = 0)."""
    x = np.full(300, 0.5, dtype=np.float32)
    for pan in (-1.0, -0.25, 0.0, 0.5, 1.0):
        sc = one_voice_scene(x, None, pan=pan, nframes=64, nblocks=1, beats=3.7)
        bus, _, _ = run_oracle(sc)
        lp = f32(0.5 * (1.0 + float(f32(pan))))
        rp = f32(0.5 * (1.0 - float(f32(pan))))
        assert bus[0, 0, 5] == lp * f32(0.5) and bus[0, 1, 5] == rp * f32(0.5)


def test_kat_iii_half_speed_alternating_alpha(lib):
    """(iii) ratio 0.5 (one octave down): positions 0, 0.5, 1, 1.5 ... so pos repeats and alpha alternates 0 / 0.5."""
    x = np.arange(400, dtype=np.float32) / f32(512)
    sc = one_voice_scene(x, None, note=48, nframes=64, nblocks=1, beats=3.7)
    tr, _ = oracle_trace(sc)
    np.testing.assert_array_equal(tr[0, 0], np.arange(64) // 2)
    bus, _, _ = run_oracle(sc)
    # frame f=3: pos 1, alpha .5: l = x1*0.5 + x2*0.5*gain(1)*env(1)*vol(1); out index 4; mono pan 0 -> *0.5
    exp = f32(0.5) * (x[1] * f32(0.5) + x[2] * f32(0.5))
    assert bus[0, 0, 4] == exp


def test_kat_vii_hermite_reproduces_quadratics(lib):
    """Build-defined Hermite mode: a Catmull-Rom cubic through samples of a quadratic IS that quadratic (c1 = 2n, c2 = 1,
    c3 = 0 for x[n] = n^2), and every intermediate of the fused evaluation is exact for small integers and alpha = 1/2:
    one octave down the output is ((n + alpha)^2) / 4096, delayed one frame (Q2), times 0.5 (mono, pan 0).  At pos = 0
    the tap pos - 1 does not exist and the frame falls back to the linear form (un-fused)."""
    n = np.arange(200, dtype=np.float64)
    x = (n * n / 4096.0).astype(np.float32)                      # exact in fp32
    sc = one_voice_scene(x, None, note=48, nframes=64, nblocks=1, beats=3.7, mode=4)
    bus, _, _ = run_oracle(sc)
    for f in range(2, 63):
        pos, a = f // 2, 0.5 * (f % 2)
        exp = f32(0.5) * f32((pos + a) ** 2 / 4096.0)
        assert bus[0, 0, f + 1] == exp and bus[0, 1, f + 1] == exp, f
    lin = f32(x[0] * f32(0.5)) + f32(x[1] * f32(0.5))            # f = 1: pos 0, alpha 1/2, linear fallback
    assert bus[0, 0, 2] == f32(0.5) * lin


def test_kat_iv_fractional_beat_wrap_index_exact(lib):
    """(iv) non-integer lengthInBeats: wrap when P >= stopPosition, P restarts at the exact integer start (Q9b)."""
    x = np.random.default_rng(1).uniform(-1, 1, 3000).astype(np.float32)
    sc = one_voice_scene(x, None, nframes=64, nblocks=8, beats=0.01)      # 0.01 beat = 0 subbeats?  use explicit fields below

    def setup(lib, clip):
        clip.lengthInBeats = 0.5                 # fractional -> sample-space branch
        clip.lengthInSeconds = f32(100.25 / 48000.0)
        clip.startPositionInSeconds = f32(17.0 / 48000.0)
    sc.clip_setup[0] = setup
    tr, osyn = oracle_trace(sc)
    start = int(float(f32(17.0 / 48000.0)) * 48000.0)
    stop = int(float(f32(f32(17.0 / 48000.0) + f32(100.25 / 48000.0))) * 48000.0)
    flat = tr[:, 0, :].reshape(-1)
    assert flat[0] == start
    period = stop - start
    np.testing.assert_array_equal(flat, start + (np.arange(flat.size) % period))


def test_kat_v_oneshot_release_tail_is_geometric(lib):
    """(v) one-shot: from the first frame with P >= stop - R*sr the reference calls noteOff EVERY frame (Q7), so the
    envelope decays by the factor (1 - 1/(R*sr)) per frame; the voice ends at the first P >= stop."""
    n = 2000
    x = np.ones(n, dtype=np.float32)
    R_, sr = 0.01, 48000.0
    sc = one_voice_scene(x, None, loop=False, nframes=64, nblocks=12, adsr=(0.0, 0.1, 1.0, R_), beats=3.7)

    def setup(lib, clip):
        clip.lengthInBeats = 0.5
        clip.lengthInSeconds = f32(600.0 / sr)
        clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = (0.0, 0.1, 1.0, R_)
    sc.clip_setup[0] = setup
    # render voice alone block by block and watch the envelope through the output (x = 1, alpha = 0 => first tap only,
    # so use ratio 0.5 to expose the envelope on the second tap)
    sc.events[0] = [("cmd", play_cmd(0, loop=False, note=48), 0)]
    osyn = zo.OracleSynth(1, 1, 48000.0, 0)
    osyn.register_clip(x, None, sr)
    setup(osyn.lib, osyn.clips[0])
    osyn.handle_clip_command(zo.clip_command(**play_cmd(0, loop=False, note=48)))
    v = osyn.voices[0]
    stop = int(float(f32(600.0 / sr)) * sr)
    tail_start = stop - float(f32(R_)) * sr
    envs, poss = [], []
    clk = zo.Clock(0, 1333, 0, 0, 5208)
    rep = zo.Report()
    while v.isPlaying:
        Lb = np.zeros(1, dtype=np.float32); Rb = np.zeros(1, dtype=np.float32)
        poss.append(v.sourceSamplePosition)
        osyn.lib.zlo_voice_process(C.byref(v), Lb.ctypes.data, Rb.ctypes.data, 1, C.byref(clk), osyn.sounds, osyn.clips, 0, 0, C.byref(rep), None)
        envs.append(v.adsr.env)
        assert len(envs) < 5000
    poss = np.array(poss); envs = np.array(envs, dtype=np.float32)
    # voice ends right after the first frame whose incremented position reaches stop
    assert poss[-1] + 0.5 >= stop > poss[-1]
    first_tail = int(np.argmax(poss + 0.5 >= tail_start))      # frame after which noteOff is first called
    q = envs[first_tail + 2:-1] / envs[first_tail + 1:-2]
    np.testing.assert_allclose(q, 1.0 - 1.0 / (float(f32(R_)) * sr), rtol=2e-6)
    assert np.all(envs[:first_tail + 1] == 1.0)


def test_kat_vi_audio_levels(lib):
    """(vi) x = 1.0 -> peak 131072 -> 131072 * 1.52587e-6 = 0.2 -> -13.98 dBFS (Q12); a silent tick decays by 10000."""
    ch = zo.LevelsChannel()
    L = np.zeros(64, dtype=np.float32); R = np.zeros(64, dtype=np.float32)
    L[7] = 1.0; R[9] = -0.5
    lib.zlo_levels_tick(C.byref(ch), L.ctypes.data, R.ctypes.data, 64, 1)
    assert (ch.peakA, ch.peakB) == (131072, 65536)
    assert abs(ch.peakDbA - 20 * np.log10(131072 * 1.52587e-6)) < 1e-4 and abs(ch.peakDbA + 13.98) < 0.01
    hold = ch.peakAHoldSignal
    lib.zlo_levels_tick(C.byref(ch), None, None, 0, 1)
    assert (ch.peakA, ch.peakB) == (121072, 55536)
    assert ch.peakAHoldSignal == f32(hold * f32(0.9))
    assert lib.zlo_convert_to_dbfs(C.c_float(0.0)) == -200.0
    assert lib.zlo_sample_to_peak_int(C.c_float(-0.9999999)) == int(abs(f32(131072.0) * f32(-0.9999999)))


def test_adsr_states(lib):
    a = zo.Adsr()
    lib.zlo_adsr_init(C.byref(a))
    p = zo.AdsrParams(0.001, 0.002, 0.5, 0.001)
    lib.zlo_adsr_set_sample_rate(C.byref(a), 10000.0)
    lib.zlo_adsr_set_parameters(C.byref(a), C.byref(p))
    lib.zlo_adsr_note_on(C.byref(a))
    seq = [lib.zlo_adsr_next(C.byref(a)) for _ in range(40)]
    assert seq[9] == 1.0 and a.state == 3            # 10 attack steps of 0.1, 20 decay steps of 0.025 -> sustain
    assert abs(seq[10] - 0.975) < 1e-6 and seq[-1] == 0.5
    lib.zlo_adsr_note_off(C.byref(a))
    rel = [lib.zlo_adsr_next(C.byref(a)) for _ in range(12)]
    assert rel[-1] == 0.0 and not lib.zlo_adsr_is_active(C.byref(a))
    # default clip ADSR (attack 0, decay .1, sustain 1, release .05): decay distance 0 -> straight to sustain, first sample 1.0
    c = zo.Clip()
    lib.zlo_clip_init(C.byref(c), C.c_float(1.0), 48000.0)
    b = zo.Adsr(); lib.zlo_adsr_init(C.byref(b)); lib.zlo_adsr_set_sample_rate(C.byref(b), 48000.0)
    lib.zlo_adsr_set_parameters(C.byref(b), C.byref(c.adsr.p)); lib.zlo_adsr_note_on(C.byref(b))
    assert lib.zlo_adsr_next(C.byref(b)) == 1.0 and b.state == 3


def test_clip_setters(lib):
    c = zo.Clip()
    lib.zlo_clip_init(C.byref(c), C.c_float(2.0), 48000.0)
    assert c.nSlicePositions == 16 and c.slicePositions[4] == 0.25 and c.lengthInBeats == -1.0
    lib.zlo_clip_set_length(C.byref(c), C.c_float(4.0), 120)
    assert c.lengthInSeconds == f32(2.0) and c.lengthInBeats == 4.0       # 4 beats at 120 bpm
    lib.zlo_clip_set_length(C.byref(c), C.c_float(4.0), 20)               # bpm clamps to 50
    assert c.lengthInSeconds == f32(f32((384 * 60000000000) // (50 * 96)) / f32(1e9))
    # Q13: each ADSR setter starts from juce defaults (attack .1, decay .1, sustain 1, release .1)
    lib.zlo_clip_set_adsr_release(C.byref(c), C.c_float(0.3))
    assert (c.adsr.p.attack, c.adsr.p.release) == (f32(0.1), f32(0.3))
    lib.zlo_clip_set_adsr_attack(C.byref(c), C.c_float(0.0))
    assert (c.adsr.p.attack, c.adsr.p.release) == (0.0, f32(0.1))
    assert lib.zlo_clip_get_stop_position(C.byref(c), 3) == f32(0.0 + float(c.lengthInSeconds) * 0.25)
    assert lib.zlo_clip_slice_for_midi_note(C.byref(c), 62) == 2
    lib.zlo_clip_set_volume_absolute(C.byref(c), C.c_float(1.7)); assert c.volumeAbsolute == 1.0
    lib.zlo_clip_set_start_position(C.byref(c), C.c_float(-3.0)); assert c.startPositionInSeconds == 0.0


def test_positions_model(lib):
    m = zo.Positions(); lib.zlo_positions_init(C.byref(m))
    ids = [lib.zlo_positions_create(C.byref(m), C.c_float(0.0), 1000) for _ in range(3)]
    assert ids == [0, 1, 2]
    lib.zlo_positions_set_gain_and_progress(C.byref(m), 1, C.c_float(0.5), C.c_float(0.25), 1000)
    assert lib.zlo_positions_peak_gain(C.byref(m)) == 0.5
    lib.zlo_positions_set_gain_and_progress(C.byref(m), 1, C.c_float(0.505), C.c_float(0.3), 1000)
    assert lib.zlo_positions_peak_gain(C.byref(m)) == 0.5                  # 0.01 hysteresis
    lib.zlo_positions_remove(C.byref(m), 0, 1000)
    assert lib.zlo_positions_create(C.byref(m), C.c_float(0.0), 1000) == 0 # first free row
    assert lib.zlo_positions_cleanup(C.byref(m), 5000) == 3                # everything older than 1 s is swept


def test_passthrough(lib):
    rng = np.random.default_rng(3)
    inL = rng.uniform(-1, 1, 32).astype(np.float32); inR = rng.uniform(-1, 1, 32).astype(np.float32)
    outs = [np.full(32, 9.0, dtype=np.float32) for _ in range(6)]
    arr = (C.c_void_p * 6)(*[o.ctypes.data for o in outs])
    p = zo.Passthrough(1.0, 0.0, 0.5, 0.0, 0)
    lib.zlo_passthrough_process(C.byref(p), inL.ctypes.data, inR.ctypes.data, arr, 32)
    np.testing.assert_array_equal(outs[0], inL); np.testing.assert_array_equal(outs[2], 0 * inL)
    np.testing.assert_array_equal(outs[4], f32(0.5) * inL * f32(1.0))
    p = zo.Passthrough(1.0, 1.0, 1.0, 0.25, 0)
    lib.zlo_passthrough_process(C.byref(p), inL.ctypes.data, inR.ctypes.data, arr, 32)
    np.testing.assert_array_equal(outs[0], f32(1.0) * inL * f32(0.75)); np.testing.assert_array_equal(outs[1], inR)
    p.muted = 1
    lib.zlo_passthrough_process(C.byref(p), inL.ctypes.data, inR.ctypes.data, arr, 32)
    assert all(not o.any() for o in outs)


def test_numpy_restatement_fma_is_one_rounding():
    """oracle/np_restatement.fma32 (the Hermite mode's fused multiply-add in the golden generator) against exact
    rational arithmetic: random operands, exact cancellation, near-ties."""
    from fractions import Fraction
    from oracle.np_restatement import fma32
    rng = np.random.default_rng(5)

    def exact32(a, b, c):
        x = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
        if x == 0:
            return f32(float(a) * float(b) + float(c))
        g = f32(float(x))
        cands = [g, np.nextafter(g, f32(np.inf)), np.nextafter(g, f32(-np.inf))]
        return min(cands, key=lambda t: (abs(Fraction(float(t)) - x), int(f32(t).view(np.int32)) & 1))     # nearest, ties to even
    for i in range(20000):
        k = i % 4
        if k == 0:
            a, b, c = (f32(v) for v in rng.uniform(-1, 1, 3))
        elif k == 1:
            a, b = f32(rng.uniform(-2, 2)), f32(rng.uniform(-2, 2)); c = f32(-float(a) * float(b))
        elif k == 2:
            a, b = f32(1 + 2.0 ** -int(rng.integers(1, 24))), f32(1 + 2.0 ** -int(rng.integers(1, 24)))
            c = f32(float(rng.choice([1, -1])) * 2.0 ** -int(rng.integers(0, 30)))
        else:
            a = f32(rng.uniform(-1, 1) * 2.0 ** int(rng.integers(-20, 20))); b = f32(rng.uniform(-1, 1))
            c = f32(rng.uniform(-1, 1) * 2.0 ** int(rng.integers(-30, 10)))
        assert fma32(a, b, c).view(np.int32) == exact32(a, b, c).view(np.int32), (a, b, c)


def test_hermite_tap_weight_form_stays_within_rounding_of_the_horner_form(lib):
    """ZLHIP_MODE_HERMITE was redefined in round 2 (ABI 2) from the Horner evaluation of the Catmull-Rom cubic to the
    tap-weight form with fused multiply-adds and the gain product formed first.  Both are roundings of the same polynomial: on random
    material, pitched, the oracle's output stays within a few fp32 ulps of the Horner form evaluated here in fp32 (JUCE's
    CatmullRomInterpolator shape), and equals the float64 cubic to ~1e-7 -- so a later change of form cannot drift
    silently past rounding."""
    rng = np.random.default_rng(0x4E4D)
    x = rng.uniform(-1, 1, 5000).astype(np.float32)
    sc = one_voice_scene(x, None, note=53, nframes=64, nblocks=6, beats=3.7, mode=4)
    tr, _ = oracle_trace(sc)
    bus, _, _ = run_oracle(sc)
    ratio = 2.0 ** ((53 - 60) / 12.0)
    P = 0.0
    worst_h, worst_64 = 0.0, 0.0
    for k in range(6):
        for f in range(64):
            pos = int(P)
            assert tr[k, 0, f] == pos
            a = f32(P - pos)
            if pos >= 1 and f + 1 < 64:
                y0, y1, y2, y3 = (x[pos - 1], x[pos], x[pos + 1], x[pos + 2])
                # Horner form in fp32: y1 + a (c1 + a (c2 + a c3))
                c1 = f32(0.5) * (y2 - y0)
                c2 = (y0 + f32(2.0) * y2) - (f32(0.5) * y3 + f32(2.5) * y1)
                c3 = (f32(0.5) * y3 + f32(1.5) * y1) - (f32(0.5) * y0 + f32(1.5) * y2)
                horner = f32(y1 + a * f32(c1 + a * f32(c2 + a * c3)))
                a64 = float(a)
                d = [float(v) for v in (y0, y1, y2, y3)]
                exact = d[1] + a64 * (0.5 * (d[2] - d[0]) + a64 * ((d[0] + 2 * d[2] - 0.5 * d[3] - 2.5 * d[1]) + a64 * (0.5 * d[3] + 1.5 * d[1] - 0.5 * d[0] - 1.5 * d[2])))
                got = float(bus[0, 0, k * 64 + f + 1]) * 2.0          # mono, pan 0: the output is 0.5 x the sample; gain, envelope, volume are 1
                worst_h = max(worst_h, abs(got - float(horner)))
                worst_64 = max(worst_64, abs(got - exact))
            P += ratio
    assert 0.0 < worst_h < 6e-7 and worst_64 < 4e-7, (worst_h, worst_64)
