"""The two restatements of the reference path -- oracle/zl_oracle.c and the independently written oracle/np_restatement.py -- against
each other on RANDOM scenes (tests/scenario.random_scene: looping, beat-locked and one-shot clips, mono / stereo, pitched and resampled,
envelopes, slices, stop / patch / retrigger commands and clip edits between blocks), beyond the committed golden scenes: audio, the
voices' playing flags and positions and the last block's reports, bit for bit, in the faithful, the fixed and the Hermite mode.
CPU tier; the scenes are small because the numpy twin steps frame by frame in Python."""
import ctypes as C

import numpy as np
import pytest

from oracle import np_restatement as nr
from scenario import random_scene, run_oracle

f32 = np.float32

_CMD = {"clip": "clip", "midiNote": "midi_note", "midiChannel": "midi_channel", "startPlayback": "start", "stopPlayback": "stop",
        "changeSlice": "change_slice", "slice": "slice", "changeLooping": "change_looping", "looping": "looping",
        "changePitch": "change_pitch", "pitchChange": "pitch_change", "changeSpeed": "change_speed", "speedRatio": "speed_ratio",
        "changeGainDb": "change_gain_db", "gainDb": "gain_db", "changeVolume": "change_volume", "volume": "volume"}
_F32 = {"volume", "pitch_change", "speed_ratio", "gain_db"}
_BOOL = {"start", "stop", "change_slice", "change_looping", "looping", "change_pitch", "change_speed", "change_gain_db", "change_volume"}


def _np_command(fields):
    kw = {}
    for k, v in fields.items():
        a = _CMD[k]
        kw[a] = f32(v) if a in _F32 else bool(v) if a in _BOOL else int(v)
    return nr.Command(**kw)


def _copy_clip(oc, c: nr.Clip):
    """the fields the voice reads, from the oracle's clip struct (whose restated setters configured it) to the numpy clip"""
    c.start_sec = f32(oc.startPositionInSeconds); c.length_sec = f32(oc.lengthInSeconds); c.length_beats = f32(oc.lengthInBeats)
    c.volume_abs = f32(oc.volumeAbsolute); c.pan = f32(oc.pan); c.duration = f32(oc.duration); c.root_note = int(oc.rootNote)
    c.slice_pos = [float(oc.slicePositions[i]) for i in range(oc.nSlicePositions)]
    c.adsr = (f32(oc.adsr.p.attack), f32(oc.adsr.p.decay), f32(oc.adsr.p.sustain), f32(oc.adsr.p.release))


def run_numpy(scene):
    from oracle import zl_oracle as zo
    ref = zo.OracleSynth(1, 1, scene.fs, scene.mode, max_sounds=max(8, len(scene.sounds)))     # only its clip structs + restated setters
    syn = nr.Synth(scene.num_buses, scene.voices_per_bus, scene.fs, scene.mode)
    for i, (L, R, sr) in enumerate(scene.sounds):
        assert ref.register_clip(L, R, sr) == i and syn.register(L, R, sr) == i
        if i in scene.clip_setup:
            scene.clip_setup[i](ref.lib, ref.clips[i])
        _copy_clip(ref.clips[i], syn.clips[i])
    N, K, B = scene.nframes, scene.nblocks, scene.num_buses
    bus = np.zeros((B, 2, K * N), dtype=np.float32)
    reports = {}
    for k in range(K):
        for ev in scene.events.get(k, []):
            if ev[0] == "cmd":
                syn.handle(_np_command(ev[1]), ev[2])
            elif ev[0] == "clip":
                ev[2](ref.lib, ref.clips[ev[1]])
                _copy_clip(ref.clips[ev[1]], syn.clips[ev[1]])
            elif ev[0] == "enable":
                syn.enabled[ev[1]] = bool(ev[2])
            elif ev[0] == "start":
                syn.start_voice(ev[1], ev[2], _np_command(ev[3]), ev[4])
            elif ev[0] == "update":
                syn.update_voice(ev[1], ev[2], _np_command(ev[3]))
            elif ev[0] == "stopv":
                syn.stop_voice(ev[1], ev[2], bool(ev[3]))
            else:
                raise AssertionError(ev[0])
        ck = scene.make_clocks(k, 1)[0]
        L, R, reports = syn.process(N, nr.Clock(int(ck.current_usecs), int(ck.next_usecs), int(ck.jack_playhead), int(ck.jack_playhead_usecs),
                                                int(ck.jack_subbeat_length_usecs)))
        bus[:, 0, k * N:(k + 1) * N] = L
        bus[:, 1, k * N:(k + 1) * N] = R
    return bus, reports, syn


_SHAPES = [{}, dict(nframes=64, nblocks=20), dict(nframes=256, nblocks=8), dict(fs=44100.0), dict(voices_per_bus=2), dict(num_buses=3, nclips=8, nblocks=16)]


@pytest.mark.parametrize("seed,mode,kw", [(7001 + i, [0, 0, 4, 3, 2, 0, 4, 1][i % 8], _SHAPES[i % len(_SHAPES)]) for i in range(32)])
def test_c_oracle_equals_the_numpy_restatement_on_random_scenes(seed, mode, kw):
    args = dict(num_buses=2, voices_per_bus=3, nframes=128, nblocks=12, nclips=5, min_len=900, max_len=2600, mode=mode)
    args.update(kw)
    sc = random_scene(seed, **args)
    obus, orep, osyn = run_oracle(sc)
    nbus, nrep, nsyn = run_numpy(sc)
    assert np.abs(obus).max() > 0
    assert np.array_equal(obus.view(np.int32), nbus.view(np.int32)), f"first difference at frame {np.argwhere(obus != nbus)[:2].tolist()}"
    VPB = sc.voices_per_bus
    for b in range(sc.num_buses):
        for i, v in enumerate(nsyn.voices[b]):
            ov = osyn.voices[b * VPB + i]
            assert bool(ov.isPlaying) == bool(v.is_playing), (b, i)
            if v.is_playing:
                assert ov.sourceSamplePosition == float(v.P), (b, i)
                r = orep[b * VPB + i]
                if (b, i) not in nrep:                              # its channel is disabled: not processed, no report
                    assert not r.valid and not nsyn.enabled[b]
                    continue
                valid, gain, prog, _ = nrep[(b, i)]
                assert bool(r.valid) == bool(valid) and r.gain == gain and r.progress == prog, (b, i)


@pytest.mark.parametrize("seed", range(20))
def test_positions_model_and_meter_chain_twins_agree(seed):
    """ClipAudioSourcePositionsModel (32 rows, the reused row 31, the one-second orphan clean-up, the 0.01 peak hysteresis) and the
    per-clip level / progress chain (ClipAudioSource.cpp:88-113,225-240: -100 dB floor, x0.94 fade, 30 / 100 ms gates, 0.1 dB and 0.001
    thresholds): random sequences of create / update / remove / idle gaps on the C oracle and on its numpy twin -- every row, every return
    value and every callback value with the tick it fired in."""
    from oracle import zl_oracle as zo
    lib = zo.load()
    rng = np.random.default_rng(8800 + seed)
    oc = zo.Clip(); lib.zlo_clip_init(C.byref(oc), C.c_float(2.5), 48000.0)
    lib.zlo_clip_set_start_position(C.byref(oc), C.c_float(float(rng.uniform(0, 1))))
    om = zo.ClipMeter(); lib.zlo_clip_meter_init(C.byref(om))
    nm, nmeter = nr.PositionsModel(), nr.ClipMeter()
    val = C.c_float()
    now = 10_000_000
    held = []                                                     # ids handed out and not yet removed (duplicates of 31 included)
    has_cb = bool(seed % 3)
    fired = 0
    for step in range(700):
        now += int(rng.choice([1, 3, 3, 5, 17, 31, 31, 120, 1500 if rng.random() < 0.05 else 40]))
        a = rng.random()
        if a < 0.25 or not held:
            o = lib.zlo_positions_create(C.byref(oc.positions), C.c_float(0.0), now)
            n = nm.create(0.0, now)
            assert o == n
            held.append(o)
        elif a < 0.8:
            for pid in held if rng.random() < 0.5 else held[:1]:
                g, p = float(np.float32(rng.uniform(0, 1.5) if rng.random() < 0.9 else 0.0)), float(np.float32(rng.uniform(0, 1)))
                lib.zlo_positions_set_gain_and_progress(C.byref(oc.positions), pid, C.c_float(g), C.c_float(p), now)
                nm.set_gain_and_progress(pid, g, p, now)
        else:
            pid = held.pop(int(rng.integers(0, len(held))))
            lib.zlo_positions_remove(C.byref(oc.positions), pid, now)
            nm.remove(pid, now)
        for i in range(32):
            r = oc.positions.pos[i]
            assert (r.id, r.progress, r.gain, r.lastUpdated) == (nm.rows[i]["id"], nm.rows[i]["progress"], nm.rows[i]["gain"], nm.rows[i]["updated"]) or \
                (r.id == -1 and nm.rows[i]["id"] == -1 and r.gain == nm.rows[i]["gain"]), (step, i)
        want = lib.zlo_sync_audio_level(C.byref(om), C.byref(oc), now, C.byref(val))
        got = nmeter.sync_audio_level(nm, now)
        assert (got is not None) == bool(want) and (not want or val.value == got), (step, val.value, got)
        fired += int(bool(want))
        want = lib.zlo_sync_progress(C.byref(om), C.byref(oc), 1 if has_cb else 0, now, C.byref(val))
        got = nmeter.sync_progress(nm, oc.startPositionInSeconds, oc.duration, has_cb, now)
        assert (got is not None) == bool(want) and (not want or val.value == got), (step, val.value, got)
        assert lib.zlo_positions_peak_gain(C.byref(oc.positions)) == nm.peak_gain()
        assert lib.zlo_positions_first_progress(C.byref(oc.positions)) == nm.first_progress()
    assert fired > 20 and len(held) >= 0


def _compare(sc):
    obus, orep, osyn = run_oracle(sc)
    nbus, nrep, nsyn = run_numpy(sc)
    assert np.abs(obus).max() > 0
    assert np.array_equal(obus.view(np.int32), nbus.view(np.int32)), f"first difference at {np.argwhere(obus != nbus)[:2].tolist()}"
    VPB = sc.voices_per_bus
    for b in range(sc.num_buses):
        for i, v in enumerate(nsyn.voices[b]):
            ov = osyn.voices[b * VPB + i]
            assert bool(ov.isPlaying) == bool(v.is_playing), (b, i)
            if v.is_playing:
                assert ov.sourceSamplePosition == float(v.P), (b, i)


def _edge_names():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from edge_scenes import SCENES
    return sorted(SCENES)


@pytest.mark.parametrize("name", _edge_names())
def test_edge_scenes_on_the_numpy_twin(name):
    """The hand-made edge scenes (tests/edge_scenes.py) that the engine and the host harness are held to the C oracle with -- here the C
    oracle itself against its numpy twin: the voice-level calls (setCurrentCommand with restart, stopNote with and without a tail),
    disabled channels, clip edits under playing loops, slices, envelope limits, the moving playhead."""
    from edge_scenes import SCENES
    sc = SCENES[name]()
    if sc.nblocks * sc.nframes > 40000:
        sc.nblocks = 40000 // sc.nframes                          # (the twin steps frame by frame in Python)
        sc.events = {k: v for k, v in sc.events.items() if k < sc.nblocks}
    _compare(sc)
