"""Loads tests/golden/*.npz (made by tests/golden/make_golden.py) as a scenario.Scene + expected outputs."""
import glob
import json
import os

import numpy as np

from libzl_amd import _abi
from scenario import Scene

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_MAP = {"clip": "clip", "midi_channel": "midiChannel", "midi_note": "midiNote", "start": "startPlayback", "stop": "stopPlayback",
        "looping": "looping", "change_looping": "changeLooping", "change_volume": "changeVolume", "volume": "volume",
        "change_slice": "changeSlice", "slice": "slice", "change_pitch": "changePitch", "pitch_change": "pitchChange",
        "change_speed": "changeSpeed", "speed_ratio": "speedRatio", "change_gain_db": "changeGainDb", "gain_db": "gainDb"}


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    sc = Scene(num_buses=meta["B"], voices_per_bus=meta["VPB"], fs=meta["fs"], mode=meta["mode"], nframes=meta["nframes"],
               nblocks=meta["nblocks"], bpm=meta["bpm"])
    for i, sr in enumerate(meta["sample_rates"]):
        L = z[f"snd{i}_L"]
        R = z[f"snd{i}_R"] if meta["stereo"][i] else None
        sc.sounds.append((L, R, sr))
        cf = meta["clips"][i]

        def setup(lib, clip, cf=cf):
            clip.startPositionInSeconds = cf["start_sec"]
            clip.lengthInSeconds = cf["length_sec"]
            clip.lengthInBeats = cf["length_beats"]
            clip.volumeAbsolute = cf["volume_abs"]
            clip.pan = cf["pan"]
            clip.duration = cf["duration"]
            clip.rootNote = cf["root_note"]
            clip.nSlicePositions = len(cf["slice_pos"])
            clip.slices = len(cf["slice_pos"])
            for j, p in enumerate(cf["slice_pos"]):
                clip.slicePositions[j] = p
            clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = cf["adsr"]
        sc.clip_setup[i] = setup
    for k, evs in meta["events"].items():
        lst = []
        for ev in evs:
            ev = dict(ev)
            tick = ev.pop("tick", 0)
            kind = ev.pop("kind", "cmd")
            if kind == "enable":
                lst.append(("enable", ev["bus"], bool(ev["on"])))
                continue
            if kind == "stopv":
                lst.append(("stopv", ev["bus"], ev["slot"], bool(ev["tail"])))
                continue
            bus, slot = ev.pop("bus", None), ev.pop("slot", None)
            fields = {_MAP[a]: (1 if b is True else 0 if b is False else b) for a, b in ev.items()}
            if kind == "start":
                lst.append(("start", bus, slot, fields, tick))
            elif kind == "update":
                lst.append(("update", bus, slot, fields))
            else:
                lst.append(("cmd", fields, tick))
        sc.events[int(k)] = lst
    clocks = z["clocks"]

    def make_clocks(start, n, clocks=clocks):
        arr = (_abi.Clock * n)()
        for j in range(n):
            row = clocks[start + j]
            arr[j].current_usecs, arr[j].next_usecs = int(row[0]), int(row[1])
            arr[j].jack_playhead, arr[j].jack_playhead_usecs, arr[j].jack_subbeat_length_usecs = int(row[2]), int(row[3]), int(row[4])
        return arr
    sc.clocks = make_clocks
    expect = dict(bus=np.stack([z["busL"], z["busR"]], axis=1), trace=z["trace"], reports=z["reports"], state=z["state"])
    return sc, expect


def load_config1():
    """tests/golden/c1_config1_shape.npz (make_golden.py --config1): BASELINE configs[0] at its stated shape.  Returns the
    Scene and the expectations: kept blocks (indices, audio [n][B][2][N], source indices [n][V][N]), SHA-256 of the whole
    bus / trace, the loop restarts per voice, the final voice state."""
    z = np.load(os.path.join(GOLDEN_DIR, "c1_config1_shape.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    L = z["source_q15"].astype(np.float32) / np.float32(32768.0)
    sc = Scene(num_buses=meta["B"], voices_per_bus=meta["VPB"], fs=meta["fs"], mode=0, nframes=meta["nframes"], nblocks=meta["nblocks"], bpm=meta["bpm"])
    import ctypes as C
    from scenario import play_cmd
    for i, beats in enumerate(meta["beats"]):
        sc.sounds.append((L, None, 44100.0))

        def setup(lib, clip, beats=beats):
            lib.zlo_clip_set_length(clip, C.c_float(beats), 120)        # ClipAudioSource::setLength, ClipAudioSource.cpp:352-360
            lib.zlo_clip_set_volume_absolute(clip, C.c_float(1.0))
            lib.zlo_clip_set_pan(clip, C.c_float(0.0))
        sc.clip_setup[i] = setup
    sc.events[0] = [("cmd", play_cmd(0, midi_channel=-2), 0), ("cmd", play_cmd(1, midi_channel=-1), 0)]
    c = z["clocks"]
    period = int(c[0][1] - c[0][0])
    assert int(c[1][0]) == period and int(c[2][0]) == (meta["nblocks"] - 1) * period      # the synthetic clock k * 5805
    expect = dict(keep=[int(k) for k in z["keep"]], bus_keep=z["bus_keep"], trace_keep=z["trace_keep"], state=z["state"],
                  bus_sha256=meta["bus_sha256"], trace_sha256=meta["trace_sha256"],
                  restarts={int(v): [tuple(r) for r in rs] for v, rs in meta["restarts"].items()})
    return sc, expect


def check_config1(bus, trace, expect, N=256):
    """bus [B][2][K*N], trace [K][V][N] of any implementation against the fixture: the kept blocks bit for bit, the rest through
    the digests of the whole arrays."""
    import hashlib
    for j, k in enumerate(expect["keep"]):
        got = bus[:, :, k * N:(k + 1) * N]
        assert np.array_equal(got.view(np.int32), expect["bus_keep"][j].view(np.int32)), f"block {k}: audio differs (max {np.abs(got - expect['bus_keep'][j]).max()})"
        if trace is not None:
            assert np.array_equal(trace[k], expect["trace_keep"][j]), f"block {k}: source indices differ"
    assert hashlib.sha256(np.ascontiguousarray(bus).tobytes()).hexdigest() == expect["bus_sha256"], "whole-bus digest differs"
    if trace is not None:
        assert hashlib.sha256(np.ascontiguousarray(trace).tobytes()).hexdigest() == expect["trace_sha256"], "whole-trace digest differs"


def scheduler_session_from_golden(g):
    """tests/golden/s1_scheduler.npz -> (N, fs, t0, ops per cycle) in the form tests/test_scheduler.py runs a session from."""
    from oracle import np_restatement as npr
    meta = json.loads(bytes(g["meta"]).decode())
    order, kinds = meta["cmd_order"], meta["op_kinds"]
    ops = [[] for _ in range(meta["ncycles"])]
    bools = {"start", "stop", "change_slice", "change_looping", "looping", "change_pitch", "change_speed", "change_gain_db", "change_volume"}
    ints = {"clip", "midi_note", "midi_channel", "slice"}
    for row in g["ops"]:
        k, kind, a, b = int(row[0]), kinds[int(row[1])], int(row[2]), int(row[3])
        if kind == "schedule":
            f = {}
            for name, v in zip(order, row[4:]):
                f[name] = bool(v) if name in bools else int(v) if name in ints else np.float32(v)
            ops[k].append(("schedule", npr.Command(**f), b))
        elif kind in ("start", "bpm"):
            ops[k].append((kind, a))
        elif kind in ("qstart", "qstop"):
            ops[k].append((kind, a, b))
        else:
            ops[k].append((kind,))
    return meta["N"], meta["fs"], meta["t0"], ops
