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
        "change_slice": "changeSlice", "slice": "slice"}


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


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
            fields = {_MAP[a]: (1 if b is True else 0 if b is False else b) for a, b in ev.items()}
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
