#!/usr/bin/env python3
"""Soak of the resident real-time kernel (not part of the test tiers): long mixed scenes cycle by cycle through zlhip_render -- commands,
clip edits, idle spells -- against the oracle, bit for bit.  usage: rt_soak.py [first_seed] [count] [cycles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ZL_RT_PERSISTENT"] = "1"
import numpy as np
from scenario import random_scene, run_oracle
from test_rt_persistent import _play_blockwise

first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([64, 128, 256, 512, 1024, 441, 48]))    # (above 256 frames the resident workgroup walks several frame tiles)
    sc = random_scene(seed, num_buses=12, voices_per_bus=8, nclips=int(rng.integers(10, 24)), mode=int(rng.choice([0, 0, 3, 4])), nframes=N, nblocks=cycles,
                      min_len=3000, max_len=60000)
    # more events than random_scene's six: commands and edits sprinkled over the whole run
    base = sc.events[0]
    for k in sorted(set(int(x) for x in rng.integers(1, cycles, size=cycles // 40))):
        i = int(rng.integers(0, len(base)))
        ev = dict(base[i][1])
        what = int(rng.integers(0, 3))
        lst = sc.events.setdefault(k, [])
        if what == 0:
            lst.append(("cmd", dict(clip=ev["clip"], midiChannel=ev["midiChannel"], midiNote=ev["midiNote"], stopPlayback=1), 0))
        elif what == 1:
            lst.append(("cmd", ev, k))
        else:
            import ctypes as C
            v, p = float(rng.uniform(0.1, 1)), float(rng.uniform(-1, 1))
            lst.append(("clip", ev["clip"], lambda lib, clip, v=v, p=p: (lib.zlo_clip_set_volume_absolute(clip, C.c_float(v)), lib.zlo_clip_set_pan(clip, C.c_float(p)))))
    ref_bus, ref_rep, ref_syn = run_oracle(sc, threads=8)
    bus, rep, syn = _play_blockwise(sc, pause_at=tuple(int(x) for x in rng.integers(1, cycles, size=2)))
    ok = np.array_equal(bus.view(np.int32), ref_bus.view(np.int32))
    starts, cyc = syn.rt_stats()
    syn.close()
    print(f"seed {seed}: N={N} mode={sc.mode} cycles={cycles} events at {len(sc.events)} cycles, peak {np.abs(ref_bus).max():.2f}, resident launches {starts}, cycles {cyc}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"rt soak done: {count} scenes, failures: {bad}")
sys.exit(1 if bad else 0)
