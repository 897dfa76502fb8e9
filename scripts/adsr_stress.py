#!/usr/bin/env python3
"""Randomised parity stress of the envelope paths (random attack / decay / sustain / release incl. zero, tiny and long
times, note-offs at random blocks) against the oracle.  usage: adsr_stress.py sim|gpu [first_seed] [count]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scenario import random_scene, run_oracle, run_backend, compare_runs
backend = sys.argv[1] if len(sys.argv) > 1 else "sim"
if backend == "sim":
    from cpu_harness.sim import SimSynth as Backend
else:
    from libzl_amd import SamplerSynth as Backend
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([64, 128, 256]))
    nb = int(rng.integers(20, 160))
    sc = random_scene(seed, nframes=N, nblocks=nb, nclips=int(rng.integers(4, 12)), min_len=2000, max_len=30000, events=True, mode=int(rng.choice([0, 3, 4])))
    # random envelopes on top of the scene's own clip set-up: long and short ramps, zero and tiny times, low sustain levels
    for i in list(sc.clip_setup):
        base = sc.clip_setup[i]
        a = float(rng.choice([0.0, 1e-4, rng.uniform(0.001, 0.2), rng.uniform(0.2, 1.5)]))
        d = float(rng.choice([0.0, 1e-4, rng.uniform(0.001, 0.2), rng.uniform(0.2, 1.5)]))
        s_ = float(rng.choice([0.0, 1.0, rng.uniform(0.01, 0.99), 1e-3]))
        r = float(rng.choice([0.0, 1e-4, rng.uniform(0.001, 0.1), rng.uniform(0.1, 0.8)]))
        def setup(lib, clip, base=base, a=a, d=d, s_=s_, r=r):
            base(lib, clip)
            clip.adsr.p.attack, clip.adsr.p.decay, clip.adsr.p.sustain, clip.adsr.p.release = a, d, s_, r
        sc.clip_setup[i] = setup
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    batch = int(rng.choice([1, 7, 64, 1 << 30]))
    try:
        bus, rep, syn, _ = run_backend(sc, Backend, batch=batch)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
        if backend != "sim": syn.close()
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, N, nb, batch, str(e)[:160], flush=True)
print("adsr stress", backend, "done:", count, "failures:", bad)
