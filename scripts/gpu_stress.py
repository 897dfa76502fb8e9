#!/usr/bin/env python3
"""Randomised parity stress on the GPU (not part of the test tiers): long batches, every block size, plan windows of
random size, pipelined calls, all modes, against the CPU oracle.  usage: gpu_stress.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scenario import random_scene, run_oracle, run_backend, compare_runs
from libzl_amd import SamplerSynth
first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([64, 128, 192, 256, 512]))
    nb = int(rng.integers(20, 700))
    ml = int(rng.choice([300, 1500, 20000]))
    sc = random_scene(seed, nframes=N, nblocks=nb, nclips=int(rng.integers(4, 14)), min_len=ml, max_len=ml + int(rng.choice([500, 5000, 40000])),
                      events=bool(rng.random() < 0.6), mode=int(rng.choice([0, 3, 4])))
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    batch = int(rng.choice([7, 64, 300, 1 << 30]))
    kw = dict(batch=batch, plan_window_blocks=int(rng.choice([0, 0, 5, 64, 300])), pipelined=bool(rng.random() < 0.5))
    try:
        bus, rep, syn, _ = run_backend(sc, SamplerSynth, **kw)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
        syn.close()
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, "N", N, "blocks", nb, kw, str(e)[:200], flush=True)
print("gpu stress done:", count, "scenes, failures:", bad)
sys.exit(1 if bad else 0)
