#!/usr/bin/env python3
"""Randomised parity stress on the GPU (not part of the test tiers): long batches, every block size, plan windows of
random size, pipelined calls, all modes, against the CPU oracle.  usage: gpu_stress.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scenario import random_scene, run_oracle, run_backend, compare_runs
from libzl_amd import PassthroughParams, SamplerSynth
import ctypes as C
from oracle import zl_oracle as zo


def oracle_fanout(bus, params):
    lib = zo.load()
    nb_, _, n = bus.shape
    out = np.zeros((nb_, 6, n), dtype=np.float32)
    for b in range(nb_):
        rows = [np.zeros(n, dtype=np.float32) for _ in range(6)]
        arr = (C.c_void_p * 6)(*[o.ctypes.data for o in rows])
        p = zo.Passthrough(params[b].dry_amount, params[b].wet_fx1_amount, params[b].wet_fx2_amount, params[b].pan_amount, params[b].muted)
        L = np.ascontiguousarray(bus[b, 0]); R = np.ascontiguousarray(bus[b, 1])
        lib.zlo_passthrough_process(C.byref(p), L.ctypes.data, R.ctypes.data, arr, n)
        out[b] = np.stack(rows)
    return out


first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([64, 128, 192, 256, 512, 441, 1024, 100]))
    nb = int(rng.integers(20, 700))
    ml = int(rng.choice([300, 1500, 20000]))
    B, VPB, G = [(3, 8, 0), (5, 8, 0), (2, 16, 0), (2, 12, 0), (1, 24, 0), (3, 8, 4), (2, 16, 8), (2, 40, 0), (1, 64, 0)][int(rng.integers(0, 9))]   # buses, width, mix group
    sc = random_scene(seed, nframes=N, nblocks=nb, nclips=int(rng.integers(4, 14)), min_len=ml, max_len=ml + int(rng.choice([500, 5000, 40000])),
                      events=bool(rng.random() < 0.6), mode=int(rng.choice([0, 3, 4])), num_buses=B, voices_per_bus=VPB, mix_group=G)
    ref_bus, ref_rep, ref_syn = run_oracle(sc)
    batch = int(rng.choice([1, 7, 64, 300, 1 << 30]))          # 1: every call a real-time block
    kw = dict(batch=batch, plan_window_blocks=int(rng.choice([0, 0, 5, 64, 300])), pipelined=bool(rng.random() < 0.5))
    fan = None
    if rng.random() < 0.3:                                    # the fused JackPassthrough fan-out next to the bus
        zoo = [PassthroughParams(1.0, 0.0, 0.5, 0.0, 0), PassthroughParams(0.8, 1.0, -1.25, -0.3, 0), PassthroughParams(1.0, 1.0, 1.0, 0.0, 1),
               PassthroughParams(-0.5, 2.0, 0.0, 1.5, 0), PassthroughParams(1.0, 1.0, 1.0, 0.0, 0)]
        fan = [zoo[int(rng.integers(0, len(zoo)))] for _ in range(B)]
        kw["fanout"] = fan
    try:
        bus, rep, syn, _ = run_backend(sc, SamplerSynth, **kw)
        compare_runs(ref_bus, ref_rep, ref_syn, bus, rep, sc.num_buses * sc.voices_per_bus)
        if fan is not None:
            want = oracle_fanout(ref_bus, fan)
            assert np.array_equal(syn.fan_result.view(np.int32), want.view(np.int32)), "fan-out differs"
        syn.close()
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, "N", N, "blocks", nb, kw, str(e)[:200], flush=True)
print("gpu stress done:", count, "scenes, failures:", bad)
sys.exit(1 if bad else 0)
