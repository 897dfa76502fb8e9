#!/usr/bin/env python3
"""Planning time per call (a synchronisation between calls: nothing hides it) for bench.py's pitched scene: does it stay flat from call
to call (the pass cache holding), and what does it cost for few / many voices?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
import bench
from libzl_amd import SamplerSynth
from libzl_amd.engine import synthetic_clocks

def run(V, B, N, KB, notes, sr, calls=12):
    fs = 48000.0
    loop = int(2.0 * sr)
    syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=KB, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(loop + 16) * 8 * V + (1 << 20))
    bench.build_scene(syn, torch, torch.device("cuda", 0), V // B, B, fs, loop, 0x5A19, notes=notes, source_rate=sr)
    syn.set_profiling(True)
    out = []
    for i in range(calls):
        syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=i * KB))
        t = syn.last_timings()
        out.append((round(t.plan_ms, 3), round(t.render_ms, 3)))
    print(f"V={V} B={B} N={N} K={KB} notes={notes} sr={sr:.0f}: (plan, render) ms per call:", out)
    syn.close()

run(64, 8, 256, 8192, (48, 72), 44100.0)
run(64, 8, 256, 8192, (60, 60), 48000.0)
run(1024, 8, 256, 8192, (48, 72), 48000.0)
