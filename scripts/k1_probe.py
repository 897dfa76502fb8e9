#!/usr/bin/env python3
"""Diagnostic: planning time (K0+K1+K1c before the first K2 launch) for different scene shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from libzl_amd import SamplerSynth, clip_command
from libzl_amd.engine import synthetic_clocks

def run(label, V=1024, B=8, N=256, KB=512, loop_s=2.0, play=True, beat=False):
    fs = 48000.0
    lf = int(loop_s * fs)
    syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=KB, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20))
    src = torch.rand((2, lf), device="cuda") * 2 - 1
    for v in range(V):
        syn.register_clip_device(src[0].data_ptr(), src[1].data_ptr(), lf, fs)
        p = syn.default_clip_params(lf / fs)
        p.length_in_beats = 4.0 if beat else 3.5
        p.length_seconds = float(np.float32((lf - 64 - (v % 17)) / fs))
        syn.set_clip_params(v, p)
    if play:
        for v in range(V):
            syn.start_voice(v // (V // B), v % (V // B), clip_command(clip=v, midi_note=60, midi_channel=v // (V // B) - 2, start_playback=1, looping=1, change_volume=1, volume=0.5), 0)
    syn.set_profiling(True)
    ts = []
    for i in range(6):
        syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=i * KB))
        t = syn.last_timings()
        ts.append((t.plan_ms, t.render_ms, t.total_ms))
    print(label, "plan/render/total ms per call:", [tuple(round(x, 3) for x in t) for t in ts[1:]])
    syn.close()

run("idle voices      ", play=False)
run("no wraps (60 s)  ", loop_s=60.0, KB=512)
run("2 s loops        ")
run("2 s loops, 64 blk", KB=64)
run("beat-locked 2 s  ", beat=True)
