#!/usr/bin/env python3
"""Diagnostic: throughput while voices are in envelope transients (attack / decay / release), i.e. the per-frame
simulation path of K1 and the per-frame control path of K2, against the steady sustain state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from libzl_amd import SamplerSynth, clip_command
from libzl_amd.engine import synthetic_clocks

def run(label, adsr, V=1024, B=8, N=256, KB=512, calls=8, retrigger=False):
    fs = 48000.0; lf = int(2.0 * fs)
    syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=KB, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20))
    src = torch.rand((2, lf), device="cuda") * 2 - 1
    for v in range(V):
        syn.register_clip_device(src[0].data_ptr(), src[1].data_ptr(), lf, fs)
        p = syn.default_clip_params(lf / fs)
        p.length_in_beats = 3.5
        p.length_seconds = float(np.float32((lf - 64 - (v % 17)) / fs))
        p.adsr_attack, p.adsr_decay, p.adsr_sustain, p.adsr_release = adsr
        syn.set_clip_params(v, p)
    vpb = V // B
    def start_all(which):
        cmds = [clip_command(clip=v, midi_note=60, midi_channel=v // vpb - 2, start_playback=1, looping=1, change_volume=1, volume=0.5) for v in which]
        for v, c in zip(which, cmds):
            syn.start_voice(v // vpb, v % vpb, c, 0)
    start_all(range(V))
    syn.set_profiling(True)
    syn.render_batch(KB, N, synthetic_clocks(KB, N, fs)); syn.synchronize()
    syn.profile_totals(reset=True)
    t0 = time.perf_counter()
    for i in range(calls):
        if retrigger:
            which = [v for v in range(V) if v % 4 == i % 4]
            for v in which: syn.stop_voice(v // vpb, v % vpb, True)
            syn.render_batch(KB // 4, N, synthetic_clocks(KB // 4, N, fs, start_block=(i + 1) * KB))
            start_all([v for v in which if not syn.voice_is_playing(v // vpb, v % vpb)])
        syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=(i + 1) * KB + KB // 4))
    syn.synchronize()
    dt = time.perf_counter() - t0
    tot, n = syn.profile_totals()
    blocks = calls * KB + (calls * (KB // 4) if retrigger else 0)
    print(f"{label:44s} {V * blocks * N / dt:.3e} vs/s  wall {dt*1e3:7.1f} ms  K2 {tot.render_ms:7.1f} ms  slow voice-blocks {tot.slow_blocks}")
    syn.close()

run("steady sustain (default ADSR)", (0.0, 0.1, 1.0, 0.05))
run("long attack+decay, voices just started", (2.0, 2.0, 0.6, 0.3), calls=2)
run("retriggered quarter of the voices per call", (0.05, 0.1, 0.7, 0.3), retrigger=True)
