#!/usr/bin/env python3
"""Per-block real-time latency of zlhip_render (host buffers in and out, synchronous), SURVEY.md H3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from libzl_amd import SamplerSynth, clip_command
from libzl_amd.engine import synthetic_clocks

def run(V, B, N, vpt=0):
    fs = 48000.0
    lf = 96000
    syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=4, max_sounds=V, playback_sample_rate=fs,
                       sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20), voices_per_task=vpt)
    src = torch.rand((2, lf), device="cuda") * 2 - 1
    for v in range(V):
        syn.register_clip_device(src[0].data_ptr(), src[1].data_ptr(), lf, fs)
        p = syn.default_clip_params(lf / fs); p.length_in_beats = 3.5; p.length_seconds = float(np.float32((lf - 64 - v % 17) / fs))
        syn.set_clip_params(v, p)
        syn.start_voice(v // (V // B), v % (V // B), clip_command(clip=v, midi_note=60, midi_channel=v // (V // B) - 2, start_playback=1, looping=1, change_volume=1, volume=0.5), 0)
    ts = []
    for k in range(300):
        clk = synthetic_clocks(1, N, fs, start_block=k)[0]
        t0 = time.perf_counter(); syn.process(N, clk); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[50:]) * 1e6
    print(f"V={V:5d} B={B:3d} N={N:4d} voices_per_task={vpt:3d}: zlhip_render median {np.median(ts):7.1f} us  p99 {np.percentile(ts, 99):7.1f} us  "
          f"(block period {1e6 * N / fs:.0f} us) -> {V * N / np.median(ts) * 1e6:.3e} voice-samples/s PCIe-inclusive")
    syn.close()

run(96, 12, 256)          # the reference's own shape: 12 channels x 8 voices
run(64, 8, 256)           # BASELINE config 1
run(1024, 8, 128)         # config 2 shape
run(1024, 8, 256)
run(1024, 8, 256, vpt=16)
run(1024, 8, 256, vpt=1)  # one voice per task: every voice rendered by its own workgroup, K3 sums them in voice order (= the reference's order)
run(1024, 8, 256, vpt=4)
