#!/usr/bin/env python3
"""Diagnostic: host time of consecutive zlhip_render_batch calls (are they asynchronous?) and the GPU time between them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from libzl_amd import SamplerSynth
from libzl_amd.engine import synthetic_clocks
V, B, N, KB = 1024, 8, 256, int(os.environ.get("KB", "8192"))
fs = 48000.0; lf = int(2.0 * fs)
dev = torch.device("cuda", 0)
syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=KB, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20))
bench.build_scene(syn, torch, dev, V // B, B, fs, lf, 1)
bus = torch.zeros((B, 2, KB * N), device=dev)
st = torch.cuda.Stream(); sp = st.cuda_stream
cks = [synthetic_clocks(KB, N, fs, start_block=i * KB) for i in range(10)]
syn.set_profiling(bool(int(os.environ.get("PROF", "1"))))
for i in range(2): syn.render_batch(KB, N, cks[i], bus_out_dev=bus.data_ptr(), stream=sp)
torch.cuda.synchronize()
t0 = time.perf_counter(); ts = []
for i in range(2, 10):
    a = time.perf_counter()
    syn.render_batch(KB, N, cks[i], bus_out_dev=bus.data_ptr(), stream=sp)
    ts.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) * 1e3
print("host ms per call:", [round(x, 3) for x in ts], " wall per call %.3f ms" % (tot / 8))
