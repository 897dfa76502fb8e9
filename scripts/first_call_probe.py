#!/usr/bin/env python3
"""Duration of a zlhip_render_batch call that follows a synchronisation (nothing hides the planning of its first
window) against a call queued behind another one.  ZL_FIRST_WINDOW_FRAMES overrides the size of the first window."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from libzl_amd import SamplerSynth
from libzl_amd.engine import synthetic_clocks

V, B, N, K = 1024, 8, 256, 8192
lf = 96000
syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=K, max_sounds=V, sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20))
bench.build_scene(syn, torch, torch.device("cuda:0"), V // B, B, 48000.0, lf, 1234)
bus = torch.zeros((B, 2, K * N), device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
cl = [synthetic_clocks(K, N, 48000.0, start_block=i * K) for i in range(40)]
for i in range(3):
    syn.render_batch(K, N, cl[i], bus_out_dev=bus.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize(); syn.synchronize()
alone = []
for i in range(3, 13):
    t0 = time.perf_counter()
    syn.render_batch(K, N, cl[i], bus_out_dev=bus.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize(); syn.synchronize()
    alone.append(time.perf_counter() - t0)
t0 = time.perf_counter()
for i in range(13, 33):
    syn.render_batch(K, N, cl[i], bus_out_dev=bus.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize(); syn.synchronize()
piped = (time.perf_counter() - t0) / 20
print(f"first window {os.environ.get('ZL_FIRST_WINDOW_FRAMES', 'default')}: call after a synchronisation {np.median(alone) * 1e3:.3f} ms, pipelined call {piped * 1e3:.3f} ms")
syn.close()
