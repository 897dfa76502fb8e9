#!/usr/bin/env python3
"""Short batches of wide buses: time of one zlhip_render_batch call of K blocks (1024 voices on 8 buses) with the whole-bus
walk (voices_per_task = 0) against one voice per workgroup (voices_per_task = 1; same summation order)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from libzl_amd import SamplerSynth
from libzl_amd.engine import synthetic_clocks

V, B, N = 1024, 8, 256
lf = 96000
for vpt in (0, 1):
    syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=256, max_sounds=V, sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20), voices_per_task=vpt)
    bench.build_scene(syn, torch, torch.device("cuda:0"), V // B, B, 48000.0, lf, 1234)
    bus = torch.zeros((B, 2, 256 * N), device="cuda")
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for K in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        ts = []
        for i in range(30):
            clk = synthetic_clocks(K, N, 48000.0, start_block=i * K)
            t0 = time.perf_counter()
            syn.render_batch(K, N, clk, bus_out_dev=bus.data_ptr(), stream=st.cuda_stream)
            syn.synchronize(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"voices_per_task {vpt}: K = {K:3d} blocks: {np.median(ts[5:]) * 1e6:8.1f} us per call")
    syn.close()
