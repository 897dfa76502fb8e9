import sys, time
sys.path.insert(0, '/root/repo')
import torch
from libzl_amd import SamplerSynth
B, K, N = 8, 8192, 256
syn = SamplerSynth(B, 2, max_frames=N, max_batch_blocks=K, max_sounds=4)
x = torch.rand((B, 2, K * N), device="cuda") * 2 - 1
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(3): syn.levels_scan_device(x.data_ptr(), K, N, stream=st.cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(20): syn.levels_scan_device(x.data_ptr(), K, N, stream=st.cuda_stream)
e1.record(st); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"levels_scan_device {K} blocks x {B} buses x {N} frames: {ms*1e3:.1f} us = {x.numel()*4/ms/1e6:.0f} GB/s")
