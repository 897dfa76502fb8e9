#!/usr/bin/env python3
"""Practical read-only HBM bandwidth of this box (torch reductions over 4 GiB, larger than the Infinity Cache): the
ceiling for K2's no-reuse variant."""
import torch
x = torch.rand(1 << 30, device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for name, fn in (("sum", lambda: x.sum()), ("amax", lambda: x.amax()), ("sum over rows [4096, 262144]", lambda: x.view(4096, -1).sum(dim=1))):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5): fn()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:32s} {x.numel() * 4 / ms / 1e6:7.0f} GB/s")
