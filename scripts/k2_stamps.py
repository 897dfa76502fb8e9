#!/usr/bin/env python3
"""Diagnostic: per-workgroup start/end timestamps of K2 (libzlhip_stamps.so, -DZL_STAMPS build).
Prints workgroup duration statistics and the concurrency profile.  Not part of the product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libzl_amd import build, _abi
_abi.LIB_PATH = build.build_engine(stamps=True)
import torch
import bench
from libzl_amd import SamplerSynth
from libzl_amd.engine import synthetic_clocks

V, B, N, KB = 1024, 8, 256, int(os.environ.get("KB", "512"))
fs = 48000.0
loop_frames = int(2.0 * fs)
dev = torch.device("cuda", 0)
syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=KB, max_sounds=V, playback_sample_rate=fs,
                   sound_arena_bytes=(loop_frames + 16) * 8 * V + (1 << 20), voices_per_task=int(os.environ.get("VPT", "0")))
bench.build_scene(syn, torch, dev, V // B, B, fs, loop_frames, 1)
for i in range(2):
    syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=i * KB))
    syn.synchronize()
syn.enable_trace(True)
syn.set_profiling(True)
syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=2 * KB))
syn.synchronize()
t = syn.last_timings()
tr = syn.read_trace().reshape(-1)
nwg = KB * B * max(1, (V // B + (int(os.environ.get("VPT", "0")) or V // B) - 1) // (int(os.environ.get("VPT", "0")) or V // B))
st = tr[: nwg * 8].view(np.uint64).reshape(nwg, 4)
t0, t1, t2, hw = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64), st[:, 2].astype(np.float64), st[:, 3]
base = t0.min()
dur = (t2 - t0) * 10e-3     # 100 MHz ticks -> microseconds
stage = (t1 - t0) * 10e-3
print(f"K2 event time {t.render_ms*1e3:.1f} us; span by stamps {(t2.max()-base)*10e-3:.1f} us; {nwg} workgroups")
print(f"WG duration us: mean {dur.mean():.1f} p50 {np.percentile(dur,50):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}")
print(f"staging+classify us: mean {stage.mean():.2f} p99 {np.percentile(stage,99):.2f}")
# concurrency over time
edges = np.linspace(0, (t2.max() - base), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2 + base
    conc = int(((t0 <= mid) & (t2 > mid)).sum())
    print(f"  t={(a*10e-3):7.1f} us  resident WGs {conc}")
paths = np.stack([(hw >> np.uint64(sh)) & np.uint64(0xffff) for sh in (0, 16, 32, 48)], axis=1).astype(np.int64)
print("chunks by path (simple1, simple2, general, ctl): total", paths.sum(axis=0))
# slowest workgroups
idx = np.argsort(-dur)[:8]
for i in idx:
    print(f"  slow wg {i}: bus {i // KB} block {i % KB} dur {dur[i]:.1f} us start {(t0[i]-base)*10e-3:.1f}")
# duration by block index (averaged over buses)
byk = dur.reshape(-1, KB).mean(axis=0) if nwg == KB * B else None
if byk is not None:
    order = np.argsort(-byk)[:24]
    print("slowest blocks (mean us over buses):", [(int(k), round(float(byk[k]), 1)) for k in sorted(order)])
    print("median block:", float(np.median(byk)))
    pk = paths.reshape(-1, KB, 4).sum(axis=0)
    for k in sorted(order)[:12]:
        print("   block", int(k), "paths", pk[k].tolist())
