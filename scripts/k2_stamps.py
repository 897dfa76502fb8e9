#!/usr/bin/env python3
"""Diagnostic: per-workgroup timestamps of K2 (libzlhip_stamps.so, the -DZL_STAMPS build: start, end of the staging prologue, end) for one
batch call of an engine shape.  usage: k2_stamps.py [voices] [buses] [blocks] [source_rate lo_note hi_note]
Prints workgroup lifetime, the share of it spent staging, and how many workgroups are resident over time.  Not part of the product."""
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

V = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
KB = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
SR = float(sys.argv[4]) if len(sys.argv) > 4 else None
NOTES = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (60, 60)
N, fs = 256, 48000.0
loop_frames = int(2.0 * (SR or fs))
dev = torch.device("cuda", 0)
syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=KB, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(loop_frames + 16) * 8 * V + (1 << 20))
bench.build_scene(syn, torch, dev, V // B, B, fs, loop_frames, 1, notes=NOTES, source_rate=SR)
for i in range(2):
    syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=i * KB))
    syn.synchronize()
syn.enable_trace(True)
syn.set_profiling(True)
syn.render_batch(KB, N, synthetic_clocks(KB, N, fs, start_block=2 * KB))
syn.synchronize()
t = syn.last_timings()
tr = syn.read_trace().reshape(-1)
nb = max(1, min(16, 128 // (V // B))) if V // B <= 64 else 1      # narrow buses per workgroup (zl_engine.cpp)
nwg_z = (B + nb - 1) // nb
nwg = 2 * KB * nwg_z          # (a split tail adds workgroups: every stamped record in this range is one)
st = tr[: nwg * 8].view(np.uint64).reshape(nwg, 4)
ok = st[:, 0] != np.uint64(0xffffffffffffffff)
st = st[ok]
t0, t1, t2 = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64), st[:, 2].astype(np.float64)
ta, tb = (st[:, 3] & np.uint64(0xffff)).astype(np.float64) * 10e-3, ((st[:, 3] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64) * 10e-3
base = t0.min()
dur, stage = (t2 - t0) * 10e-3, (t1 - t0) * 10e-3          # 100 MHz ticks -> us
print(f"V={V} B={B} blocks={KB} source rate {SR or fs:.0f} notes {NOTES}: K2 {t.render_ms * 1e3:.1f} us by its events, {(t2.max() - base) * 10e-3:.1f} us by the stamps; {len(st)} workgroups stamped")
order = np.argsort(t0)
late = dur[order][-len(dur) // 16:]
print(f"lifetime of the last sixteenth of the workgroups to start: mean {late.mean():.2f} us")
print(f"workgroup lifetime us: mean {dur.mean():.2f} p50 {np.percentile(dur, 50):.2f} p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f}")
print(f"staging prologue us:   mean {stage.mean():.2f} p50 {np.percentile(stage, 50):.2f} p99 {np.percentile(stage, 99):.2f}  = {100 * stage.sum() / dur.sum():.1f} % of the lifetimes")
print(f"  of it: up to the first barrier {ta.mean():.2f} us, loads landed at {tb.mean():.2f} us (p99 {np.percentile(tb, 99):.2f}), classification + LDS + second barrier end at {stage.mean():.2f} us")
edges = np.linspace(0, (t2.max() - base), 17)
print("resident workgroups at 16 instants:", [int(((t0 <= (a + b) / 2 + base) & (t2 > (a + b) / 2 + base)).sum()) for a, b in zip(edges[:-1], edges[1:])])
