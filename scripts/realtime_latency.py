#!/usr/bin/env python3
"""Per-block real-time latency of zlhip_render (host buffers in and out, synchronous), SURVEY.md H3: the launched path (three
kernels + one completion event per block) against the resident kernel (ZL_RT_PERSISTENT=1), p50 / p99 / max over 10^4 blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ZL_RT_TRACE"] = "1"
import numpy as np, torch
from libzl_amd import SamplerSynth, clip_command
from libzl_amd.engine import synthetic_clocks

def run(V, B, N, vpt=0, resident=False, blocks=10000, paced=False):
    os.environ["ZL_RT_PERSISTENT"] = "1" if resident else "0"
    os.environ["ZL_RT_WIDE"] = "1" if resident else "0"         # (wide buses: opt-in, measured slower than launches)
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
    ts, slow = [], []
    period = N / fs
    t_next = time.perf_counter()
    for k in range(blocks + 50):
        clk = synthetic_clocks(1, N, fs, start_block=k)[0]
        if paced:                                   # one block per JACK period, as in production (the GPU idles in between)
            while time.perf_counter() < t_next: pass
            t_next += period
        t0 = time.perf_counter(); syn.process(N, clk); ts.append(time.perf_counter() - t0)
        if k >= 50 and ts[-1] > 1e-3:                # a cycle over a millisecond: where was it?  (the engine's own trace, ZL_RT_TRACE=1)
            tr = syn.rt_last_cycle()
            slow.append((k, ts[-1] * 1e6, tr.total_us, tr.before_post_us, tr.wait_us, tr.after_us, tr.max_poll_gap_us, tr.involuntary_switches))
    ts = np.array(ts[50:]) * 1e6
    print(f"V={V:5d} B={B:3d} N={N:4d} vpt={vpt:3d} {'resident' if resident else 'launched'} {'paced   ' if paced else 'back2back'}: zlhip_render p50 {np.median(ts):7.1f} us  p99 {np.percentile(ts, 99):7.1f} us  "
          f"max {ts.max():7.1f} us  (block period {1e6 * period:.0f} us, {blocks} blocks)", flush=True)
    for (k, harness, total, before, wait, after, gap, sw) in slow:
        where = ("the Python harness (allocation of the result arrays, ctypes, the interpreter): the engine call itself was fast" if total < 0.5 * harness else
                 "the waiting THREAD was off its core (host scheduler / cgroup quota of the box), not the device" if gap > 0.5 * total else
                 "host side before the post (a HIP call)" if before > 0.5 * total else "the device or its runtime" if wait > 0.5 * total else "host side after the device was done")
        print(f"    cycle {k:6d}: {harness:8.0f} us seen by this script; inside zlhip_render {total:8.0f} = before the post {before:6.0f} + wait {wait:8.0f} + after {after:5.0f}; "
              f"longest poll gap {gap:8.0f} us, involuntary context switches {sw} -> {where}", flush=True)
    syn.close()

if __name__ == "__main__":
    quick = "--quick" in sys.argv
    n = 2000 if quick else 10000
    for res in (False, True):
        run(96, 12, 256, resident=res, blocks=n)          # the reference's own shape: 12 channels x 8 voices
        run(64, 8, 256, resident=res, blocks=n)           # BASELINE configs[1]
        run(96, 12, 128, resident=res, blocks=n)
        run(96, 12, 512, resident=res, blocks=n)          # JACK at 512 / 1024 frames: the resident workgroup walks 2 / 4 frame tiles
        run(96, 12, 1024, resident=res, blocks=n)
    for res in (False, True):
        run(96, 12, 256, resident=res, blocks=1500, paced=True)
    if not quick or "--wide" in sys.argv:
        # wide buses: launched (per-voice split + K3) against resident (one workgroup per few voices, the bus summed by its last arrival)
        for res in (False, True):
            run(1024, 8, 128, resident=res, blocks=n); run(1024, 8, 256, resident=res, blocks=n); run(256, 8, 256, resident=res, blocks=n)
        run(1024, 8, 256, vpt=16, blocks=n)
        run(1024, 8, 256, resident=True, blocks=1500, paced=True)
