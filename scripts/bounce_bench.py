#!/usr/bin/env python3
"""PCIe-inclusive rate of the offline bounce (zlhip_bounce) against the device-resident rate of the same blocks (zlhip_render_batch):
the per-GPU share of BASELINE configs[4] (4096 stereo voices on 32 buses, 96 kHz, 3750 blocks of 256 = 10 s) and the headline shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from libzl_amd import SamplerSynth
from libzl_amd.engine import pinned_array, synthetic_clocks


def run(V, B, fs, K, N=256, loop_seconds=2.0, reps=3):
    loop = int(loop_seconds * fs)
    syn = SamplerSynth(B, V // B, max_frames=N, max_batch_blocks=K, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(loop + 16) * 8 * V + (1 << 20))
    bench.build_scene(syn, torch, torch.device("cuda", 0), V // B, B, fs, loop, 0x5A17 + 5)
    dev = torch.zeros((B, 2, K * N), device="cuda")
    torch.cuda.synchronize(); time.sleep(2.0)
    vs = float(V) * K * N
    blk = 0

    def clocks():
        nonlocal blk
        c = synthetic_clocks(K, N, fs, start_block=blk); blk += K
        return c

    rows = []
    for name, fn in (("device-resident (zlhip_render_batch)", lambda c, o: (syn.render_batch(K, N, c, bus_out_dev=dev.data_ptr()), syn.synchronize())),
                     ("bounce to host, fp32 planar", lambda c, o: syn.bounce(K, N, c, fmt="f32", out=o[0])),
                     ("bounce to host, 16-bit stereo", lambda c, o: syn.bounce(K, N, c, fmt="pcm16", out=o[1]))):
        outs = (pinned_array(syn._lib, (B, 2, K * N), np.float32), pinned_array(syn._lib, (B, K * N, 2), np.int16))
        fn(clocks(), outs)                                           # warm-up (allocations, first touch of the host pages)
        ts = []
        for _ in range(reps):
            c = clocks()
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(c, outs); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        host_bytes = 0 if name.startswith("device") else (outs[0].nbytes if "fp32" in name else outs[1].nbytes)
        rows.append((name, t, vs / t, host_bytes))
        del outs
    print(f"V={V} B={B} fs={fs:.0f} blocks={K} x {N} frames ({K * N / fs:.1f} s of audio, {vs:.3e} voice-samples per bounce)")
    for name, t, rate, hb in rows:
        extra = f"  {hb / 1e6:7.1f} MB to host = {hb / t / 1e9:5.1f} GB/s over PCIe" if hb else ""
        print(f"  {name:40s} {t * 1e3:8.2f} ms  {rate:.3e} voice-samples/s  ({rows[0][1] / t * 100:5.1f} % of device-resident){extra}")
    syn.close()


if __name__ == "__main__":
    # one process per shape, and a pause before timing: D2H copies run at half rate for a while after a large hipFree (the driver
    # clears freed memory with the copy engines; scripts/probes/d2h_probe4.hip), e.g. the arena of the previous shape's engine
    if "--one" in sys.argv:
        i = sys.argv.index("--one")
        V, B, fs, K = int(sys.argv[i + 1]), int(sys.argv[i + 2]), float(sys.argv[i + 3]), int(sys.argv[i + 4])
        run(V, B, fs, K)
    else:
        import subprocess
        for shape in (("4096", "32", "96000", "3750"), ("1024", "8", "48000", "8192")):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", *shape], check=True)
