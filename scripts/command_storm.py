#!/usr/bin/env python3
"""SURVEY 8(f) n2 -- the command front-end under a retrigger storm: every real-time block stops half of the 1024 voices
and starts 512 others (1024 ClipCommands per block through zlhip_handle_commands, one K0 voice-table update inside the
block's render).  Reports the host time of the command batch and the block latency with and without commands."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from libzl_amd import SamplerSynth, clip_command
from libzl_amd.engine import synthetic_clocks


def main(B=128, VPB=8, N=256, blocks=200):
    fs, lf = 48000.0, 48000
    V = B * VPB
    syn = SamplerSynth(B, VPB, max_frames=N, max_batch_blocks=4, max_sounds=V, playback_sample_rate=fs,
                       sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20))
    src = torch.rand((2, lf), device="cuda") * 2 - 1
    for v in range(V):
        syn.register_clip_device(src[0].data_ptr(), src[1].data_ptr(), lf, fs)
        p = syn.default_clip_params(lf / fs); p.adsr_release = 0.0          # noteOff frees the voice in the same block
        syn.set_clip_params(v, p)
    half = VPB // 2
    sets = [[clip_command(clip=b * VPB + s * half + j, midi_note=60, midi_channel=b - 2, start_playback=1, looping=1, change_volume=1, volume=0.5)
             for b in range(B) for j in range(half)] for s in range(2)]
    stops = [[clip_command(clip=b * VPB + s * half + j, midi_note=60, midi_channel=b - 2, stop_playback=1)
              for b in range(B) for j in range(half)] for s in range(2)]
    import ctypes as C
    from libzl_amd._abi import ClipCommand
    n = len(stops[0]) + len(sets[0])
    arrays = [(ClipCommand * n)(*(stops[s] + sets[1 - s])) for s in range(2)]     # marshalled once: the C call is what is timed
    got = (C.c_int32 * n)()
    t_cmd, t_blk, t_quiet, taken = [], [], [], []
    syn.handle_clip_commands(sets[0], 0)
    for k in range(blocks):
        clk = synthetic_clocks(1, N, fs, start_block=k)[0]
        s = k & 1
        t0 = time.perf_counter()
        rc = syn._lib.zlhip_handle_commands(syn._e, arrays[s], n, k * 10, got)
        t1 = time.perf_counter()
        assert rc >= 0
        syn.process(N, clk)
        t2 = time.perf_counter()
        t_cmd.append(t1 - t0); t_blk.append(t2 - t1); taken.append(rc)
    for k in range(blocks):
        clk = synthetic_clocks(1, N, fs, start_block=blocks + k)[0]
        t0 = time.perf_counter(); syn.process(N, clk); t_quiet.append(time.perf_counter() - t0)
    playing = sum(1 for r in syn.voice_reports() if r.playing)
    c, b, q = (np.array(x[20:]) * 1e6 for x in (t_cmd, t_blk, t_quiet))
    print(f"{B} buses x {VPB} voices, {N}-frame blocks (period {1e6 * N / fs:.0f} us), {n} commands per block, {np.mean(taken[20:]):.0f} taken, {playing} voices playing at the end")
    print(f"  zlhip_handle_commands (one call, whole batch):   median {np.median(c):7.1f} us  p99 {np.percentile(c, 99):7.1f} us  -> {n / np.median(c):.2f} M commands/s")
    print(f"  zlhip_render of a block with {n} voice operations: median {np.median(b):7.1f} us  p99 {np.percentile(b, 99):7.1f} us")
    print(f"  zlhip_render of a block without commands:          median {np.median(q):7.1f} us  p99 {np.percentile(q, 99):7.1f} us")
    syn.close()


if __name__ == "__main__":
    main()
