#!/usr/bin/env python3
"""What a command costs a real-time cycle (resident kernel, 12 x 8 voices): mean stage times of workgroup 0 (ZL_RT_STAMPS=1) and the cycle latency,
quiet against cycles that carry a note-off + a start (a retrigger) or a start on a free slot.  usage: rt_cmd_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ZL_RT_PERSISTENT"] = "1"; os.environ["ZL_RT_STAMPS"] = "1"
import numpy as np, torch
from libzl_amd import SamplerSynth, clip_command
from libzl_amd.engine import synthetic_clocks

def run(label, mode, cycles=3000, N=256, fs=48000.0):
    V, B = 96, 12
    lf = 96000
    syn = SamplerSynth(B, 8, max_frames=N, max_batch_blocks=4, max_sounds=V, playback_sample_rate=fs, sound_arena_bytes=(lf + 16) * 8 * V + (1 << 20))
    src = torch.rand((2, lf), device="cuda") * 2 - 1
    nplay = 7 if mode == "start_free" else 8                     # leave one slot per bus free for the starts
    for v in range(V):
        syn.register_clip_device(src[0].data_ptr(), src[1].data_ptr(), lf, fs)
        p = syn.default_clip_params(lf / fs); p.length_in_beats = 3.5; p.length_seconds = float(np.float32((lf - 64 - v % 17) / fs)); p.adsr_release = 0.004
        syn.set_clip_params(v, p)
        if v % 8 < nplay:
            syn.start_voice(v // 8, v % 8, clip_command(clip=v, midi_note=60, midi_channel=v // 8 - 2, start_playback=1, looping=1, change_volume=1, volume=0.5), 0)
    ts = []
    for k in range(cycles + 50):
        if k >= 50 and mode != "quiet" and k % 4 == 0:
            b = (k // 4) % B
            if mode == "retrigger":
                v = b * 8 + (k // (4 * B)) % 8
                syn.stop_voice(b, v % 8, True)                   # note-off: a 4 ms release tail (192 frames: inside one block)
            elif mode == "start_free":
                v = b * 8 + 7
                syn.stop_voice(b, 7, False)                      # (make room: hard stop of the last start)
                syn.start_voice(b, 7, clip_command(clip=v, midi_note=60 + k % 5, midi_channel=b - 2, start_playback=1, looping=1, change_volume=1, volume=0.4), k)
            elif mode == "patch":
                syn.update_voice(b, (k // (4 * B)) % 8, clip_command(clip=b * 8, midi_note=60, midi_channel=b - 2, change_volume=1, volume=0.3 + 0.1 * (k % 5)))
        clk = synthetic_clocks(1, N, fs, start_block=k)[0]
        t0 = time.perf_counter(); syn.process(N, clk); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[50:]) * 1e6
    cmd = ts[0::4] if mode != "quiet" else ts
    print(f"{label:44s} all cycles p50 {np.median(ts):6.1f} us p99 {np.percentile(ts, 99):6.1f}   cycles that carry the command: p50 {np.median(cmd):6.1f} p99 {np.percentile(cmd, 99):6.1f}", flush=True)
    sys.stderr.flush()
    syn.close()                                                   # prints the stage means (stderr)

if __name__ == "__main__":
    run("quiet", "quiet")
    run("volume patch every 4th cycle", "patch")
    run("note-off (4 ms tail) every 4th cycle", "retrigger")
    run("hard stop + start on a free slot every 4th", "start_free")
