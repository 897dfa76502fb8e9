"""SURVEY 8(f) n1 -- JackPassthrough fan-out: bandwidth of the stand-alone kernel and cost of the fan-out per bench step,
fused into K2's bus write against a separate pass over the bus.  Usage (GPU box): python3 scripts/fanout_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench(mode):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-reuse-check", "--steps", "8", "--warmup", "2", "--fanout", mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return d["ms_per_step"], d["value"], d["roofline"]["achieved"]


def main():
    rows = {m: bench(m) for m in ("none", "fused", "separate")}      # child processes first: this one has not touched the GPU yet
    import torch
    from libzl_amd import PassthroughParams, SamplerSynth
    B, frames = 8, 8192 * 256
    syn = SamplerSynth(B, 2, max_frames=64, max_batch_blocks=1, max_sounds=4)
    x = torch.rand((B, 2, frames), device="cuda") * 2 - 1
    out = torch.zeros((B, 6, frames), device="cuda")
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    res = {}
    for name, params in (("multiply (pan != 0)", [PassthroughParams(0.9, 0.5, 0.25, 0.1, 0)] * B),
                         ("copy / zero fast paths (defaults)", [PassthroughParams(1.0, 1.0, 1.0, 0.0, 0)] * B)):
        for _ in range(3):
            syn.passthrough(params, x.data_ptr(), out.data_ptr(), frames, stream=stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20):
            syn.passthrough(params, x.data_ptr(), out.data_ptr(), frames, stream=stream.cuda_stream)
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[name] = (ms, B * frames * 32 / (ms * 1e-3) / 1e9)
    syn.close()
    print(f"stand-alone zl_k_passthrough, {B} buses x {frames} frames (8 B read + 24 B written per bus frame = {B * frames * 32 / 1e6:.0f} MB):")
    for k, (ms, gbs) in res.items():
        print(f"  {k:36s} {ms * 1e3:8.1f} us  {gbs:7.0f} GB/s  ({gbs / 8000 * 100:.0f} % of 8 TB/s)")
    print("bench step (1024 voices, 8192 blocks of 256):")
    base = rows["none"][0]
    for m, (ms, v, gbs) in rows.items():
        print(f"  fan-out {m:9s} {ms:7.3f} ms/step  (+{(ms - base) * 1e3:6.1f} us)  {v:.3e} voice-samples/s  K2 {gbs:.0f} GB/s")


if __name__ == "__main__":
    main()
