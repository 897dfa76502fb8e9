#!/usr/bin/env python3
"""Per-GPU shares of BASELINE.json configs 2..5 through bench.py (no CPU leg), every line with its BINDING UNIT as measured:
for each shape one un-profiled bench run (K2's share of the 8 TB/s nominal HBM peak from HIP events, output check against the oracle)
and one `rocprofv3 --pmc` pass of the same command, from which K2's busy fractions are taken:

    TA   = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs)                      the texture addresser: one gather instruction per lane and tap pair
    VALU = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)  vector issue
    HBM  = algorithmic GB/s / 6300 (what this chip's HBM delivers to a pure streaming kernel, MI355X_MICROARCH.md) -- meaningful
           on the `--loop-seconds 12 / 45` shapes only, whose sources are never re-read inside a launch (the 2 s shapes are served
           by the Infinity Cache to a large part and may exceed it)

`bound` names the busiest of the three.  Usage (on the GPU box, from the repository root): python3 scripts/config_sweep.py <tag> [--no-pmc]
Writes gpurun_out/config_sweep_<tag>.txt (and .jsonl with the raw bench lines)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_STREAM_GBS = 6300.0

SHAPES = [
    ("configs[1]: 64 voices at the playback rate", "--voices 64 --buses 8 --frames 256"),
    ("configs[1]: 64 voices, 44.1 kHz sources resampled, notes +-12", "--voices 64 --buses 8 --frames 256 --source-rate 44100 --notes 48,72"),
    ("the reference's shape: 96 voices on 12 x 8", "--voices 96 --buses 12 --frames 256"),
    ("the reference's shape, resampled", "--voices 96 --buses 12 --frames 256 --source-rate 44100 --notes 48,72"),
    ("configs[2]: 1024 loops, 128-frame blocks", "--voices 1024 --buses 8 --frames 128"),
    ("configs[3] share: 1024 voices pitched 0.5-2x, linear", "--voices 1024 --buses 8 --notes 48,72"),
    ("configs[3] share: 1024 voices pitched 0.5-2x, 4-tap Hermite", "--voices 1024 --buses 8 --notes 48,72 --hermite"),
    ("(variant) 1024 voices at ratio 1, 4-tap Hermite", "--voices 1024 --buses 8 --hermite"),
    ("configs[4] share: 4096 voices @ 96 kHz, 3750-block bounce", "--voices 4096 --buses 32 --fs 96000 --loop-seconds 2 --blocks-per-step 3750"),
    ("HBM only: headline shape on 45 s sources (one launch per 43.7 s call)", "--voices 1024 --buses 8 --loop-seconds 45"),
    ("HBM only: pitched linear", "--voices 1024 --buses 8 --notes 48,72 --loop-seconds 12"),
    ("HBM only: pitched 4-tap Hermite", "--voices 1024 --buses 8 --notes 48,72 --hermite --loop-seconds 12"),
    ("HBM only: 128-frame blocks", "--voices 1024 --buses 8 --frames 128 --loop-seconds 12"),
]
COUNTERS = ["GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "TA_BUSY_avr"]


def bench_line(args, extra=()):
    # (the timed region starts behind a synchronisation: its first call plans with nothing to hide behind.  Small engines' calls are a
    # quarter of a millisecond, so they get more of them per region -- 48 instead of 6 -- or the start-up reads as 15 % of every step)
    small = any(a in args for a in ("--voices 64", "--voices 96"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-reuse-check", "--no-repeats", "--steps", "48" if small else "6", "--warmup", "2"] + args.split() + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    if p.returncode != 0:
        return None, p.stderr.decode(errors="replace")[-600:]
    return json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]), ""


def pmc_pass(args, tag, i):
    out = os.path.join(ROOT, "gpurun_out", f"sweep_pmc_{tag}_{i}")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline",
                                                "--no-reuse-check", "--no-repeats", "--no-spot-check", "--steps", "3", "--warmup", "1"] + args.split()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env, cwd="/tmp")
    acc = defaultdict(float)
    n = defaultdict(int)
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "zl_k2_render" in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    shutil.rmtree(out, ignore_errors=True)
    if not acc or acc.get("GRBM_GUI_ACTIVE", 0) <= 0:
        return None, (p.stderr.decode(errors="replace")[-400:] if p.returncode else "no K2 rows in the counter file")
    cyc = acc["GRBM_GUI_ACTIVE"] / 8.0
    return {"ta": acc["TA_BUSY_avr"] / cyc, "valu": acc["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0), "wait": acc["SQ_WAIT_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1.0),
            "dispatches": n["GRBM_GUI_ACTIVE"], "valu_insts": acc["SQ_INSTS_VALU"]}, ""


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r"
    pmc = "--no-pmc" not in sys.argv
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    lines, raw = [], []
    for i, (what, args) in enumerate(SHAPES):
        d, err = bench_line(args)
        if d is None:
            lines.append(f"### {what}: bench.py {args}\n  FAILED: {err}")
            print(lines[-1], flush=True)
            continue
        raw.append(json.dumps({"args": args, "line": d}))
        r = d["roofline"]
        hbm_only = "--loop-seconds 12" in args or "--loop-seconds 45" in args
        chk = d.get("output_check") or {}
        ok = all(c["bit_exact"] for c in (chk.get("rows_vs_oracle") or [])) if chk else None
        head = (f"### {what}: bench.py {args}\n  value {d['value']:.3e} vs/s  {d['ms_per_step']:.3f} ms/step  K2 {r['achieved']:.0f} GB/s = {r['frac'] * 100:.1f} % of 8 TB/s "
                f"({'HBM only' if hbm_only else 'cache-assisted'})  {r['avg_launch_ms']:.3f} ms/launch x{r['launches_per_step']}  {r['bytes_per_voice_sample']:.2f} B/vs  "
                f"other ms {[round(x, 3) for x in r['other_ms_per_step'].values()]}  output check vs oracle: {'bit-exact' if ok else ('n/a' if ok is None else 'FAILED')}")
        bound = ""
        if pmc:
            c, err = pmc_pass(args, tag, i)
            if c is None:
                bound = f"\n  bound: (PMC pass failed: {err})"
            else:
                units = {"ta": c["ta"], "valu-issue": c["valu"]}
                if hbm_only:
                    units["hbm"] = r["achieved"] / HBM_STREAM_GBS
                name = max(units, key=units.get)
                vs = float(d["config"]["voices_per_gpu"]) * d["config"]["blocks_per_step"] * 4 * max(1, d["config"]["frames_per_block"] // 64) / 4   # voice-waves per step
                bound = (f"\n  bound: {name}  --  TA {c['ta']:.2f}, VALU {c['valu']:.2f}" + (f", HBM {units['hbm']:.2f} of the {HBM_STREAM_GBS:.0f} GB/s stream rate" if hbm_only else
                         ", HBM n/a (cache-assisted: part of the source reads are Infinity-Cache hits)") +
                         f"; waves in s_waitcnt {c['wait']:.2f} of their cycles; {c['valu_insts'] / (4 * vs):.1f} VALU instructions per voice-wave (4 profiled steps, {c['dispatches']} K2 dispatches)")
        lines.append(head + bound)
        print(lines[-1], flush=True)
    open(os.path.join(ROOT, "gpurun_out", f"config_sweep_{tag}.txt"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(ROOT, "gpurun_out", f"config_sweep_{tag}.jsonl"), "w").write("\n".join(raw) + "\n")


if __name__ == "__main__":
    main()
