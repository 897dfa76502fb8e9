#!/usr/bin/env python3
"""Copies the results of scripts/final_check.sh <tag> from gpurun_out/ into profiles/<name>_* with a header that states
what was profiled and how the figures were derived.  usage: publish_profile.py <tag> <name>   (e.g. r1h round1_d)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def jline(path):
    return [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1]


bl = jline(os.path.join(G, f"{tag}_bench_line.json"))
d = json.loads(bl); r = d["roofline"]
open(os.path.join(P, f"{name}_bench_line.json"), "w").write(bl + "\n")
body = open(os.path.join(G, f"{tag}p_summary.txt")).read().splitlines()
K2 = "void zl_k2_render<0u, 1>(ZlBatch)"


def counter(c):
    for l in body:
        if l.startswith(f"{K2}, {c},"):
            p = [x.strip() for x in l.split(",")]
            return int(p[-3]), float(p[-2]), float(p[-1])
    raise SystemExit(f"counter {c} not found")


fetch, write, valu, grbm, insts = (counter(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"))
stats = [l for l in body if l.startswith(K2 + ",")][0].split(",")
ndisp, avg_ns = int(stats[-7]), float(stats[-5])
shapes = [l.split(",") for l in body if l.startswith(K2 + ", 256x")]
big = max(shapes, key=lambda f: int(f[-4]))                       # the launch shape with the most dispatches: the full windows
grid, n_big, avg_big, min_big, max_big = big[-5].strip(), int(big[-4]), float(big[-3]), float(big[-2]), float(big[-1])
shape_list = ", ".join(f"{f[-4].strip()} x {f[-5].strip().split('x')[1]} blocks" for f in sorted(shapes, key=lambda f: -int(f[-5].strip().split('x')[1])))
calls = 5
alg_step = r["algorithmic_bytes_per_launch"] * r["launches_per_step"]
alg_total = alg_step * calls
traffic = 2 * fetch[2] * 1024 + write[2] * 1024
ratio = traffic / alg_total
json.dump({"kernel": "zl_k2_render<0,1>", "workload": "bench.py defaults (1024 voices, 8 buses, 256 frames, 8192 blocks per step, 2 s loops)",
           "fetch_size_kib_sum": fetch[2], "write_size_kib_sum": write[2], "dispatches": fetch[0], "calls": calls,
           "gfx950_fetch_correction": "FETCH_SIZE counts half of the bytes of 16-B/lane streaming loads: reads = 2 x FETCH_SIZE",
           "traffic_bytes_total": traffic, "algorithmic_bytes_total": alg_total, "traffic_over_algorithmic": ratio,
           "source": f"profiles/{name}_rocprofv3_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"},
          open(os.path.join(P, f"{name}_pmc.json"), "w"), indent=1)
ach, lm, ab = r["achieved"], r["avg_launch_ms"] * 1e3, r["algorithmic_bytes_per_launch"] / 1e9
busy = valu[1] * 4 / ((grbm[1] / 8) * 1024)
nr = r.get("no_reuse_variant") or {}
hdr = f"""# rocprofv3 summary, round 1, final state (profiles/{name}_*; written by scripts/publish_profile.py from scripts/final_check.sh)
# commands (scripts/prof_counters.sh; raw CSVs condensed by scripts/summarize_prof.py; long torch kernel names cut):
#   rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reuse-check
#   rocprofv3 --pmc <counters> --output-format csv           -- same command, one pass per counter group
# {calls} zlhip_render_batch calls of 8192 blocks x 1024 voices = {ndisp} K2 dispatches ({shape_list}).  A call that
# follows a synchronisation (the warm-up call and the first timed call) starts with a quarter-size plan window, whose
# planning nothing hides; calls queued behind a call that is still rendering use full windows of 2048 blocks.
# zl_k1_plan / zl_k1c_assemble run on the planning stream(s) concurrently with zl_k2_render.  zl_k3_finalize is not
# launched in this configuration (levels are scanned inside K2).  The torch / copyBuffer kernels in the statistics are
# the scene set-up of bench.py (source generation, clip upload), outside the timed region.
#
# K2 zl_k2_render<0u, 1> (MODE 0 = faithful linear, 1 block per workgroup).  bench.py times the K2 launches of its timed
#   steps with the dispatches' own start / stop events (hipExtLaunchKernel) on the launch stream: {lm:.1f} us per 2048-block launch (unprofiled run, {name}_bench_line.json):
#   {ab:.3f} GB algorithmic / {lm:.1f} us = {ach / 1e3:.2f} TB/s = {ach / 80:.1f} % of 8 TB/s.  rocprofv3, same launch shape
#   (section "engine kernels by grid", {grid} threads): {avg_big / 1e3:.1f} us average over {n_big} dispatches (min {min_big / 1e3:.1f},
#   max {max_big / 1e3:.1f}) = {ab / (avg_big / 1e9) / 1e3:.2f} TB/s -- within {abs(avg_big / 1e3 - lm) / lm * 100:.1f} % of each other (kernels run a little slower under the profiler).  (The --stats average over all {ndisp} dispatches,
#   {avg_ns / 1e3:.1f} us, mixes in the shorter windows.)
# HBM traffic of K2 (PMC, separate passes), summed over the {fetch[0]} dispatches: FETCH_SIZE {fetch[2]:,.0f} KiB; on gfx950
#   FETCH_SIZE counts one half of the bytes of 16-byte-per-lane streaming loads (MI355X_MICROARCH.md, HBM), so reads =
#   2 x {fetch[2] * 1024 / 1e9:.2f} GB = {2 * fetch[2] * 1024 / 1e9:.2f} GB; WRITE_SIZE {write[2]:,.0f} KiB = {write[2] * 1024 / 1e9:.2f} GB.  Algorithmic bytes of the same {calls} calls:
#   {calls} x {alg_step / 1e9:.2f} GB = {alg_total / 1e9:.1f} GB (8192 x 1024 voice-blocks x 2056 B + bus).  traffic / algorithmic = {ratio:.3f}.  (With
#   launch-order block numbering the ratio was 1.061: the cache line at the common edge of two neighbouring blocks of a
#   source was fetched by two XCDs' L2s; the XCD-aware block order lets the two blocks meet in one L2.)  Per 2048-block
#   launch: {ab:.3f} GB algorithmic, {ab * ratio:.2f} GB traffic.  FETCH_SIZE includes Infinity-Cache hits (same guide), so it does
#   not separate HBM from MALL service; bench.py therefore also reports the kernel on sources that are not re-read inside
#   a window (roofline.no_reuse_variant: {nr.get('achieved', 0) / 1e3:.2f} TB/s = {nr.get('frac', 0) * 100:.0f} %).
# VALU: SQ_INSTS_VALU {insts[2]:.4g} over the {calls} calls = {insts[2] / (calls * 8192 * 1024 * 4):.1f} per voice-wave, staging included (28.6 before
#   the interior / unit-step chunk variants); SQ_ACTIVE_INST_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) =
#   {busy * 100:.0f} % VALU busy.
"""
open(os.path.join(P, f"{name}_rocprofv3_summary.txt"), "w").write(hdr + "\n".join(l[:260] for l in body) + "\n")
for f in ("config_sweep", "fanout", "command_storm", "realtime"):
    src = os.path.join(G, f"{tag}_{f}.txt")
    if os.path.exists(src):
        open(os.path.join(P, f"{name}_{f}.txt"), "w").write("".join(l for l in open(src) if "amdgpu.ids" not in l))
print(hdr)
print(f"value {d['value']:.4e}  ms/step {d['ms_per_step']:.3f}  K2 {ach:.0f} GB/s  traffic/algorithmic {ratio:.4f}  cpu {d.get('cpu_baseline', {}).get('value')}")
