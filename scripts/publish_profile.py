#!/usr/bin/env python3
"""Copies the results of scripts/profile_r4.sh <tag> a|b|c from gpurun_out/<tag>/ into profiles/<name>_* with headers that state what
was profiled and how every figure is derived.  usage: publish_profile.py <tag> <name>   (e.g. r4a round4_a)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, name = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
K2 = "void zl_k2_render<0u, 1, false>(ZlBatch)"


def jline(path):
    return [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1]


def parse(summary):
    body = open(summary).read().splitlines()
    ctr = {}
    for l in body:
        if l.startswith(K2 + ", ") and l.count(",") == 6:
            p = [x.strip() for x in l.split(",")]
            try:
                ctr[p[-4]] = (int(p[-3]), float(p[-2]), float(p[-1]))      # dispatches, mean, sum
            except ValueError:
                pass
    shapes = [l.split(",") for l in body if l.startswith(K2 + ", 256x")]
    big = max(shapes, key=lambda f: int(f[-5].strip().split("x")[1]))      # the launches with the most blocks (grid y = 256 x blocks x buses)
    total_ns = sum(int(f[-4]) * float(f[-3]) for f in shapes)             # all K2 dispatches of the run
    return body, ctr, dict(grid=big[-5].strip(), n=int(big[-4]), avg=float(big[-3]), mn=float(big[-2]), mx=float(big[-1]), total_ns=total_ns,
                           dispatches=sum(int(f[-4]) for f in shapes)), shapes


bl = jline(os.path.join(G, "bench_line.json"))
d = json.loads(bl); r = d["roofline"]; nr = r["no_reuse_variant"]
open(os.path.join(P, f"{name}_bench_line.json"), "w").write(bl + "\n")
from bench import kernel_source_digest  # noqa: E402
digest = kernel_source_digest()
calls = 5                                                           # profile_r4.sh: --steps 4 --warmup 1
out = {}
for wl, loops in (("reuse", 2.0), ("noreuse", nr["loop_seconds"])):
    body, c, big, shapes = parse(os.path.join(G, f"{wl}_summary.txt"))
    # bytes of one FULL launch (a whole 8192-block call since round 4: every voice of the workload is cheap to plan, one plan window per
    # call); both workloads have the same shape (ratio 1, 256-frame blocks), so the figure holds for both
    alg_call = r["algorithmic_bytes_per_launch"] * r["launches_per_step"]        # per 8192-block call: the same for both workloads (same shape)
    alg_total = alg_call * calls
    BPL = int(big["grid"].split("x")[1])                                         # blocks of the largest launches
    alg_launch = alg_call * BPL / d["config"]["blocks_per_step"]
    rocprof_gbs = alg_total / (big["total_ns"] / 1e9) / 1e9                       # every K2 dispatch of the run: bytes / time
    fetch, write = c["FETCH_SIZE"], c["WRITE_SIZE"]
    traffic = 2 * fetch[2] * 1024 + write[2] * 1024
    rd = c["TCC_EA0_RDREQ_sum"][2]
    busy = c["SQ_ACTIVE_INST_VALU"][2] * 4 / ((c["GRBM_GUI_ACTIVE"][2] / 8) * 1024)
    wait = c["SQ_WAIT_ANY"][2] / c["SQ_WAVE_CYCLES"][2]
    ta = c["TA_BUSY_avr"][2] / (c["GRBM_GUI_ACTIVE"][2] / 8)
    live_ms = (r if wl == "reuse" else nr)["avg_launch_ms"]
    out[wl] = dict(loop_seconds=loops, dispatches=fetch[0], fetch_size_kib_sum=fetch[2], write_size_kib_sum=write[2],
                   tcc_ea0_rdreq_sum=rd, tcc_ea0_rdreq_32b_sum=c["TCC_EA0_RDREQ_32B_sum"][2], tcc_hit_sum=c["TCC_HIT_sum"][2], tcc_miss_sum=c["TCC_MISS_sum"][2],
                   traffic_bytes_total=traffic, algorithmic_bytes_total=alg_total, traffic_over_algorithmic=traffic / alg_total,
                   rocprof_avg_launch_us=big["avg"] / 1e3, rocprof_launches=big["n"], bench_live_avg_launch_us=live_ms * 1e3,
                   achieved_GBs_from_rocprof=rocprof_gbs, valu_busy=busy, wait_any_frac=wait, ta_busy=ta)
    hdr = f"""# rocprofv3 summary ({name}; written by scripts/publish_profile.py from scripts/profile_r4.sh {tag} a), kernel-source digest {digest}
# workload: bench.py defaults (1024 stereo voices, 8 buses x 128, 256-frame blocks, 8192 blocks per call, ratio 1, linear, faithful) with
#   {loops:g} s sources{' -- the BASELINE workload: every source is re-read every 375 blocks, most re-reads are Infinity-Cache hits; one launch per 8192-block call' if wl == 'reuse' else ' and plan windows of 2048 blocks (10.9 s) -- NO source byte is re-read inside a launch: every source read comes from HBM'}
# commands (one pass each; raw CSVs condensed by scripts/summarize_prof.py; long torch kernel names cut):
#   rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-reuse-check --no-spot-check --no-repeats --steps 4 --warmup 1{' --loop-seconds 12 --plan-window 2048' if wl == 'noreuse' else ''}
#   rocprofv3 --pmc <counters> --output-format csv -- same command (5 passes: FETCH_SIZE | WRITE_SIZE TCC_HIT/MISS | TCC_EA0_RDREQ/WRREQ | SQ_* | LDS / TA / TCP)
# {calls} zlhip_render_batch calls = {fetch[0]} K2 dispatches; the full launches are the {big['grid']}-thread ones ({BPL} blocks).
#
# K2 zl_k2_render<0u, 1, false> (faithful linear, 1 block per workgroup, register gather):
#   rocprofv3 kernel trace, full windows: {big['avg'] / 1e3:.1f} us average over {big['n']} dispatches (min {big['mn'] / 1e3:.1f}, max {big['mx'] / 1e3:.1f}).
#   algorithmic bytes per full launch: {alg_launch / 1e9:.3f} GB ({BPL} x 1024 voice-blocks x (ceil(256 x ratio) + 1) x 8 B summed by K1, + the bus write)
#   => {alg_launch / (big['avg'] / 1e9) / 1e12:.2f} TB/s = {alg_launch / (big['avg'] / 1e9) / 8e12 * 100:.1f} % of 8 TB/s under the profiler (all {big['dispatches']} K2 dispatches of the run: {rocprof_gbs / 1e3:.2f} TB/s);
#   bench.py's own HIP-event figure of the un-profiled run ({name}_bench_line.json, {'roofline.achieved' if wl == 'reuse' else 'roofline.achieved_hbm_no_reuse: 12 calls'}):
#   {(r['achieved'] if wl == 'reuse' else nr['achieved']) / 1e3:.2f} TB/s = {(r['frac'] if wl == 'reuse' else nr['frac']) * 100:.1f} % ({abs(rocprof_gbs / (r['achieved'] if wl == 'reuse' else nr['achieved']) - 1) * 100:.1f} % apart; kernels run a little slower under the profiler).
# HBM-side traffic (PMC): FETCH_SIZE {fetch[2]:,.0f} KiB = TCC_EA0_RDREQ {rd:,.0f} requests x 64 B (TCC_EA0_RDREQ_32B = {c['TCC_EA0_RDREQ_32B_sum'][2]:.0f}).  On gfx950 the
#   memory-side read requests of 16-byte-per-lane loads are 128-byte requests tallied at 64 B (MI355X_MICROARCH.md, HBM): reads = 2 x FETCH_SIZE
#   = {2 * fetch[2] * 1024 / 1e9:.2f} GB; WRITE_SIZE {write[2]:,.0f} KiB = {write[2] * 1024 / 1e9:.2f} GB.  Algorithmic bytes of the same {calls} calls: {alg_total / 1e9:.2f} GB.
#   traffic / algorithmic = {traffic / alg_total:.4f}.{' CALIBRATION: in this workload no source byte can come from a cache (nothing is re-read), so the true HBM traffic is >= the algorithmic bytes; 1 x FETCH_SIZE would be half of that minimum, 2 x FETCH_SIZE is 1.00 of it -- the factor 2 holds for this access pattern (8-byte-strided 16-byte gathers).' if wl == 'noreuse' else ' FETCH_SIZE counts Infinity-Cache hits too (same guide): this ratio shows that nothing is over-fetched, not that the bytes came from HBM -- for that see the no-reuse profile.'}
#   L2: TCC_HIT {c['TCC_HIT_sum'][2]:.3g}, TCC_MISS {c['TCC_MISS_sum'][2]:.3g} (source lines always miss L2; what differs between the two workloads is who serves the miss).
# Issue: VALU busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = {busy * 100:.0f} %; waves parked in s_waitcnt {wait * 100:.0f} % of their cycles;
#   TA_BUSY_avr / kernel cycles = {ta * 100:.0f} %; SQ_INSTS_VALU per voice-wave = {c['SQ_INSTS_VALU'][2] / (calls * 8192 * 1024 * 4):.1f}.
"""
    open(os.path.join(P, f"{name}_{wl}_rocprofv3_summary.txt"), "w").write(hdr + "\n".join(l[:260] for l in body) + "\n")
    print(hdr)
json.dump({"kernel": "zl_k2_render<0,1,false>", "kernel_source_digest": digest,
           "workload": "bench.py defaults (1024 voices, 8 buses, 256 frames, 8192 blocks per step, 2 s loops; HBM-only leg: 12 s loops, plan windows of 2048 blocks = 10.9 s)",
           "traffic_over_algorithmic": out["reuse"]["traffic_over_algorithmic"],
           "gfx950_fetch_correction": "reads = 2 x FETCH_SIZE (FETCH_SIZE = TCC_EA0_RDREQ x 64 B, the requests are 128 B); calibrated on the no-reuse workload, "
                                      "where every source byte must come from HBM and 2 x FETCH_SIZE + WRITE_SIZE = %.4f x algorithmic" % out["noreuse"]["traffic_over_algorithmic"],
           "reuse": out["reuse"], "noreuse": out["noreuse"],
           "source": f"profiles/{name}_reuse_rocprofv3_summary.txt, profiles/{name}_noreuse_rocprofv3_summary.txt"},
          open(os.path.join(P, f"{name}_pmc.json"), "w"), indent=1)
for f in ("config_sweep.txt", "host.txt", "realtime_cpp.txt", "realtime_python.txt", "rt_setter_latency.txt", "bounce.txt", "two_ranks_one_gpu.json", "rccl_one_rank.json"):
    src = os.path.join(G, f)
    if os.path.exists(src):
        open(os.path.join(P, f"{name}_{f}"), "w").write("".join(l for l in open(src) if "amdgpu.ids" not in l))
cb = d.get("cpu_baseline", {})
print(f"value {d['value']:.4e}  ms/step {d['ms_per_step']:.3f}  frac {r['frac']:.3f}  frac_hbm_no_reuse {r['frac_hbm_no_reuse']:.3f}  cpu {cb.get('value'):.3e} on {cb.get('cores')} threads")
