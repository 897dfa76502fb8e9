#!/usr/bin/env python3
"""Timeline of the last bounce in a rocprofv3 --kernel-trace --memory-copy-trace run of scripts/bounce_bench.py: render kernels, conversion kernels
and copies in start order with their durations and the gaps between consecutive render kernels (us)."""
import csv, glob, os, sys
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ks = rows("*kernel_trace.csv"); cs = rows("*memory_copy_trace.csv")
ev = []
for r in ks:
    n = r.get("Kernel_Name", "")
    short = "K2" if "zl_k2_render" in n else "deliver" if "zl_k_deliver" in n else "K1" if "zl_k1_plan" in n else "K1c" if "k1c" in n else "K0" if "k0_apply" in n else "reports" if "zl_k_reports" in n else n[:30]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, ""))
for r in cs:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), r.get("Size", r.get("Bytes", ""))))
ev.sort()
# the last bounce = the last burst of D2H copies; print from 12 ms before the last copy's end
last = max(e[1] for e in ev if e[2].startswith("COPY"))     # (copies into page-locked host memory are listed as device-to-device)
t0 = last - 8_000_000
prevK2 = None
print(f"{'t_us':>9} {'dur_us':>8}  what")
for s, e, w, x in ev:
    if s < t0 or s > last: continue
    gap = ""
    if w == "K2":
        if prevK2 is not None: gap = f"  (gap to previous K2 end {(s - prevK2) / 1e3:7.1f} us)"
        prevK2 = e
    if w in ("K2", "deliver") or w.startswith("COPY"):
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {w} {x}{gap}")
