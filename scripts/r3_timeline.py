#!/usr/bin/env python3
"""Timeline of the engine's kernels in a rocprofv3 --kernel-trace csv: the last N microseconds before the last kernel's end, in start order,
with the gap between consecutive render kernels.  usage: r3_timeline.py <dir> [window_us]"""
import csv, glob, os, sys
d = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "zl_k" not in n: continue
        short = "K2" if "zl_k2_render" in n else "K1c" if "k1c" in n else "K1" if "zl_k1_plan" in n else "K0" if "k0_apply" in n else "reports" if "zl_k_reports" in n else "K3" if "k3" in n else n[:24]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Grid_Size_Y", "")))
ev.sort()
last = max(e[1] for e in ev if e[2] == "K2")
t0 = last - int(win * 1000)
prev = None
for s, e, w, gy in ev:
    if s < t0 or s > last: continue
    gap = ""
    if w == "K2":
        if prev is not None: gap = f"   gap to previous K2 end {(s - prev) / 1e3:7.1f} us"
        prev = e
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  {w:8s} y={gy}{gap}")
