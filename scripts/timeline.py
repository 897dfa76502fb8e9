#!/usr/bin/env python3
"""Ordered timeline (kernels + memory copies) of the tail of a rocprofv3 trace directory: name, start offset, duration.
usage: scripts/timeline.py <trace dir> <out.txt> [last_n]"""
import csv, glob, os, sys
root, out = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 120
ev = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K  " + r["Kernel_Name"][:48] + f' grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}'))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MC " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")) + " B"))
ev.sort()
ev = ev[-last:]
t0 = ev[0][0]
with open(out, "w") as fh:
    for s, e, n in ev:
        fh.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n}\n")
