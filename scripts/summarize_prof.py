#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / stats / PMC) into a small per-kernel summary."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"## kernel stats ({os.path.relpath(f, root)})")
        with open(f) as fh:
            for i, row in enumerate(csv.reader(fh)):
                if i < 12:
                    lines.append(",".join(row))
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        dur = defaultdict(list)
        meta = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")
                dur[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                meta[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                      "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")}
        lines.append(f"## kernel trace ({os.path.relpath(f, root)}): name, calls, avg_ns, min_ns, max_ns, meta")
        for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{name[:90]}, {len(d)}, {sum(d) / len(d):.0f}, {min(d)}, {max(d)}, {meta[name]}")
        # the engine's kernels by launch shape (a call's plan windows differ in size: compare like with like)
        shaped = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")
                if "zl_k" in name and "interleave" not in name:
                    shaped[(name, row.get("Grid_Size_X"), row.get("Grid_Size_Y"), row.get("Grid_Size_Z"))].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        lines.append("## engine kernels by grid (threads x, y, z): name, grid, calls, avg_ns, min_ns, max_ns")
        for (name, gx, gy, gz), d in sorted(shaped.items(), key=lambda kv: (kv[0][0], -len(kv[1]))):
            lines.append(f"{name[:60]}, {gx}x{gy}x{gz}, {len(d)}, {sum(d) / len(d):.0f}, {min(d)}, {max(d)}")
        # the engine's dispatches of the last few calls on one time axis (what overlaps what, where the gaps are)
        tl = []
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")
                if "zl_k" in name and "interleave" not in name:
                    tl.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name.split("(")[0][-28:], row.get("Grid_Size_Y")))
        tl.sort()
        if tl:
            t0 = tl[-min(len(tl), 16)][0]
            lines.append("## timeline of the last engine dispatches: start us, end us, duration us, kernel, grid y")
            for a, b, n, gy in tl[-16:]:
                lines.append(f"  {(a - t0) / 1e3:10.1f} {(b - t0) / 1e3:10.1f} {(b - a) / 1e3:9.1f}  {n}  y={gy}")
        # what sits between two render kernels (packets, the planner's hand-off): end of one K2 dispatch -> start of the next
        k2 = []
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "zl_k2_render" in row.get("Kernel_Name", ""):
                    k2.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        k2.sort()
        gaps = [b[0] - a[1] for a, b in zip(k2, k2[1:]) if 0 <= b[0] - a[1] < 1_000_000]     # (gaps of a millisecond and more: between phases of the run)
        if gaps:
            gs = sorted(gaps)
            lines.append(f"## gaps between consecutive K2 dispatches (end -> next start, ns; {len(gaps)} gaps under 1 ms): median {gs[len(gs) // 2]}, mean {sum(gs) / len(gs):.0f}, min {gs[0]}, max {gs[-1]}")
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row.get("Kernel_Name", "?")][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
        lines.append(f"## counters ({os.path.relpath(f, root)}): kernel, counter, dispatches, mean per dispatch, sum over dispatches")
        for name, cs in acc.items():
            if "zl_k" not in name:
                continue
            for c, vals in sorted(cs.items()):
                lines.append(f"{name[:60]}, {c}, {len(vals)}, {sum(vals) / len(vals):.1f}, {sum(vals):.1f}")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
