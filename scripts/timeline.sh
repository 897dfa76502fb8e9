#!/bin/bash
# kernel timeline of the last bench step: name, start (us from first), duration (us)
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/zltimeline; rm -rf $RAW; mkdir -p $RAW $REPO/gpurun_out
cd $REPO
rocprofv3 --kernel-trace --output-format csv -d $RAW/trace -- python3 bench.py $ARGS > $RAW/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$RAW/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "zl_k" in r["Kernel_Name"] and "interleave" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = last block of launches after the last big gap
last = rows[-int("${TL_ROWS:-40}"):]
t0 = int(last[0]["Start_Timestamp"])
out = []
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{r['Kernel_Name'][:28]:28s} start {((s - t0) / 1e3):9.1f} us  dur {((e - s) / 1e3):8.1f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}")
open("$REPO/gpurun_out/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
