cd /tmp && export TMPDIR=/tmp
RAW=/tmp/zlrt; rm -rf $RAW; mkdir -p $RAW
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $RAW/trace -- python3 scripts/realtime_latency.py > $RAW/log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$RAW/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "zl_k" in r["Kernel_Name"] and "interleave" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# first engine shape (96 voices): rows 3*100 .. 3*110
sel = rows[300:318]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:26]:26s} start {((s - t0) / 1e3):8.1f} us  dur {((e - s) / 1e3):6.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
PY
tail -5 $RAW/log
