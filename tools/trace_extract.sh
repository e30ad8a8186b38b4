#!/bin/bash
# kernel timeline of one stand-alone extraction (start / end / queue per kernel of the last frame) -> gpurun_out/trace_extract/timeline.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_extract
TILE=${1:-4096}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/extract_one.py $TILE 6 > $OUT/t.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/t/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "apds" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "base_strip_kernel" in r["Kernel_Name"] or "gray_kernel" in r["Kernel_Name"]]
i0, i1 = starts[-2], starts[-1]
# the frame starts with the memset before the base kernel
while i0 > 0 and "rocclr" in rows[i0 - 1]["Kernel_Name"] and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 20000: i0 -= 1
while i1 > 0 and "rocclr" in rows[i1 - 1]["Kernel_Name"] and int(rows[i1]["Start_Timestamp"]) - int(rows[i1 - 1]["End_Timestamp"]) < 20000: i1 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
with open(out + "/timeline.txt", "w") as o:
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("apds::", "")[:44]
        o.write(f"{s/1e3:8.1f} {e/1e3:8.1f} dur {(e-s)/1e3:6.1f} q{r.get('Queue_Id','?'):>3s} grid {r.get('Grid_Size_X','?'):>8s}x{r.get('Grid_Size_Y','?'):>4s} {name}\n")
    o.write(f"frame span us: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3}\n")
print(open(out + "/timeline.txt").read()[-6000:])
PY
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/t.log
