#!/bin/bash
# per-launch durations (in launch order) of kernels matching $1 during one serial bench step
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kseq
rm -rf $OUT && mkdir -p $OUT
PAT=$1; shift
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --serial --steps 1 --warmup 1 --frames 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; durations us:", " ".join(f"{x:.1f}" for x in d[-60:]))
PY
find $OUT -name "*kernel_trace.csv" -delete
