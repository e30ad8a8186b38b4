#!/bin/bash
# rocprofv3 kernel stats of tools/ransac_probe.py, summary only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ransac_prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/ransac_probe.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'kernel':64s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s}")
for r in rows[:10]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete
grep "^n=" $OUT/log.txt
