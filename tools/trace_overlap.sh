#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
ev = []
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, n, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
ev.sort()
t0 = ev[0][0]
big = [x for x in ev if "hamming_topk" in x[2] and (x[1]-x[0]) > 5e6]
print("main match launches:")
for s, e, n, q, st in big:
    print(f"  start {(s-t0)/1e6:9.2f} ms  dur {(e-s)/1e6:7.2f} ms  queue {q} stream {st}")
# what else ran during the last main launch
s0, e0 = big[-2][0], big[-2][1]
inside = [x for x in ev if x[0] >= s0 and x[1] <= e0 and "hamming" not in x[2]]
print("kernels fully inside match launch:", len(inside), "busy ms", sum(x[1]-x[0] for x in inside)/1e6)
gap = [x for x in ev if x[0] >= e0 and x[0] < big[-1][0]]
print("timeline after match end (first 40 kernels by start):")
for x in gap[:40]:
    print(f"   +{(x[0]-e0)/1e3:9.1f} us  dur {(x[1]-x[0])/1e3:8.1f} us  q{x[3]} s{x[4]}  {x[2][:60]}")
print("kernels between this match end and next match start:", len(gap), "span ms", (big[-1][0]-e0)/1e6)
from collections import Counter
c = Counter()
for x in gap: c[x[2][:40]] += (x[1]-x[0])/1e6
for k, v in c.most_common(12): print(f"   {k:42s} {v:8.3f} ms")
PY
find $OUT -name "*kernel_trace.csv" -delete
