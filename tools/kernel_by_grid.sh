#!/bin/bash
# rocprofv3 kernel trace of a serial bench run, durations grouped by (kernel, grid size): which launch sizes of a kernel cost what
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kbg
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --serial --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("apds::", "")
    key = (name, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = collections.defaultdict(float)
for (n, g), (c, t) in agg.items():
    tot[n] += t
for n in sorted(tot, key=lambda k: -tot[k])[:8]:
    print(f"{n}  total {tot[n]/1e3:.3f} ms")
    for (nn, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[0][1]):
        if nn == n:
            print(f"    grid_threads {g:>10d}  calls {c:>5d}  avg {t/c:9.2f} us  total {t/1e3:8.3f} ms")
PY
find $OUT -name "*kernel_trace.csv" -delete
