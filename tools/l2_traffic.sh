#!/bin/bash
# HBM-side fetch traffic of the L2-match kernel (FETCH_SIZE, own --pmc run)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/l2_traffic
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $R/bench.py --workload l2 --steps 1 --warmup 0 --l2-queries 262144 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot = 0; n = 0
for r in csv.DictReader(open(f)):
    if "l2_topk_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        tot += float(r["Counter_Value"]); n += 1
print(f"l2_topk_kernel launches {n}: FETCH_SIZE {tot:.4g} KiB raw -> {tot*1024*2/1e9:.2f} GB (x2 gfx950 correction); algorithmic: train 0.512 GB + queries 0.134 GB per launch")
PY
grep roofline $OUT/log.txt | cut -c1-300
