#!/bin/bash
# HBM-side fetch of hamming_mfma_kernel per launch (rocprofv3 --pmc FETCH_SIZE in a run of its own), for the shapes of tools/match_mfma_probe.py.
# usage: tools/pmc_match_mfma.sh <label>   (environment switches such as APDS_MATCH_MFMA_XCD are inherited)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_mfma_$1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $R/tools/match_mfma_probe.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" "$1" <<'PY'
import csv, glob, sys, collections
out, label = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
for f in glob.glob(out + "/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hamming_mfma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            k = r["Grid_Size"]
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    # KiB, 64 B per request: doubled for this kernel's 16-byte-per-lane loads (MI355X_MICROARCH.md)
    print(f"{label}: grid {k:>8s} threads: {len(v)} launches, fetch {2 * sum(v) / len(v) * 1024 / 1e6:9.1f} MB per launch (counter x 2)")
PY
grep mfma $OUT/run.log
find $OUT -name "*.csv" -delete
