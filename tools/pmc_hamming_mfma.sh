#!/bin/bash
# Matrix-pipe occupancy and the clock hamming_mfma_kernel runs at (own run, --pmc only, then a plain kernel trace for the duration).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_hamming_mfma
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/tools/match_mfma_probe.py > $OUT/a.log 2>&1
echo "pmc a rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 $R/tools/match_mfma_probe.py > $OUT/b.log 2>&1
echo "pmc b rc=$?"
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/match_mfma_probe.py > $OUT/t.log 2>&1
echo "trace rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hamming_mfma_kernel" in r["Kernel_Name"]:
                per[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hamming_mfma_kernel" in r["Kernel_Name"]:
            g = str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)) if "Grid_Size_X" in r else r.get("Grid_Size")
            dur[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
shapes = {"518144": (35312, 983616)}
for g, c in per.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    d = sum(dur[g]) / len(dur[g]) if dur.get(g) else None
    print(f"grid {g}: {m}  duration {d}")
    if d and "GRBM_GUI_ACTIVE" in m:
        gui = m["GRBM_GUI_ACTIVE"] / 8
        print(f"   {d * 1e3:.3f} ms; GRBM_GUI_ACTIVE / 8 = {gui:.4g} cycles -> {gui / d / 1e9:.2f} GHz; MFMA busy per SIMD = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:.3f} of the kernel's cycles")
        if g in shapes:
            q, n = shapes[g]
            n_mfma = -(-q // 16) * -(-n // 16) * 4
            print(f"   algorithmic MFMAs {n_mfma:.4g} x 16 cycles / 1024 SIMDs / cycles = {n_mfma * 16 / 1024 / gui:.3f}")
PY
grep '"mfma"' $OUT/t.log | head -2
find $OUT -name "*.csv" -delete
