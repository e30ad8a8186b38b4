#!/bin/bash
# MFMA pipe occupancy and the clock the bf16 screen kernel actually runs at (own run, --pmc only): is the distance to the 2.5 PF figure
# pipe idle time, or cycles that are not there (the figure assumes 2.4 GHz)?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_l2_screen
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/bench.py --workload l2 --l2-mode screen --steps 1 --warmup 0 > $OUT/a.log 2>&1
echo "pmc rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --workload l2 --l2-mode screen --steps 1 --warmup 0 > $OUT/t.log 2>&1
echo "trace rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
fs = glob.glob(f"{out}/a/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0]))) if fs else []
print("counter columns:", list(rows[0].keys()) if rows else None)
per = collections.defaultdict(list)
for r in rows:
    if "l2_screen_kernel<1" in r["Kernel_Name"] or "l2_screen_kernelILi1" in r["Kernel_Name"]:
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: (sum(v) / len(v), len(v)) for k, v in per.items()})
dur = []
for f in glob.glob(f"{out}/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "l2_screen_kernel<1" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
if dur and per:
    d = sum(dur) / len(dur)
    gui = sum(per["GRBM_GUI_ACTIVE"]) / len(per["GRBM_GUI_ACTIVE"]) / 8
    mf = sum(per["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(per["SQ_VALU_MFMA_BUSY_CYCLES"])
    print(f"full pass: {d * 1e3:.1f} ms (kernel trace, no counters); GRBM_GUI_ACTIVE / 8 XCDs = {gui:.4g} cycles -> {gui / d / 1e9:.2f} GHz average clock")
    print(f"SQ_VALU_MFMA_BUSY_CYCLES = {mf:.4g}; per SIMD (1024) = {mf / 1024:.4g} = {mf / 1024 / gui:.3f} of the kernel's cycles")
    n_mfma = 1048576 / 16 * (1000000 / 16) * 4
    print(f"MFMAs issued (Q/16 x N/16 x K/32) = {n_mfma:.4g}; x 16 cycles = {n_mfma * 16:.4g} pipe cycles = {n_mfma * 16 / 1024 / gui:.3f} of the kernel's SIMD cycles")
PY
