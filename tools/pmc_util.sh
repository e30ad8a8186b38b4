#!/bin/bash
# Utilisation counters (own runs, --pmc only): MFMA busy for the L2 match kernel, VALU busy for the Hamming match kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_util
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -E "MFMA|VALU_BUSY|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU\b|SQ_WAVE_CYCLES" $OUT/counters.txt | cut -c1-160 | sort -u | head -40
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/l2 -- python3 $R/bench.py --workload l2 --steps 1 --warmup 0 --l2-queries 262144 > $OUT/l2.log 2>&1
echo "l2 pmc rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/ham -- python3 $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $OUT/ham.log 2>&1
echo "hamming pmc rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub, pat in (("l2", "l2_topk_kernel"), ("ham", "hamming_topk_kernel<4")):
    fs = glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counter file"); continue
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    per = {k: v / max(n[k], 1) for k, v in agg.items()}
    print(sub, "per-launch sums over all XCDs:", {k: (per[k], n[k]) for k in per})
    gui = per.get("GRBM_GUI_ACTIVE", 0) / 8          # the counter is summed over the 8 XCDs
    if gui:
        print(f"   kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs) = {gui:.4g}  (= {gui / 2.4e6:.2f} ms at 2.4 GHz)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            print(f"   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 256 CUs x 4 SIMDs) = {per['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}")
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in per:
            print(f"   f32 MFMA flops = MOPS x 512 = {per['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512:.4g}")
        if "SQ_ACTIVE_INST_VALU" in per:
            print(f"   VALUBusy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / cycles = {per['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / gui:.3f}")
        if "SQ_INSTS_VALU" in per:
            print(f"   VALU wave-instructions = {per['SQ_INSTS_VALU']:.4g}")
PY
find $OUT -name "*counter_collection.csv" -size +20M -delete
