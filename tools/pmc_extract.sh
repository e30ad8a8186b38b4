#!/bin/bash
# Where do the extraction kernels' wave cycles go? SQ counters (own --pmc run) per kernel, for the largest launches of each (level size 4096^2).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_extract
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/bench.py --serial --steps 1 --warmup 1 --frames 1 --no-cpu-baseline > $OUT/a.log 2>&1
echo "pass a rc=$?"
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $R/bench.py --serial --steps 1 --warmup 1 --frames 1 --no-cpu-baseline > $OUT/b.log 2>&1
echo "pass b rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    fs = glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counter file"); continue
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("apds::", "")
        disp[(name, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    by = collections.defaultdict(list)
    for (name, _), c in disp.items():
        by[name].append(c)
    for name in ("doh_strip_kernel<3", "doh_strip_kernel<4", "doh_fused_kernel<3", "doh_fused_kernel<4", "level_strip_kernel<3", "level_strip_kernel<4", "level_fused_kernel<512", "level_fused_kernel<1024",
                 "nld_strip_kernel<3", "mldb_kernel", "orientation_kernel", "suppress_round_kernel", "base_strip_kernel"):
        ds = [c for k, v in by.items() if k.startswith(name) for c in v]   # every instantiation whose name starts so
        if not ds:
            continue
        ds.sort(key=lambda c: -c.get("GRBM_GUI_ACTIVE", 0))
        top = ds[:max(1, len(ds) // 5)]
        avg = {k: sum(c.get(k, 0) for c in top) / len(top) for k in top[0]}
        cyc = avg.get("GRBM_GUI_ACTIVE", 0) / 8
        print(f"{name}: {len(top)} largest launches, {cyc / 2.4e3:.1f} us")
        for k, v in sorted(avg.items()):
            extra = ""
            if k.startswith("SQ_WAIT") or k.startswith("SQ_ACTIVE_INST_ANY"):
                extra = f"  ({v / max(avg.get('SQ_WAVE_CYCLES', 1), 1):.3f} of wave cycles)" if "SQ_WAVE_CYCLES" in avg else ""
            if k == "SQ_ACTIVE_INST_VALU" and cyc:
                extra = f"  (VALUBusy {v * 4 / 1024 / cyc:.3f})"
            if k == "SQ_ACTIVE_INST_LDS" and cyc:
                extra = f"  (per CU-cycle {v * 4 / 256 / cyc:.3f})"
            print(f"    {k:26s} {v:14.4g}{extra}")
PY
find $OUT -name "*counter_collection.csv" -size +20M -delete
