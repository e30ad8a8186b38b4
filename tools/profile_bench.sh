#!/bin/bash
# Collect the rocprofv3 evidence for the bench command (run on the GPU box through gpurun).
#   1) --kernel-trace --stats of `python bench.py`  -> per-kernel time summary
#   2) separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> HBM traffic of the dominant kernel
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu-baseline"
# serial run first: per-kernel durations without cross-stage overlap (the streamed run stretches the short kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 $R/bench.py --serial $ARGS > $OUT/bench_stats_serial.log 2>&1
echo "serial stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.log 2>&1
echo "stats rc=$?"
# counter passes run --serial: TCC counters are device-wide, so with the stages overlapped the extraction's HBM traffic would be
# charged to whichever match launch it ran under
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
echo "write rc=$?"
find $OUT -name "*.csv" | head -20
python3 $R/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
mkdir -p $OUT/tmp_serial && cp -r $OUT/stats_serial $OUT/tmp_serial/stats && python3 $R/tools/summarize_profile.py $OUT/tmp_serial > $OUT/summary_serial.txt 2>&1; rm -rf $OUT/tmp_serial
cat $OUT/summary.txt | head -60
# keep the merged output small: drop the raw per-dispatch traces after summarising
find $OUT -name "*kernel_trace.csv" -size +20M -delete
