#!/bin/bash
# One gpurun call that regenerates everything under profiles/<round>/ (run on the GPU box).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
rm -rf $O && mkdir -p $O
cd $R
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
cp gpurun_out/prof/summary.txt $O/bench_rocprofv3_summary.txt
cp gpurun_out/prof/summary_serial.txt $O/bench_serial_rocprofv3_summary.txt
cp gpurun_out/prof/stats_serial/*/*_kernel_stats.csv $O/bench_serial_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof/stats/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof/traffic.json $O/traffic_hamming_topk.json
bash tools/xcd_ab.sh > $O/match_xcd_placement_ab.log 2>&1
bash tools/pmc_util.sh > $O/pmc_utilisation.log 2>&1
bash tools/pnp_stats.sh > $O/pnp_probe.log 2>&1
bash tools/ransac_stats.sh > $O/ransac_probe.log 2>&1
cd $R
python3 tools/match_probe.py > $O/match_probe.log 2>&1
python3 tools/l2_probe.py > $O/l2_probe.log 2>&1
python3 bench.py --serial --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_serial.json 2>/dev/null
python3 bench.py --workload l2 --steps 2 --warmup 1 > $O/bench_l2.json 2>/dev/null
python3 bench.py > $O/bench_default.json 2>/dev/null
ls -la $O
