#!/bin/bash
# second evidence pass of a round: the bench lines and the profiles that depend on bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
mkdir -p $O
cd $R
python3 bench.py --host-frames > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --db real --no-cpu-baseline > $O/bench_db_real.json 2>/dev/null
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
cp gpurun_out/prof/summary.txt $O/bench_rocprofv3_summary.txt
cp gpurun_out/prof/summary_serial.txt $O/bench_serial_rocprofv3_summary.txt
cp gpurun_out/prof/traffic.json $O/traffic_hamming_topk.json
bash tools/pmc_extract.sh > $O/pmc_extract.log 2>&1
bash tools/traffic_akaze.sh 4096 3 > $O/traffic_akaze.log 2>&1
cp gpurun_out/traffic_akaze/traffic_akaze.json $O/traffic_akaze.json
ls -la $O
