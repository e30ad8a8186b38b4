#!/bin/bash
# One gpurun call that regenerates the evidence under profiles/<round>/ (run on the GPU box; copy gpurun_out/evidence/* afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
rm -rf $O && mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --db real --no-cpu-baseline > $O/bench_db_real.json 2>/dev/null
python3 bench.py --workload l2 --steps 2 --warmup 1 > $O/bench_l2_exact.json 2>/dev/null
python3 bench.py --workload l2 --l2-mode screen --steps 3 --warmup 1 > $O/bench_l2_screen.json 2>/dev/null
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
cp gpurun_out/prof/summary.txt $O/bench_rocprofv3_summary.txt
cp gpurun_out/prof/summary_serial.txt $O/bench_serial_rocprofv3_summary.txt
cp gpurun_out/prof/traffic.json $O/traffic_hamming_topk.json
bash tools/traffic_akaze.sh 4096 3 > $O/traffic_akaze.log 2>&1
cp gpurun_out/traffic_akaze/traffic_akaze.json $O/traffic_akaze.json
python3 tools/extract_probe.py > $O/extract_probe.log 2>&1
python3 tools/batch_probe.py > $O/batch_probe.log 2>&1
python3 tools/extract_threads_probe.py > $O/extract_threads_probe.log 2>&1
python3 tools/tile_throughput.py > $O/tile_throughput.log 2>&1
python3 tools/match_probe.py > $O/match_probe.log 2>&1
python3 tools/extract_sweep.py > $O/extract_sweep.json 2> $O/extract_sweep.err
bash tools/trace_extract.sh 4096 > $O/trace_extract.log 2>&1
cp gpurun_out/trace_extract/timeline.txt $O/extract_timeline.txt
ls -la $O
