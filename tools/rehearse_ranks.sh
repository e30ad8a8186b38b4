#!/bin/bash
# The multi-rank bench path on the ONE-GPU box, started exactly as the driver starts it (bare `python3 bench.py --gpus N`: the parent
# spawns the ranks): 2 and 4 ranks sharing the card, the exchange step through the C ABI's host-callback transport over gloo
# (APDS_BENCH_BACKEND=gloo) - same choreography (csrc/shard_core.h), threads, streams and buffers as over RCCL. At most 4 processes touch
# the GPU. RCCL itself at world size 1: tools/nccl_selftest.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python3 tools/nccl_selftest.py > $O/nccl_selftest.log 2>&1; echo "selftest rc=$?"; tail -2 $O/nccl_selftest.log
for N in 2 4; do
  APDS_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus $N --steps 6 --warmup 2 --db-rows 400000 --tile 2048 --no-cpu-baseline > $O/rehearse_gloo_world$N.json 2> $O/rehearse_w$N.err
  echo "world $N rc=$?"; cut -c1-300 $O/rehearse_gloo_world$N.json; tail -2 $O/rehearse_w$N.err
done
