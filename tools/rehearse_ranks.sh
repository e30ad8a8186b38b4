#!/bin/bash
# The multi-rank bench path on the ONE-GPU box: (1) RCCL itself at world size 1 (tools/nccl_selftest.py: the matcher's collectives on
# device tensors), (2) bench.py with 2 and with 4 ranks sharing the card, collectives over gloo (APDS_BENCH_BACKEND=gloo: device
# tensors staged through the host) - same choreography, threads, streams and buffers as over RCCL. At most 4 processes touch the GPU.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python3 tools/nccl_selftest.py > $O/nccl_selftest.log 2>&1; echo "selftest rc=$?"; tail -2 $O/nccl_selftest.log
for N in 2 4; do
  APDS_BENCH_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) \
      bench.py --gpus $N --steps 6 --warmup 2 --db-rows 400000 --tile 2048 > $O/rehearse_w$N.json 2> $O/rehearse_w$N.err
  echo "world $N rc=$?"; cut -c1-420 $O/rehearse_w$N.json; tail -2 $O/rehearse_w$N.err
done
