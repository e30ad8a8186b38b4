#!/bin/bash
# same-box A/B of environment switches: tools/ab_env.sh "<VAR=a VAR=b ...>" [rounds] -> stand-alone 4096^2 extraction time per setting
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${3:-$R/gpurun_out/r04/ab_env.txt}
mkdir -p $(dirname $OUT)
for round in $(seq 1 ${2:-3}); do
  for kv in $1; do
    echo -n "round $round $kv  " >> $OUT
    env $kv python3 $R/tools/extract_probe.py 4096 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
