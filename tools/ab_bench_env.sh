#!/bin/bash
# same-box A/B of environment switches on the default bench line: tools/ab_bench_env.sh "<VAR=a VAR=b ...>" [rounds] [out]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${3:-$R/gpurun_out/r04/ab_bench_env.txt}
mkdir -p $(dirname $OUT)
for round in $(seq 1 ${2:-2}); do
  for kv in $1; do
    echo -n "round $round $kv  " >> $OUT
    env $kv python3 $R/bench.py --no-cpu-baseline --steps 60 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],2), 'frames/s', round(d['ms_per_step'],3), 'ms/step; match live', round(r['avg_launch_ms'],3), 'solo', round((r.get('solo') or {}).get('avg_launch_ms') or 0,3), d['stages_ms_per_step'])" >> $OUT
  done
done
cat $OUT
