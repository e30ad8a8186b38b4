#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 0 4 3 2; do
  echo "== APDS_MATCH_PERSIST=$p (serial / streamed)"
  for mode in --serial ""; do
  APDS_MATCH_PERSIST=$p timeout -k 10 200 python3 $R/bench.py $mode --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  fps', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), d['stages_ms_per_step'], 'valu', round(d['valu']['frac_of_measured'],3))" || echo failed
  done
done
