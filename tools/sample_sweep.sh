#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout -k 10 200 python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  fps', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), d['stages_ms_per_step'])" || echo failed; }
for smp in 16384 8192 4096; do for mr in 256 512; do for st in 1 2; do
  echo "== SAMPLE=$smp SAMPLE_MIN_ROWS=$mr SAMPLE_T=$st"
  APDS_MATCH_SAMPLE=$smp APDS_MATCH_SAMPLE_MIN_ROWS=$mr APDS_MATCH_SAMPLE_T=$st run
done; done; done
