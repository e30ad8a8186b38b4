#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for lay in tail head wordtop; do
for r in 16 32; do
  echo "== layout=$lay reserve_cus=$r"
  APDS_CU_MASK_LAYOUT=$lay timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --reserve-cus $r 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), d['stages_ms_per_step'])" || echo failed
done
done
