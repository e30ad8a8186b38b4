#!/bin/bash
# GPU sweep of the match kernel's work-item size (not part of the test suite)
export APDS_VALU_PROBE=1
for tw in 12288 24576 49152 98304 196608; do
  for mr in 512 1024; do
    echo "== target_waves=$tw min_rows=$mr"
    APDS_MATCH_TARGET_WAVES=$tw APDS_MATCH_MIN_ROWS=$mr python tools/match_probe.py 2>&1 | grep -E "nq\": 20000, \"nt\": 1000000|nq\": 5000|VALU lane" | cut -c1-200
    unset APDS_VALU_PROBE
  done
done
for t in 1 2 4; do echo "== T=$t"; APDS_MATCH_T=$t python tools/match_probe.py 2>&1 | grep -E "nq\": 20000, \"nt\": 1000000" | cut -c1-200; done
