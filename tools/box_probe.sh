#!/bin/bash
# Streamed bench under a few settings, interleaved, on whatever box this lands on.
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 2" "40000 2" "33000 2" "0 2" "40000 2"; do
  set -- $cfg
  APDS_MATCH_LDS_CAP=$1 APDS_EXTRACT_WORKERS=$2 python3 $R/bench.py --no-cpu-baseline > /tmp/bp.json 2>/dev/null
  python3 - $1 $2 <<'PY'
import json, sys
j = json.load(open("/tmp/bp.json"))
print("cap", sys.argv[1], "workers", sys.argv[2], round(j["value"], 3), "fps", round(j["ms_per_step"], 3), "ms", {k: round(v, 2) for k, v in j["stages_ms_per_step"].items()}, "gap", j["config"].get("match_stream_gap_ms"), flush=True)
PY
done
