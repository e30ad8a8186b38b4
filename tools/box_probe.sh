#!/bin/bash
# The default streamed bench, three times, on whatever box this lands on (run-to-run and box-to-box spread).
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  python3 $R/bench.py --no-cpu-baseline > /tmp/bp.json 2>/dev/null
  python3 - $i <<'PY'
import json, sys
j = json.load(open("/tmp/bp.json"))
print("run", sys.argv[1], round(j["value"], 3), "fps", round(j["ms_per_step"], 3), "ms", {k: round(v, 2) for k, v in j["stages_ms_per_step"].items()}, j["config"]["match_occupancy_cap"], "gap", j["config"].get("match_stream_gap_ms"), j["config"].get("match_stream_gaps_ms_first16"), flush=True)
PY
done
