#!/bin/bash
# The default streamed bench under different numbers of HIP hardware queues, on whatever box this lands on.
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in 4 8 16 4 16; do
  GPU_MAX_HW_QUEUES=$q python3 $R/bench.py --no-cpu-baseline > /tmp/bp.json 2>/dev/null
  python3 - $q <<'PY'
import json, sys
j = json.load(open("/tmp/bp.json"))
print("hwq", sys.argv[1], round(j["value"], 3), "fps", round(j["ms_per_step"], 3), "ms", {k: round(v, 2) for k, v in j["stages_ms_per_step"].items()}, "gap", j["config"].get("match_stream_gap_ms"), flush=True)
PY
done
