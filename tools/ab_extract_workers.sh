#!/bin/bash
# A/B of the number of extraction workers (and optionally other env knobs) on ONE box, interleaved repeats.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/ab_extract_workers.log
: > $OUT
for rep in 1 2 3; do
  for w in 1 2; do
    APDS_EXTRACT_WORKERS=$w timeout -k 10 200 python3 $R/bench.py --steps 50 --warmup 3 > /tmp/b.json 2>/tmp/b.err || { echo "fail w=$w" >> $OUT; tail -5 /tmp/b.err >> $OUT; exit 1; }
    python3 - $w $rep >> $OUT <<'PY'
import json, sys
j = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
st = j["stages_ms_per_step"]
print(f"workers {sys.argv[1]} rep {sys.argv[2]}: {j['value']:.2f} fps {j['ms_per_step']:.2f} ms  akaze {st['akaze_extract']:.2f} match {st['hamming_topk']:.2f} sample {st['hamming_topk_sample']:.2f}")
PY
  done
done
cat $OUT
