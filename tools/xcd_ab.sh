#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for x in 0 1; do
  export APDS_MATCH_XCD=$x
  echo "== APDS_MATCH_XCD=$x"
  python3 $R/bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['stages_ms_per_step'], 'valu frac', d['valu']['frac_of_measured'])"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python3 - <<PY
import csv, glob
tot=n=0
for f in glob.glob("/tmp/pmc_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hamming_topk_kernel<4" in r["Kernel_Name"] and r["Counter_Name"]=="$c":   # the main launch only (the sample pass is instance <1,K>)
            tot+=float(r["Counter_Value"]); n+=1
print("$c per launch (KiB)", tot/max(n,1), "launches", n)
PY
  done
done
