#!/bin/bash
# HBM traffic of every extraction kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes over a stand-alone
# 4096^2 extraction (program directly after --), summed per kernel and per frame -> profiles/r02/traffic_akaze.json.
# FETCH_SIZE counts fabric read requests at 64 B; gfx950 issues 128-B requests for wide coalesced vector loads (MI355X_MICROARCH.md,
# "HBM / rocprofv3"), so the vector-load kernels' bytes are reported raw and x2 (upper bound: narrow or scalar requests are exact raw).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/traffic_akaze
TILE=${1:-4096}
REPS=${2:-3}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $R/tools/extract_one.py $TILE $REPS > $OUT/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $R/tools/extract_one.py $TILE $REPS > $OUT/w.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/extract_one.py $TILE 10 > $OUT/t.log 2>&1 || exit 1
python3 - "$OUT" "$TILE" "$REPS" <<'PY'
import collections, csv, glob, json, sys
out, tile, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = reps + 1
def sums(sub, counter):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    by = collections.defaultdict(float); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("apds::", "")
        by[name] += float(r["Counter_Value"]) * 1024.0   # FETCH_SIZE / WRITE_SIZE are in KiB
        calls[name] += 1
    return by, calls
fetch, calls = sums("f", "FETCH_SIZE")
write, _ = sums("w", "WRITE_SIZE")
times = {}
for f in glob.glob(f"{out}/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("apds::", "")
        times[name] = float(r["TotalDurationNs"]) / 11 / 1e3   # us per frame (11 frames in the trace run)
rows = []
for name in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    if name.startswith("at::") or "elementwise" in name: continue
    rows.append({"kernel": name, "launches_per_frame": calls[name] / frames, "fetch_raw_bytes_per_frame": fetch.get(name, 0) / frames,
                 "write_bytes_per_frame": write.get(name, 0) / frames, "us_per_frame": times.get(name)})
tot_f = sum(r["fetch_raw_bytes_per_frame"] for r in rows); tot_w = sum(r["write_bytes_per_frame"] for r in rows)
alg = {4096: 7.41e9, 1024: 0.463e9, 512: 0.111e9}.get(tile)
res = {"tile": tile, "frames_counted": frames, "fetch_raw_bytes_per_frame": tot_f, "fetch_x2_bytes_per_frame": 2 * tot_f, "write_bytes_per_frame": tot_w,
       "algorithmic_bytes_per_frame": alg,
       "traffic_over_algorithmic_raw": (tot_f + tot_w) / alg if alg else None, "traffic_over_algorithmic_fetch_x2": (2 * tot_f + tot_w) / alg if alg else None,
       "note": "FETCH_SIZE = TCC_EA0_RDREQ x 64 B; wide coalesced vector loads are 128-B requests on gfx950, so their true bytes are 2x raw (guide); scalar / narrow requests are exact raw. Both totals are given.",
       "source": "tools/traffic_akaze.sh: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --kernel-trace --stats, three separate runs of tools/extract_one.py",
       "kernels": rows}
json.dump(res, open(f"{out}/traffic_akaze.json", "w"), indent=1)
print(f"tile {tile}: fetch raw {tot_f/1e9:.3f} GB (x2 {2*tot_f/1e9:.3f}), write {tot_w/1e9:.3f} GB per frame; algorithmic {alg}")
for r in rows[:24]:
    print(f"  {r['kernel'][:60]:60s} {r['launches_per_frame']:6.1f} launches  fetch {r['fetch_raw_bytes_per_frame']/1e6:9.1f} MB raw  write {r['write_bytes_per_frame']/1e6:9.1f} MB  {r['us_per_frame'] or 0:8.1f} us")
PY
find $OUT -name "*counter_collection.csv" -size +20M -delete
cat $OUT/t.log | tail -2
