#!/bin/bash
# A/B: the streaming kernels on smaller levels (forced on every level) for 2048^2 and 1024^2 tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" python3 tools/extract_probe.py 2048 1024 4096 2>/dev/null | grep tile; }
for rep in 1 2; do
run APDS_LEVEL_STREAM=1 APDS_DOH_STRIP=1
run APDS_LEVEL_STREAM=2 APDS_DOH_STRIP=1
run APDS_LEVEL_STREAM=1 APDS_DOH_STRIP=2
run APDS_LEVEL_STREAM=2 APDS_DOH_STRIP=2
done
