#!/bin/bash
# A/B: keypoint stages of the large octaves under the small octaves' chain, with an occupancy cap on their per-keypoint kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" python3 tools/extract_probe.py 4096 2>/dev/null | grep tile; }
run APDS_AKAZE_STAGES=0
run APDS_AKAZE_STAGES=1
run APDS_AKAZE_STAGES=1 APDS_STAGE_LDS_PAD=17000
run APDS_AKAZE_STAGES=1 APDS_STAGE_LDS_PAD=30000
run APDS_AKAZE_STAGES=1 APDS_STAGE_LDS_PAD=57000
run APDS_AKAZE_STAGES=0
