#!/bin/bash
# A/B of differently built libraries (cubesat-apds_amd/libapds_hip_<tag>.so) on stand-alone extraction
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for tag in "" _w4 _w9; do
  echo "== lib$tag (rep $rep)"; APDS_LIB_PATH=$R/cubesat-apds_amd/libapds_hip$tag.so python3 tools/extract_probe.py 4096 2048 2>/dev/null | grep tile
done; done
