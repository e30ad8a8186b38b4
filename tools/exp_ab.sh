#!/bin/bash
# A/B of the APDS_EXP experiment bits on one box: stand-alone 4096^2 extraction time per variant (two alternating rounds), then parity of the
# listed variant. usage: tools/exp_ab.sh "<exp values>" <parity exp> [out]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${3:-$R/gpurun_out/r04/exp_ab.txt}
mkdir -p $(dirname $OUT)
: > $OUT
for round in 1 2 3; do
  for e in $1; do
    echo -n "round $round APDS_EXP=$e  " >> $OUT
    APDS_EXP=$e python3 $R/tools/extract_probe.py 4096 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
if [ -n "$2" ]; then
  APDS_EXP=$2 python3 -m pytest $R/tests/test_akaze_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3 | tee -a $OUT
  APDS_EXP=$2 APDS_LEVEL_FUSE=2 APDS_LEVEL_STRIP=0 python3 -m pytest $R/tests/test_akaze_gpu.py "$R/tests/test_fuzz_gpu.py::test_akaze_random_shapes" -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3 | tee -a $OUT
fi
