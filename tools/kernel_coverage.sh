#!/bin/bash
# Which kernels does the GPU test suite launch? rocprofv3 --kernel-trace --stats around `pytest -m gpu` (children included), then the
# kernel names of all per-process stats files against the __global__ functions of csrc/ (tools/kernel_coverage.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kernel_coverage
rm -rf $OUT /tmp/kcov && mkdir -p $OUT /tmp/kcov
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kcov -- python3 -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1
echo "pytest under rocprofv3 rc=$?"
tail -3 $OUT/pytest.log
find /tmp/kcov -name "*kernel_stats.csv" | while read f; do cut -d, -f1-3 "$f"; done | grep -v '^"Name"' | sed 's/^"//' | awk -F'",' '{print $1}' | sort | uniq -c | sort -rn > $OUT/launched_kernels.txt
wc -l $OUT/launched_kernels.txt
du -sh /tmp/kcov | tail -1
