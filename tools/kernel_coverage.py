"""Kernels of csrc/ that the traced GPU test run (tools/kernel_coverage.sh) never launched: dead code or a test gap."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
launched = set()
for line in open(os.path.join(ROOT, "gpurun_out", "kernel_coverage", "launched_kernels.txt")):
    name = line.strip().split(None, 1)[1] if len(line.strip().split(None, 1)) > 1 else ""
    m = re.search(r"(?:apds::)?([A-Za-z_0-9]+)(?:<|\()", name)
    if m:
        launched.add(m.group(1))
defined = {}
for f in sorted(glob.glob(os.path.join(ROOT, "cubesat-apds_amd", "csrc", "*.hip"))):
    src = open(f).read()
    for m in re.finditer(r"__global__[^;{]*?\bvoid\s+([A-Za-z_0-9]+)\s*\(", src, re.S):
        defined.setdefault(m.group(1), os.path.basename(f))
never = sorted(k for k in defined if k not in launched)
print(f"{len(defined)} kernels defined, {len(defined) - len(never)} launched by the GPU tests; never launched:")
for k in never:
    print(f"  {k}  ({defined[k]})")
