"""Stand-alone extraction of resident tiles, nothing else on the GPU: the command behind the per-kernel traffic / time tables.
usage: extract_one.py [tile=4096] [reps=4]"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
pl = importlib.import_module("cubesat-apds_amd.pipeline")
L = pkg._lib.lib()
check = pkg._lib.check
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cap = (1 << 18) - 1
kps = torch.empty((cap, 7), dtype=torch.float32, device=dev)
desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
frames = [torch.from_numpy(pkg.synth.make_tile(T, T, frame_index=i)).to(dev) for i in range(2)]
n = C.c_int(0)
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    for rep in range(reps + 1):
        if rep == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        f = frames[rep % 2]
        check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
    torch.cuda.synchronize()
print(f"tile {T}^2: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per extraction over {reps} (after 1 warm-up), {n.value} keypoints", flush=True)
