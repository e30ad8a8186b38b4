"""Stand-alone extraction time per tile size (wall clock around 20 back-to-back calls on resident tiles, no other stage on the GPU)."""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
from importlib import import_module
pl = import_module("cubesat-apds_amd.pipeline")
L = pkg._lib.lib()
check = pkg._lib.check
dev = torch.device("cuda:0")
cap = (1 << 18) - 1
kps = torch.empty((cap, 7), dtype=torch.float32, device=dev)
desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
for T in ([int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096]):
    frames = [torch.from_numpy(pkg.synth.make_tile(T, T, frame_index=i)).to(dev) for i in range(2)]
    n = C.c_int(0)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        for rep in range(3):
            f = frames[rep % 2]
            check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for rep in range(reps):
            f = frames[rep % 2]
            check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"tile {T:5d}^2: {ms:7.3f} ms per extraction, {n.value} keypoints, {T * T / ms / 1e3:8.1f} Mpx/s", flush=True)
