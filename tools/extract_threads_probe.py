"""Aggregate extraction throughput with N host threads calling the C ABI concurrently (each thread: own stream + workspace).
Two legs: the host-pointer entry (every call uploads its 4 T^2-byte image from pageable memory and downloads the result) and the
resident entry (image and outputs stay on the device): the difference is what PCIe and the staging copies cost."""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch   # before the library is loaded: torch brings its own HIP runtime, and a process must not initialise two

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
L = pkg._lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imgs = [np.ascontiguousarray(pkg.synth.make_tile(T, T, frame_index=i)) for i in range(4)]
reps = 24


def worker(k, out, ends):
    for r in range(reps + 4):
        if r == 4:
            barrier.wait()
            out[k] = time.perf_counter()
        img = imgs[(k + r) % 4]
        kps, desc, n, nb = C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_int(0)
        pkg._lib.check(L.apds_akaze_extract(img.ctypes.data, T, T, 4, img.strides[0], 0, C.byref(kps), C.byref(desc), C.byref(n), C.byref(nb)))
        L.apds_free(kps)
        L.apds_free(desc)
    ends[k] = time.perf_counter()   # (the thread's release — stream teardown, workspace to the cache — is not extraction time)
    L.apds_thread_release()


dev = torch.device("cuda:0")
d_imgs = [torch.from_numpy(i).to(dev) for i in imgs]
CAP = (1 << 18) - 1


def worker_resident(k, out, ends):
    kps = torch.empty((CAP, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((CAP, 64), dtype=torch.uint8, device=dev)
    n = C.c_int(0)
    for r in range(reps + 4):
        if r == 4:
            barrier.wait()
            out[k] = time.perf_counter()
        img = d_imgs[(k + r) % 4]
        pkg._lib.check(L.apds_dev_akaze_extract(img.data_ptr(), T, T, 4, img.stride(0), CAP, kps.data_ptr(), desc.data_ptr(), CAP, C.byref(n), None))
    ends[k] = time.perf_counter()
    L.apds_thread_release()


for leg, fn in (("host pointers", worker), ("resident", worker_resident)):
  for nthreads in (1, 2, 4, 8):
    barrier = threading.Barrier(nthreads)
    starts = [0.0] * nthreads
    ends = [0.0] * nthreads
    ts = [threading.Thread(target=fn, args=(k, starts, ends)) for k in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = max(ends) - min(starts)
    print(f"tile {T}^2, {leg}, {nthreads} threads: {nthreads * reps / dt:8.1f} extractions/s ({dt / reps * 1e3:.3f} ms per extraction per thread)", flush=True)
