"""Small-tile throughput: tiles/s of 512^2 and 1024^2 BGRA tiles, ONE host thread, resident images:
unbatched apds_dev_akaze_extract in a loop against apds_dev_akaze_extract_batch for several batch sizes; and the host-pointer
forms (apds_akaze_extract in a loop / apds_akaze_extract_batch), which include the PCIe copies of images and results."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
pl = importlib.import_module("cubesat-apds_amd.pipeline")
L, check = pkg._lib.lib(), pkg._lib.check
fe = pkg.feature_extraction
dev = torch.device("cuda:0")
res = []
for T in (512, 1024):
    tiles = np.stack([pkg.synth.make_tile(T, T, frame_index=100 + i) for i in range(8)])
    NB = 32
    host = np.ascontiguousarray(np.concatenate([tiles] * (NB // 8)))
    d = torch.from_numpy(host).to(dev)
    cap = 16384 if T == 512 else 32768
    kps = torch.empty((NB, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((NB, cap, 64), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(dev)
    row = {"tile": T}
    with torch.cuda.stream(st):
        n = C.c_int(0)
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(NB):
                check(L.apds_dev_akaze_extract(d[i].data_ptr(), T, T, 4, d.stride(1), cap, kps[i].data_ptr(), desc[i].data_ptr(), cap, C.byref(n), pl.torch_stream()))
            torch.cuda.synchronize()
            row["unbatched_resident_tiles_per_s"] = round(NB / (time.perf_counter() - t0), 1)
        for B in (2, 4, 8, 16, 32):
            counts = (C.c_int * B)()
            for rep in range(3):
                t0 = time.perf_counter()
                for i in range(0, NB, B):
                    check(L.apds_dev_akaze_extract_batch(d[i].data_ptr(), B, d.stride(0), T, T, 4, d.stride(1), cap, kps[i].data_ptr(), desc[i].data_ptr(), cap, counts,
                                                         pl.torch_stream()))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            row[f"batch{B}_resident_tiles_per_s"] = round(NB / dt, 1)
        row["keypoints_per_tile"] = int(np.mean(list(counts)))
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(NB):
            fe.akaze_keypoint_descriptor_extraction_def(host[i], None)
        row["unbatched_host_api_tiles_per_s"] = round(NB / (time.perf_counter() - t0), 1)
    for B in (8, 32):
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(0, NB, B):
                fe.akaze_keypoint_descriptor_extraction_batch(host[i:i + B], None)
            row[f"batch{B}_host_api_tiles_per_s"] = round(NB / (time.perf_counter() - t0), 1)
    print(json.dumps(row), flush=True)
    res.append(row)
    del d, kps, desc
