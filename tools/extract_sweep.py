"""The reference bench's shape (benchmarks/benches/feature_extraction.rs:14-46): AKAZE extraction of ONE image Lanczos-resized to
128, 256, 512, 1024, 2048, 4096, 8192 px squares, 3 channels, timed region = Mat construction (here: the host -> device copy) +
AKAZE create + detectAndCompute. Per size: GPU ms through the host-pointer C ABI (what the reference's bench times), GPU ms with
the image resident in HBM, oracle ms on the host cores, and keypoint-count / descriptor equality. One JSON -> stdout.
The source image is a 2 x 2 flip mosaic of one synthetic 4096^2 tile (the reference's Denmark_8192.png is absent)."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
pl = importlib.import_module("cubesat-apds_amd.pipeline")
import oracle  # noqa: E402  (checker + cpu timing only)

L, check = pkg._lib.lib(), pkg._lib.check
dev = torch.device("cuda:0")
cap = (1 << 18) - 1
kps = torch.empty((cap, 7), dtype=torch.float32, device=dev)
desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
t = pkg.synth.make_tile(4096, 4096, frame_index=5, channels=1)
big = np.ascontiguousarray(np.block([[t, t[:, ::-1]], [t[::-1], t[::-1, ::-1]]]))
nproc, _ = oracle.host_threads()
rows = []
for size in (128, 256, 512, 1024, 2048, 4096, 8192):
    g = big if size == 8192 else np.asarray(Image.fromarray(big).resize((size, size), Image.LANCZOS))
    img = np.ascontiguousarray(np.dstack([g, g, g]))
    fe = pkg.feature_extraction
    got = fe.akaze_keypoint_descriptor_extraction_def(img, None)            # warm-up (workspace growth) + the result to compare
    reps = 20 if size <= 2048 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        fe.akaze_keypoint_descriptor_extraction_def(img, None)
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    d_img = torch.from_numpy(img).to(dev)
    n = C.c_int(0)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        for rep in range(reps + 1):
            if rep == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            check(L.apds_dev_akaze_extract(d_img.data_ptr(), size, size, 3, d_img.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
        torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    ref = oracle.akaze(img)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    rows.append({"size": size, "keypoints_gpu": int(len(got.keypoints)), "keypoints_oracle": int(len(ref.keypoints)),
                 "descriptors_equal": bool(np.array_equal(got.descriptors, ref.descriptors)), "keypoints_equal": bool(np.array_equal(got.keypoints, ref.keypoints)),
                 "gpu_ms_host_pointer_api": round(host_ms, 3), "gpu_ms_resident": round(dev_ms, 3), "oracle_ms": round(cpu_ms, 1), "oracle_threads": nproc,
                 "mpx_per_s_resident": round(size * size / dev_ms / 1e3, 1)})
    print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    del d_img
print(json.dumps({"bench": "extract_features_from_image (reference benchmarks/benches/feature_extraction.rs:14-46)", "channels": 3,
                  "source": "2x2 flip mosaic of synth.make_tile(4096, frame 5), PIL Lanczos resize", "rows": rows}, indent=1))
