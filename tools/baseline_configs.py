"""One line per BASELINE.json config that fits one GPU (configs[0] .. configs[3]; configs[4] needs the 8-GPU node: `python3 bench.py --gpus 8
--db-rows 10000000`). bench.py's default line is the headline (4096^2 frame vs 1 M rows); this script runs the other configurations' shapes
through the same entry points so that every config has a measured line under profiles/.
    python3 tools/baseline_configs.py > gpurun_out/r03/baseline_configs.jsonl"""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
pl = importlib.import_module("cubesat-apds_amd.pipeline")
L, check, ptr = pkg._lib.lib(), pkg._lib.check, pkg._lib.ptr
dev = torch.device("cuda:0")


def bench_line(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT, timeout=900)
    return json.loads(out.stdout.strip().splitlines()[-1])


def emit(config, what, **kw):
    print(json.dumps(dict(config=config, what=what, **kw)), flush=True)


# configs[0]: one 512^2 tile, extraction + brute-force match against a 10 k-descriptor DB (the reference's CPU-runnable case), here on the GPU
d = bench_line("--tile", "512", "--db-rows", "10000", "--steps", "200", "--warmup", "20", "--no-cpu-baseline")
emit(0, "512^2 tile: detect+describe -> Hamming top-2 vs 10 000 rows -> ratio test -> RANSAC homography, streamed", frames_per_s=round(d["value"], 1),
     ms_per_step=round(d["ms_per_step"], 3), keypoints_per_frame=d["config"].get("keypoints_per_frame"), homography_found=d["config"].get("homography_found"))

# configs[1]: 4096^2 detect + describe, resident frame, stand-alone
cap = (1 << 18) - 1
kps = torch.empty((cap, 7), dtype=torch.float32, device=dev)
desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
frames = [torch.from_numpy(pkg.synth.make_tile(4096, 4096, frame_index=i)).to(dev) for i in range(2)]
n = C.c_int(0)
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    for rep in range(3):
        check(L.apds_dev_akaze_extract(frames[rep % 2].data_ptr(), 4096, 4096, 4, frames[0].stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        check(L.apds_dev_akaze_extract(frames[rep % 2].data_ptr(), 4096, 4096, 4, frames[0].stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
emit(1, "4096^2 BGRA frame: detect + describe, frame resident", ms_per_frame=round(ms, 3), keypoints=n.value, mpx_per_s=round(4096 * 4096 / ms / 1e3, 1),
     frac_of_hbm_roofline=round(7.41e9 / (ms * 1e-3) / 8e12, 3))
del frames, kps, desc

# configs[2]: a batch of 256 tiles of 1024^2 in one call, then the float-descriptor L2 match vs 1 M x 128 (bench.py --workload l2, both modes)
B, T, capb = 256, 1024, 8192
tiles = np.stack([pkg.synth.make_tile(T, T, frame_index=100 + i) for i in range(8)])
dimg = torch.from_numpy(np.ascontiguousarray(np.concatenate([tiles] * (B // 8)))).to(dev)
kb = torch.empty((B, capb, 7), dtype=torch.float32, device=dev)
db = torch.empty((B, capb, 64), dtype=torch.uint8, device=dev)
counts = (C.c_int * B)()
with torch.cuda.stream(st):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(L.apds_dev_akaze_extract_batch(dimg.data_ptr(), B, dimg.stride(0), T, T, 4, dimg.stride(1), capb, kb.data_ptr(), db.data_ptr(), capb, counts, pl.torch_stream()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
emit(2, "batch of 256 tiles of 1024^2 in one apds_dev_akaze_extract_batch call, tiles resident", ms_per_batch=round(dt * 1e3, 2), tiles_per_s=round(B / dt, 1),
     keypoints_total=int(sum(counts)), mpx_per_s=round(B * T * T / dt / 1e6, 1), frac_of_hbm_roofline=round(B * T * T / 16777216 * 7.41e9 / dt / 8e12, 3))
del dimg, kb, db
torch.cuda.empty_cache()
for mode in ("exact", "screen"):
    d = bench_line("--workload", "l2", "--l2-mode", mode, "--steps", "2", "--warmup", "1")
    emit(2, f"L2 top-2 of 2^20 x 128 float queries vs 1 M x 128 rows, mode {mode}", ms_per_step=round(d["ms_per_step"], 1), roofline=d["roofline"].get("frac"),
         bound=d["roofline"].get("bound"), kernel=d["roofline"].get("kernel"))

# configs[3]: RANSAC on 50 k tentative matches, the full 4096-hypothesis budget (confidence ~ 1), homography against the oracle
import oracle  # noqa: E402

src, dst, Ht, inl = pkg.synth.make_ransac_set(50000, inlier_frac=0.4)
H, mask = np.zeros(9), np.zeros(50000, np.uint8)
L.apds_find_homography_ex(ptr(src), ptr(dst), 50000, 8, 3.0, 4096, 0.9999999999, ptr(H), ptr(mask))
t0 = time.perf_counter()
for _ in range(5):
    L.apds_find_homography_ex(ptr(src), ptr(dst), 50000, 8, 3.0, 4096, 0.9999999999, ptr(H), ptr(mask))
tg = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
ok, Ho, mo = oracle.find_homography(src, dst, 8, 3.0, 4096, 0.9999999999)
to = time.perf_counter() - t0
emit(3, "RANSAC homography, 50 000 pairs, 4096 hypotheses scored (host pointers: PCIe of the points included)", ms=round(tg * 1e3, 2), oracle_ms=round(to * 1e3, 1),
     point_evals_per_s=round(4096 * 50000 / tg / 1e9, 2), unit="G point-evaluations/s", mask_equal_oracle=bool(np.array_equal(mask, mo)),
     H_equal_oracle=bool(np.array_equal(H.reshape(3, 3), Ho.reshape(3, 3))), inliers=int(mask.sum()))
