"""BASELINE config 4: homographier RANSAC on 50 k tentative matches with 4096 hypotheses scored (confidence ~1 forces the full budget),
HIP path next to the oracle; also the default call (2000 iterations, 0.995) as the frame pipeline issues it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
import oracle  # noqa: E402

L, ptr = pkg.lib(), pkg._lib.ptr
for n, frac, iters, conf in ((50000, 0.4, 4096, 0.9999999999), (50000, 0.15, 4096, 0.995), (50000, 0.4, 2000, 0.995), (26000, 0.99, 2000, 0.995)):
    src, dst, Ht, inl = pkg.synth.make_ransac_set(n, inlier_frac=frac)
    H, mask = np.zeros(9), np.zeros(n, np.uint8)
    L.apds_find_homography_ex(ptr(src), ptr(dst), n, 8, 3.0, iters, conf, ptr(H), ptr(mask))     # warm-up
    t0 = time.perf_counter()
    for _ in range(3):
        rc = L.apds_find_homography_ex(ptr(src), ptr(dst), n, 8, 3.0, iters, conf, ptr(H), ptr(mask))
    tg = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    ok, Ho, mo = oracle.find_homography(src, dst, 8, 3.0, iters, conf)
    to = time.perf_counter() - t0
    print(f"n={n} inlier_frac={frac} iters={iters} conf={conf}: hip {tg*1e3:.2f} ms (incl. PCIe of the points)  oracle {to*1e3:.1f} ms  "
          f"inliers {int(mask.sum())}/{int(mo.sum())} mask_equal={np.array_equal(mask, mo)} H_close={np.allclose(H.reshape(3,3), Ho, rtol=1e-6, atol=1e-8)}")
