"""Times apds_pnp_solver_ransac (HIP) next to the oracle on the same synthetic correspondences (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
import oracle  # noqa: E402

hg = pkg.homographier
for n, frac, iters, conf in ((50000, 0.4, 4096, 0.9999999), (50000, 0.4, 2000, 0.995), (2000, 0.6, 1000, 0.99)):
    obj, img, K, rvec, tvec, inl = pkg.synth.make_pnp_set(n, inlier_frac=frac, noise=0.5)
    corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
    Kc = hg.Cmat(K, np.float64)
    hg.pnp_solver_ransac(corr[:100], Kc, 10, 3.0, 0.9)       # warm-up (module load, workspace)
    o = np.ascontiguousarray(obj)
    i2 = np.ascontiguousarray(img)
    rv, tv, inliers = np.zeros(3), np.zeros(3), np.zeros(n, np.int32)
    import ctypes as C
    ni, found = C.c_int(0), C.c_int(0)
    t0 = time.perf_counter()
    for _ in range(3):
        rc = pkg.lib().apds_pnp_solver_ransac(pkg._lib.ptr(o), pkg._lib.ptr(i2), n, pkg._lib.ptr(K), iters, 3.0, conf, 1, pkg._lib.ptr(rv), pkg._lib.ptr(tv),
                                              pkg._lib.ptr(inliers), C.byref(ni), C.byref(found))
    tg = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    rc2, r, t, idx = oracle.solve_pnp_ransac(obj, img, K, iters, 3.0, conf)
    to = time.perf_counter() - t0
    print(f"n={n} iters={iters} conf={conf}: hip {tg*1e3:.2f} ms  oracle {to*1e3:.2f} ms  inliers {ni.value}/{len(idx)} equal={np.array_equal(inliers[:ni.value], idx)} "
          f"pose_equal={np.array_equal(rv, r) and np.array_equal(tv, t)}")
    t0 = time.perf_counter()
    for _ in range(3):
        rc = pkg.lib().apds_pnp_solver_ransac(pkg._lib.ptr(o), pkg._lib.ptr(i2), n, pkg._lib.ptr(K), iters, 3.0, conf, 2, pkg._lib.ptr(rv), pkg._lib.ptr(tv),
                                              pkg._lib.ptr(inliers), C.byref(ni), C.byref(found))
    tg = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    rc2, r, t, idx = oracle.solve_pnp_ransac(obj, img, K, iters, 3.0, conf, method=2)
    to = time.perf_counter() - t0
    print(f"   P3P kernel: hip {tg*1e3:.2f} ms  oracle {to*1e3:.2f} ms  inliers {ni.value}/{len(idx)} equal={np.array_equal(inliers[:ni.value], idx)} "
          f"pose_equal={np.array_equal(rv, r) and np.array_equal(tv, t)}")
