"""Randomised parity run: HIP path (through the C ABI) against the oracle on shapes and contents no fixed test names.
    python3 tools/fuzz_parity.py [seconds] [seed] [large]
Extraction (random sizes / channels / strides / contents / max_points), Hamming k-NN (random shapes, descriptor lengths, k up to 40,
duplicated rows), Lowe-filtered and cross-checked match lists, findHomography (every method, random inlier shares, degenerate sets),
pnp_solver_ransac (EPnP / P3P / ITERATIVE, 4 .. 1500 correspondences), batched extraction, band_merger and warp_image_perspective.
Stops at the first difference with the case's parameters (exit code 1); prints the number of cases per family otherwise."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime per process: before the library)

pkg = importlib.import_module("cubesat-apds_amd")
import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fe, hg, L, ptr = pkg.feature_extraction, pkg.homographier, pkg.lib(), pkg._lib.ptr
oracle.set_threads(min(16, os.cpu_count() or 1))
counts = {"akaze": 0, "akaze_keypoints": 0, "knn": 0, "lists": 0, "homography": 0, "pnp": 0, "batch": 0, "ingest": 0}


def fail(family, params, what):
    print(f"MISMATCH in {family}: {what}\n  case: {params}", flush=True)
    sys.exit(1)


def image(h, w, ch, kind):
    if kind == 0:
        img = pkg.synth.make_tile(h, w, frame_index=int(rng.integers(0, 1000)), channels=ch)
    elif kind == 1:
        img = rng.integers(0, 256, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)
    elif kind == 2:   # smooth ramps + a few hard edges
        y, x = np.mgrid[0:h, 0:w]
        g = ((x * 3 + y * 5) % 256).astype(np.uint8)
        g[h // 3: h // 3 + 4] = 255
        g[:, w // 2: w // 2 + 3] = 0
        img = g if ch == 1 else np.dstack([g] * ch)
    elif kind == 3:   # sparse bright dots on black
        g = np.zeros((h, w), np.uint8)
        n = max(1, h * w // 400)
        g[rng.integers(0, h, n), rng.integers(0, w, n)] = rng.integers(100, 256, n)
        img = g if ch == 1 else np.dstack([g, g // 2, 255 - g][:ch] + ([np.full_like(g, 255)] if ch == 4 else []))
    else:             # constant
        img = np.full((h, w) if ch == 1 else (h, w, ch), int(rng.integers(0, 256)), np.uint8)
    return np.ascontiguousarray(img)


LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"   # only frames of 8 Mpx and more (the streaming kernels' sizes), few cases per minute
_large_tile = None


def case_akaze():
    global _large_tile
    big = rng.random() < 0.15
    h = int(rng.integers(600, 1400)) if big else int(rng.integers(8, 520))
    w = int(rng.integers(600, 1400)) if big else int(rng.integers(8, 520))
    ch = int(rng.choice([1, 3, 4]))
    kind = int(rng.integers(0, 5))
    if LARGE:
        h = int(rng.integers(2050, 3700))
        w = max(int(rng.integers(2050, 3700)), (1 << 23) // h + 1)
        kind = int(rng.integers(0, 4))
        if kind == 0:   # a random crop of one synthetic tile (the generator needs 15 s per 4096^2)
            if _large_tile is None:
                _large_tile = pkg.synth.make_tile(4096, 4096, frame_index=3, channels=1)
            y0, x0 = int(rng.integers(0, 4096 - h + 1)), int(rng.integers(0, 4096 - w + 1))
            g = _large_tile[y0:y0 + h, x0:x0 + w]
            ch = int(rng.choice([1, 3]))
            img0 = np.ascontiguousarray(g if ch == 1 else np.dstack([g, g, g]))
    mp = None if rng.random() < 0.7 else int(rng.integers(1, 400))
    params = dict(h=h, w=w, ch=ch, kind=kind, max_points=mp)
    img = img0 if (LARGE and kind == 0) else image(h, w, ch, kind)
    if rng.random() < 0.3:   # rows with padding behind them
        pad = int(rng.integers(1, 9)) * (1 if ch != 4 else 4)
        wide = np.zeros((h, w * ch + pad), np.uint8)
        wide[:, : w * ch] = img.reshape(h, w * ch)
        img = wide[:, : w * ch].reshape(img.shape)
        params["row_padding"] = pad
    got = fe.akaze_keypoint_descriptor_extraction_def(img, mp)
    ref = oracle.akaze(np.ascontiguousarray(img), mp if mp is not None else (1 << 18) - 1)
    if len(got.keypoints) != len(ref.keypoints):
        fail("akaze", params, f"{len(got.keypoints)} keypoints, oracle {len(ref.keypoints)}")
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        if not np.array_equal(got.keypoints[f], ref.keypoints[f]):
            fail("akaze", params, f"field {f} differs")
    if not np.array_equal(got.descriptors, ref.descriptors):
        fail("akaze", params, "descriptors differ")
    counts["akaze"] += 1
    counts["akaze_keypoints"] += len(ref.keypoints)


def descriptors(n, nbytes, dup):
    d = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
    if dup and n > 4:
        src = rng.integers(0, n, n // 4)
        d[rng.integers(0, n, n // 4)] = d[src]   # equal rows: ties everywhere
    return d


def case_knn():
    nbytes = int(rng.choice([61, 61, 61, 32, 64, 16]))
    nq, nt = int(rng.integers(1, 3000)), int(rng.integers(1, 60000 if rng.random() < 0.2 else 4000))
    k = int(rng.choice([1, 2, 2, 3, 5, 8, 16, 17, 33, 40]))
    dup = rng.random() < 0.5
    params = dict(nq=nq, nt=nt, k=k, desc_bytes=nbytes, duplicates=dup)
    t = descriptors(nt, nbytes, dup)
    q = descriptors(nq, nbytes, False)
    if nq > 2:
        q[: nq // 3] = t[rng.integers(0, nt, nq // 3)]   # exact hits
    idx, dist = fe.knn_match(q, t, k)
    oi, od = oracle.knn_hamming(q, t, k)
    if not (np.array_equal(idx, oi) and np.array_equal(dist, od)):
        fail("knn", params, "indices / distances differ")
    counts["knn"] += 1


def case_lists():
    nq, nt = int(rng.integers(2, 2500)), int(rng.integers(2, 5000))
    fs = float(rng.choice([0.3, 0.5, 0.7, 0.9, 1.0]))
    params = dict(nq=nq, nt=nt, filter_strength=fs)
    t = pkg.synth.make_descriptor_db(nt, seed=int(rng.integers(1, 1 << 30)))
    q, _ = pkg.synth.make_queries(t, nq, seed=int(rng.integers(1, 1 << 30)), planted=float(rng.uniform(0, 1)), flip=float(rng.uniform(0, 0.2)))
    a, b = fe.get_knn_matches(q, t, 2, fs), oracle.get_knn_matches(q, t, 2, fs)
    if not np.array_equal(a, b):
        fail("lists", params, "get_knn_matches differs")
    a, b = fe.get_bruteforce_matches(q, t), oracle.get_bruteforce_matches(q, t)
    if not np.array_equal(a, b):
        fail("lists", params, "get_bruteforce_matches differs")
    counts["lists"] += 1


def case_homography():
    n = int(rng.choice([4, 5, 8, 30, 200, 3000, 20000]))
    method = int(rng.choice([0, 4, 8, 8, 16]))
    inl = float(rng.uniform(0.15, 1.0))
    noise = float(rng.choice([0.0, 0.3, 1.0]))
    thr = float(rng.choice([1.0, 3.0, 5.0]))
    iters = int(rng.choice([500, 2000, 4096]))
    params = dict(n=n, method=method, inliers=inl, noise=noise, thr=thr, max_iters=iters)
    src, dst, _, _ = pkg.synth.make_ransac_set(n, seed=int(rng.integers(1, 1 << 30)), inlier_frac=inl, noise=noise)
    if rng.random() < 0.1:
        dst = dst.copy()
        dst[:] = dst[0]          # degenerate: every point maps to one point
        params["degenerate"] = True
    H = np.zeros(9)
    mask = np.zeros(len(src), np.uint8)
    rc = L.apds_find_homography_ex(ptr(src), ptr(dst), len(src), method, thr, iters, 0.995, ptr(H), ptr(mask))
    found, Ho, mo = oracle.find_homography(src, dst, method, thr, iters, 0.995)
    if (rc == 0) != bool(found):
        fail("homography", params, f"rc {rc}, oracle found {found}")
    if found:
        if method != 0 and not np.array_equal(mask, mo):
            fail("homography", params, f"masks differ ({int(mask.sum())} vs {int(mo.sum())} inliers)")
        if not np.array_equal(H.reshape(3, 3), Ho.reshape(3, 3)):
            fail("homography", params, f"H differs by {np.abs(H.reshape(3, 3) - Ho.reshape(3, 3)).max():.3g}")
    counts["homography"] += 1


def case_pnp():
    n = int(rng.choice([4, 5, 6, 12, 100, 1500]))
    method = [None, hg.SolvePnPMethod.SOLVEPNP_EPNP, hg.SolvePnPMethod.SOLVEPNP_P3P, hg.SolvePnPMethod.SOLVEPNP_ITERATIVE][int(rng.integers(0, 4))]
    code = {None: 1, hg.SolvePnPMethod.SOLVEPNP_EPNP: 1, hg.SolvePnPMethod.SOLVEPNP_P3P: 2, hg.SolvePnPMethod.SOLVEPNP_ITERATIVE: 0}[method]
    frac = float(rng.uniform(0.3, 1.1))
    noise = float(rng.choice([0.0, 0.3, 1.0]))
    iters, thr, conf = int(rng.choice([50, 200, 1000])), float(rng.choice([1.0, 3.0, 8.0])), float(rng.choice([0.9, 0.99, 0.999]))
    params = dict(n=n, method=str(method), inliers=frac, noise=noise, iters=iters, thr=thr, conf=conf)
    obj, img, K, _, _, _ = pkg.synth.make_pnp_set(n, seed=int(rng.integers(1, 1 << 30)), inlier_frac=frac, noise=noise)
    corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
    sol = hg.pnp_solver_ransac(corr, hg.Cmat(np.ascontiguousarray(K, np.float64), np.float64), iters, thr, conf, None, method)
    rc, r, t, idx = oracle.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=code)
    if (sol is not None) != (rc == 1):
        fail("pnp", params, f"found {sol is not None}, oracle rc {rc}")
    if sol is not None:
        if not np.array_equal(sol.inliers.mat.ravel(), idx):
            fail("pnp", params, "inlier lists differ")
        if not (np.array_equal(sol.rvec.mat.ravel(), r, equal_nan=True) and np.array_equal(sol.tvec.mat.ravel(), t, equal_nan=True)):
            fail("pnp", params, f"pose differs: {sol.rvec.mat.ravel()} {r} / {sol.tvec.mat.ravel()} {t}")
    counts["pnp"] += 1


def case_batch():
    """apds_akaze_extract_batch: every image of a batch == the oracle on that image alone."""
    b = int(rng.integers(2, 9))
    h, w = int(rng.integers(16, 400)), int(rng.integers(16, 400))
    ch = int(rng.choice([1, 3, 4]))
    mp = None if rng.random() < 0.7 else int(rng.integers(1, 300))
    params = dict(batch=b, h=h, w=w, ch=ch, max_points=mp)
    imgs = np.stack([image(h, w, ch, int(rng.integers(0, 5))) for _ in range(b)])
    got = fe.akaze_keypoint_descriptor_extraction_batch(list(imgs), mp)
    for i in range(b):
        ref = oracle.akaze(np.ascontiguousarray(imgs[i]), mp if mp is not None else (1 << 18) - 1)
        if len(got[i].keypoints) != len(ref.keypoints) or not np.array_equal(got[i].descriptors, ref.descriptors):
            fail("batch", dict(params, image=i), f"{len(got[i].keypoints)} keypoints / descriptors, oracle {len(ref.keypoints)}")
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            if not np.array_equal(got[i].keypoints[f], ref.keypoints[f]):
                fail("batch", dict(params, image=i), f"field {f} differs")
    counts["batch"] += 1


def case_ingest():
    """band_merger (mod.rs:346-378) and warp_image_perspective (mod.rs:271-300)."""
    n = int(rng.integers(1, 200000))
    bands = [rng.uniform(-0.3, 1.4, n).astype(np.float32) for _ in range(3)]
    for b in bands:
        b[rng.random(n) < 0.02] = np.nan
    lo = rng.uniform(-0.2, 0.3, 3)
    mm = pkg.geotiff_extractor.BandsMinMax(lo[0], lo[0] + rng.uniform(0.1, 1.2), lo[1], lo[1] + rng.uniform(0.1, 1.2), lo[2], lo[2] + rng.uniform(0.1, 1.2))
    got = pkg.geotiff_extractor.band_merger(bands, mm)
    if not np.array_equal(got, oracle.band_merger(bands[0], bands[1], bands[2], mm.as_array())):
        fail("ingest", dict(n=n, minmax=list(mm.as_array())), "band_merger differs")
    h, w = int(rng.integers(2, 300)), int(rng.integers(2, 300))
    img = hg.Cmat(rng.integers(0, 256, (h, w, 4), dtype=np.uint8), np.uint8, 4)
    ang, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.6, 1.6)
    M = np.array([[sc * np.cos(ang), -sc * np.sin(ang), rng.uniform(-40, 40)], [sc * np.sin(ang), sc * np.cos(ang), rng.uniform(-40, 40)],
                  [rng.uniform(-5e-4, 5e-4), rng.uniform(-5e-4, 5e-4), rng.uniform(0.8, 1.2)]])
    out = hg.warp_image_perspective(img, hg.Cmat(M, np.float64), None).mat
    if not np.array_equal(out, oracle.warp_perspective(img.mat, M)):
        fail("ingest", dict(h=h, w=w, M=M.tolist()), "warp_image_perspective differs")
    counts["ingest"] += 1


t_end = time.time() + budget
families = [case_akaze] if LARGE else [case_akaze, case_akaze, case_knn, case_lists, case_homography, case_pnp, case_batch, case_ingest]
i = 0
last = time.time()
while time.time() < t_end:
    families[i % len(families)]()
    i += 1
    if time.time() - last > 30:
        print(f"[{i} cases] {counts}", flush=True)
        last = time.time()
print(f"fuzz_parity: seed {seed}, {budget:.0f} s, no difference: {counts}", flush=True)
