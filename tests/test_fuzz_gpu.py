"""GPU: seeded random sweeps over shapes and parameters the hand-written cases do not enumerate — every case bit-exact against the
oracle (match lists, AKAZE output, RANSAC masks), homography within the stated tolerance. Sizes are small: the oracle is the slow side."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_match_random_shapes(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(2024)
    fe = gpu_pkg.feature_extraction
    oracle_mod.set_threads(8)
    for case in range(250):
        nb = int(rng.choice([8, 32, 61, 61, 61, 64]))
        nq = int(rng.integers(1, 700))
        nt = int(rng.choice([rng.integers(1, 50), rng.integers(50, 3000), rng.integers(3000, 70000)]))
        k = int(rng.choice([1, 2, 2, 2, 3, 7, 16]))
        few = rng.random() < 0.3                       # few distinct rows: heavy ties
        pool = rng.integers(0, 256, (5 if few else nt, nb), dtype=np.uint8)
        t = pool[rng.integers(0, len(pool), nt)] if few else pool
        q = t[rng.integers(0, nt, nq)].copy()
        flips = rng.random((nq, nb)) < 0.05
        q ^= (flips * rng.integers(0, 256, (nq, nb))).astype(np.uint8)
        if nb == 61:
            t = t.copy(); t[:, 60] &= 0x3F; q[:, 60] &= 0x3F
        idx, dist = fe.knn_match(q, t, k)
        oi, od = oracle_mod.knn_hamming(q, t, k)
        assert np.array_equal(dist, od) and np.array_equal(idx, oi), (case, nb, nq, nt, k, few)
        if nt >= 2:
            fs = float(rng.choice([0.3, 0.7, 0.9, 1.0]))
            assert np.array_equal(fe.get_knn_matches(q, t, 2, fs), oracle_mod.get_knn_matches(q, t, 2, fs)), (case, "ratio")
        assert np.array_equal(fe.get_bruteforce_matches(q, t), oracle_mod.get_bruteforce_matches(q, t)), (case, "crosscheck")


def test_akaze_random_shapes(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(77)
    oracle_mod.set_threads(8)
    for case in range(48):
        h, w = int(rng.integers(81, 520)), int(rng.integers(81, 700))
        ch = int(rng.choice([1, 3, 4]))
        tile = gpu_pkg.synth.make_tile(h, w, frame_index=100 + case, channels=ch)
        if case % 4 == 1:
            tile = np.ascontiguousarray(tile[::-1])                     # different content statistics
        if case % 5 == 2:
            tile = (tile // 3 + 90).astype(np.uint8)                    # low contrast
        max_points = None if case % 3 else int(rng.integers(1, 400))
        got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, max_points)
        ref = oracle_mod.akaze(tile, max_points=max_points) if max_points else oracle_mod.akaze(tile)
        assert len(got.keypoints) == len(ref.keypoints), (case, h, w, ch, max_points)
        assert np.array_equal(got.keypoints, ref.keypoints) and np.array_equal(got.descriptors, ref.descriptors), (case, h, w, ch, max_points)


def test_homography_random_sets(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(5)
    L, ptr = gpu_pkg.lib(), gpu_pkg._lib.ptr
    for case in range(160):
        n = int(rng.choice([4, 5, 8, 30, 200, 3000]))
        frac = float(rng.choice([0.2, 0.5, 0.9, 1.1]))
        noise = float(rng.choice([0.0, 0.3, 1.5]))
        method = int(rng.choice([0, 4, 8, 8, 8]))
        thr = float(rng.choice([0.5, 1.0, 3.0, 7.5]))
        iters = int(rng.choice([10, 200, 2000]))
        conf = float(rng.choice([0.9, 0.995]))
        src, dst, _, _ = gpu_pkg.synth.make_ransac_set(n, seed=1000 + case, inlier_frac=frac, noise=noise, extent=float(rng.choice([64, 1024, 4096])))
        H, mask = np.zeros(9), np.zeros(n, np.uint8)
        rc = L.apds_find_homography_ex(ptr(src), ptr(dst), n, method, thr, iters, conf, ptr(H), ptr(mask))
        found, Ho, mo = oracle_mod.find_homography(src, dst, method, thr, iters, conf)
        assert (rc == 0) == found, (case, rc, found)
        if found:
            if method:
                assert np.array_equal(mask, mo), (case, n, method, int(mask.sum()), int(mo.sum()))
            assert np.allclose(H.reshape(3, 3), Ho, rtol=1e-6, atol=1e-8), (case, n, method)
            if (int(mo.sum()) if method else n) <= 256:      # small refits run on the host in index order: bit-identical
                assert np.array_equal(H.reshape(3, 3), Ho), (case, n, method, np.abs(H.reshape(3, 3) - Ho).max())


def test_pnp_random_sets(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(31)
    hg = gpu_pkg.homographier
    for case in range(60):
        n = int(rng.choice([4, 5, 6, 9, 40, 400, 5000]))
        frac = float(rng.choice([0.3, 0.6, 0.95, 1.1]))
        noise = float(rng.choice([0.0, 0.4, 2.0]))
        method = int(rng.choice([1, 1, 2]))
        iters = int(rng.choice([5, 100, 1000]))
        thr = float(rng.choice([1.0, 3.0, 8.0]))
        conf = float(rng.choice([0.9, 0.99]))
        obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(n, seed=500 + case, inlier_frac=frac, noise=noise)
        corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
        sol = hg.pnp_solver_ransac(corr, hg.Cmat(K, np.float64), iters, thr, conf, None, hg.SolvePnPMethod(method))
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=method)
        assert (sol is not None) == (rc == 1), (case, n, method, rc)
        if sol is not None:
            assert np.array_equal(sol.inliers.mat.ravel(), idx), (case, n, method)
            assert np.array_equal(sol.rvec.mat.ravel(), r, equal_nan=True) and np.array_equal(sol.tvec.mat.ravel(), t, equal_nan=True), (case, n, method)


def test_ingest_and_warp_random(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(99)
    ge, hg = gpu_pkg.geotiff_extractor, gpu_pkg.homographier
    for case in range(40):
        n = int(rng.integers(1, 50000))
        bands = [(rng.normal(rng.uniform(-5, 5), rng.uniform(0.01, 50), n)).astype(np.float32) for _ in range(3)]
        for b in bands:
            b[rng.random(n) < 0.02] = np.nan
            b[rng.random(n) < 0.005] = np.inf
            b[rng.random(n) < 0.005] = -np.inf
        lo = [float(np.nanmin(np.where(np.isfinite(b), b, np.nan))) if np.isfinite(b).any() else 0.0 for b in bands]
        hi = [float(np.nanmax(np.where(np.isfinite(b), b, np.nan))) if np.isfinite(b).any() else 1.0 for b in bands]
        if case % 7 == 3:
            hi[1] = lo[1]                                            # degenerate range: division by zero -> None -> 0
        mm = ge.BandsMinMax(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
        assert np.array_equal(ge.band_merger(bands, mm), oracle_mod.band_merger(*bands, mm.as_array())), case
    for case in range(40):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        img = hg.Cmat(rng.integers(0, 256, (h, w, 4), dtype=np.uint8), np.uint8, 4)
        M = np.eye(3) + rng.normal(0, 0.15, (3, 3)) * np.array([[1, 1, 40], [1, 1, 40], [1e-3, 1e-3, 0.2]])
        if case % 9 == 4:
            M[2] = [0.01, -0.02, 0.0]                                # rows where W crosses zero
        size = None if case % 3 else (int(rng.integers(1, 350)), int(rng.integers(1, 350)))
        got = hg.warp_image_perspective(img, hg.Cmat(M, np.float64), size).mat
        want = oracle_mod.warp_perspective(img.mat, M, size)
        assert np.array_equal(got, want), (case, h, w, size)
