"""CPU: pin the oracle against every runnable known-answer test the reference holds on this path
 and check its internal consistency. No GPU."""
import numpy as np
import pytest


def test_homography_success_kat(oracle_mod):
    # /root/reference/homographier/src/homographier/mod.rs:437-472: 10x10 grid mapped to itself, RANSAC thr 1.0
    pts = np.array([(i, j) for i in range(1, 11) for j in range(1, 11)], np.float32)
    found, H, mask = oracle_mod.find_homography(pts, pts, 8, 1.0)
    assert found
    assert np.array_equal(np.round(H), np.eye(3))
    assert mask.sum() == 100


def test_raster_to_mat_kat(oracle_mod):
    # mod.rs:556-603: 4x4 RGBA(1, col, row, 1) -> BGRA
    n = 4
    px = np.array([[1, (i % n) + 1, (i // n) + 1, 1] for i in range(n * n)], np.uint8)
    m = oracle_mod.raster_to_mat(px, n, n)
    assert tuple(m[0, 0]) == (1, 1, 1, 1)
    assert tuple(m[3, 3]) == (4, 4, 1, 1)
    assert tuple(m[0, 3]) == (1, 4, 1, 1)
    assert tuple(m[3, 0]) == (4, 1, 1, 1)
    with pytest.raises(ValueError):      # mod.rs:185-187
        oracle_mod.raster_to_mat(px, 5, 4)


def test_homography_methods_agree_on_clean_data(oracle_mod, pkg):
    src, dst, H_true, inl = pkg.synth.make_ransac_set(4000, inlier_frac=1.0, noise=0.0)
    for method in (0, 4, 8):
        found, H, mask = oracle_mod.find_homography(src, dst, method, 3.0)
        assert found
        assert np.allclose(H, H_true, rtol=1e-4, atol=1e-3), method
        # LMEDS derives its own threshold from the median error (~1e-3 px on clean data), so it may drop a few points
        assert mask.all() if method != 4 else mask.mean() > 0.5


def test_ransac_recovers_planted_model(oracle_mod, pkg):
    src, dst, H_true, inl = pkg.synth.make_ransac_set(5000)
    found, H, mask = oracle_mod.find_homography(src, dst, 8, 3.0)
    assert found
    got = mask.astype(bool)
    assert (got & inl).sum() >= 0.99 * inl.sum()
    assert (got & ~inl).sum() <= 0.01 * len(src)
    assert np.allclose(H, H_true, rtol=5e-3, atol=0.5)


def test_too_few_points_is_an_error(oracle_mod):
    pts = np.zeros((3, 2), np.float32)
    with pytest.raises(RuntimeError):
        oracle_mod.find_homography(pts, pts, 8, 3.0)


def test_hamming_knn_definition(oracle_mod):
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (37, 61), dtype=np.uint8)
    t = rng.integers(0, 256, (211, 61), dtype=np.uint8)
    t[17] = q[3]            # exact duplicate pair -> distance 0
    t[90] = q[3]            # tie: the lower index must come first
    idx, dist = oracle_mod.knn_hamming(q, t, 2)
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    order = np.lexsort((np.arange(t.shape[0])[None, :].repeat(len(q), 0), d), axis=1)[:, :2]
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.take_along_axis(d, order, 1))
    assert tuple(idx[3]) == (17, 90) and tuple(dist[3]) == (0, 0)


def test_knn_ratio_and_errors(oracle_mod, pkg):
    db = pkg.synth.make_descriptor_db(3000)
    q, src = pkg.synth.make_queries(db, 400)
    m = oracle_mod.get_knn_matches(q, db, 2, 0.3)
    planted = np.nonzero(src >= 0)[0]
    assert np.array_equal(m["query_idx"], planted)
    assert np.array_equal(m["train_idx"], src[planted])
    assert (m["img_idx"] == 0).all()
    with pytest.raises(RuntimeError):          # lib.rs:108: i.get(1)? with k == 1
        oracle_mod.get_knn_matches(q, db, 1, 0.3)
    with pytest.raises(RuntimeError):          # one train row -> one neighbour
        oracle_mod.get_knn_matches(q, db[:1], 2, 0.3)
    assert len(oracle_mod.get_knn_matches(q[:0], db, 2, 0.3)) == 0
    assert len(oracle_mod.get_knn_matches(q, db[:0], 2, 0.3)) == 0


def test_cross_check_definition(oracle_mod):
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, (50, 61), dtype=np.uint8)
    t = rng.integers(0, 256, (80, 61), dtype=np.uint8)
    m = oracle_mod.get_bruteforce_matches(q, t)
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)     # [q, t]
    nn_q_of_t = d.argmin(0)                                              # first minimum = lower query index
    want = {}
    for ti in range(t.shape[0]):
        qi = nn_q_of_t[ti]
        if qi not in want or d[qi, ti] < want[qi][1]:
            want[qi] = (ti, d[qi, ti])
    assert list(m["query_idx"]) == sorted(want)
    for row in m:
        assert want[row["query_idx"]] == (row["train_idx"], int(row["distance"]))


def test_points_from_matches_modes(oracle_mod):
    K = oracle_mod.KEYPOINT_DTYPE
    k1 = np.zeros(3, K)
    k2 = np.zeros(4, K)
    k1["x"], k1["y"] = [1, 2, 3], [10, 20, 30]
    k2["x"], k2["y"] = [5, 6, 7, 8], [50, 60, 70, 80]
    m = np.zeros(2, oracle_mod.DMATCH_DTYPE)
    m["query_idx"], m["train_idx"] = [2, 1], [0, 3]
    p1, p2 = oracle_mod.get_points_from_matches(k1, k2, m)
    assert p1.tolist() == [[3, 30], [2, 20]] and p2.tolist() == [[5, 50], [8, 80]]
    b1, b2 = oracle_mod.get_points_from_matches(k1, k2, m, bug_compatible=True)   # lib.rs:169,176-177
    assert b1.tolist() == [[1, 10], [1, 10]] and np.array_equal(b1, b2)


def test_akaze_structure(oracle_mod, pkg):
    tile = pkg.synth.make_tile(256, 320, frame_index=3)
    r = oracle_mod.akaze(tile, keep_planes=True)
    # evolution: octaves halve, a new octave is dropped when width < 80 or height < 40
    assert [(lv["w"], lv["h"]) for lv in r.levels[::4]] == [(320, 256), (160, 128), (80, 64)]
    assert [lv["sigma_size"] for lv in r.levels[:4]] == [2, 3, 3, 4]
    assert [lv["nsteps"] for lv in r.levels[:5]] == [0, 3, 3, 4, 4]
    assert len(r.keypoints) > 20
    kp = r.keypoints
    assert r.descriptors.shape == (len(kp), 61)
    assert (r.descriptors[:, 60] & 0xC0 == 0).all()          # bits 486,487 never set
    assert ((kp["angle"] >= 0) & (kp["angle"] < 360)).all()
    assert (kp["response"] > 0.001).all()
    assert np.array_equal(kp["octave"], np.array([r.levels[c]["octave"] for c in kp["class_id"]]))
    # level-major, then row-major output order
    order = np.lexsort((np.arange(len(kp)), kp["class_id"]))
    assert np.array_equal(order, np.arange(len(kp)))
    # grey replicated into BGR(A) and a single-channel image give the same result
    r1 = oracle_mod.akaze(tile[..., 0].copy())
    assert np.array_equal(r1.descriptors, r.descriptors) and np.array_equal(r1.keypoints, kp)
    # max_points keeps the strongest
    r2 = oracle_mod.akaze(tile, max_points=10)
    assert len(r2.keypoints) == 10
    assert np.array_equal(np.sort(r2.keypoints["response"])[::-1], np.sort(kp["response"])[::-1][:10])


def test_akaze_blank_image(oracle_mod):
    r = oracle_mod.akaze(np.full((128, 128, 3), 77, np.uint8))
    assert len(r.keypoints) == 0 and r.descriptors.shape == (0, 61)
    assert abs(r.kcontrast - 0.03) < 1e-9


# ---- "next" rows: geotiff_extractor pixel math and warp_image_perspective, pinned by reference KATs ----
def test_gamma_and_f32_to_u8_kats(oracle_mod):
    # /root/reference/geotiff_extractor/src/image_extractor/mod.rs:517-525, 527-545, 547-555
    assert oracle_mod.gamma_correction(0.5) == np.float32(0.7297401)
    assert oracle_mod.gamma_correction(1.5) is None and oracle_mod.gamma_correction(-0.5) is None
    assert oracle_mod.f32_to_u8(0.2, 0.1, 0.3) == 186
    assert oracle_mod.f32_to_u8(float("nan"), 0.1, 0.3) is None


def test_merging_bands_kat(oracle_mod):
    # mod.rs:626-646: bands (0, 0.5, 1) with min -1, max 2 -> first red is 155
    out = oracle_mod.band_merger([0.0, 0.5, 1.0], [0.0, 0.5, 1.0], [0.0, 0.5, 1.0], [-1, 2, -1, 2, -1, 2])
    assert out.shape == (3, 4) and out[0, 0] == 155 and (out[:, 3] == 255).all()
    nan = float("nan")
    out = oracle_mod.band_merger([nan, nan, 0.5], [nan, 0.2, 5.0], [nan, nan, -3.0], [0, 1, 0, 1, 0, 1])
    assert out[0].tolist() == [0, 0, 0, 0]          # all three NaN -> alpha 0 (mod.rs:353-357)
    assert out[1, 3] == 255 and out[1, 0] == 0      # a NaN band alone -> 0, alpha stays 255
    assert out[2].tolist() == [oracle_mod.f32_to_u8(0.5, 0, 1), 0, 0, 255]   # out-of-range gamma input -> 0


def test_warp_image_empty_kat(oracle_mod):
    # /root/reference/homographier/src/homographier/mod.rs:683-707: the identity warp is idempotent
    n = 4
    img = np.array([[1, (i % n) + 1, (i // n) + 1, 1] for i in range(n * n)], np.uint8).reshape(n, n, 4)[..., [2, 1, 0, 3]].copy()
    assert np.array_equal(oracle_mod.warp_perspective(img, np.eye(3)), img)
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    assert np.array_equal(oracle_mod.warp_perspective(big, np.eye(3)), big)
    shifted = oracle_mod.warp_perspective(big, np.array([[1, 0, 3.0], [0, 1, 2.0], [0, 0, 1]]))
    assert np.array_equal(shifted[2:, 3:], big[:-2, :-3]) and (shifted[:2] == 1).all() and (shifted[:, :3] == 1).all()


# ---- pnp_solver_ransac ----------------------------------------------------------------------------------
def test_pnp_fewer_than_4_points_is_an_error_kat(oracle_mod):
    # /root/reference/homographier/src/homographier/mod.rs:627-638: 2 correspondences, zero camera matrix -> Err
    obj = np.array([[1, 2, 3], [4, 5, 6]], np.float64)
    img = np.array([[1, 2], [4, 5]], np.float64)
    rc, _, _, inl = oracle_mod.solve_pnp_ransac(obj, img, np.zeros((3, 3)), 50, 2.0, 0.99)
    assert rc == -215 and len(inl) == 0


def test_p3p_exact_on_noise_free_points(oracle_mod, pkg):
    # every admissible 4-point sample of exact projections gives the planted pose (to the float rounding of the inputs)
    obj, img, K, rvec, tvec, _ = pkg.synth.make_pnp_set(500, seed=77, inlier_frac=1.1, noise=0.0)
    idx = oracle_mod.pnp_ransac_samples4(len(obj), 200)
    errs = []
    for i in range(len(idx)):
        found, r, t = oracle_mod.pnp_p3p_hypothesis(obj, img, idx[i], K)
        if found:
            errs.append(max(np.abs(r - rvec).max(), np.abs(t - tvec).max() / 100))
    assert len(errs) > 190 and np.median(errs) < 1e-4


def test_pnp_ignored_reference_case_finds_a_pose(oracle_mod):
    # mod.rs:640-682 (#[ignore]d upstream): 5 correspondences, SOLVEPNP_P3P requested, no assertion on values upstream.
    obj = np.array([[0, 5, 1], [5, 0, 0], [5, 5, 1.5], [0, 0, 1], [2, 8, -2]], np.float64)
    img = np.array([[-1.48, 0.39], [2.14, -1.92], [1.74, 0.56], [-2, -1.62], [-0.16, 0.3]], np.float64)
    K = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1]])
    rc, rvec, tvec, inl = oracle_mod.solve_pnp_ransac(obj, img, K, 10000, 100.0, 0.5)
    assert rc == 1 and list(inl) == [0, 1, 2, 3, 4] and np.isfinite(rvec).all() and np.isfinite(tvec).all()
    rc, rvec, tvec, inl = oracle_mod.solve_pnp_ransac(obj, img, K, 10000, 100.0, 0.5, method=2)      # as upstream asks for it
    assert rc in (0, 1) and (rc == 0 or (len(inl) >= 4 and np.isfinite(rvec).all()))


def test_pnp_fixed_elementary_functions(oracle_mod):
    c = np.concatenate([np.linspace(-1, 1, 20001), [1 - 1e-12, -1 + 1e-12, 1 - 1e-9]])
    assert np.abs(oracle_mod.det_acos(c) - np.arccos(c)).max() < 2e-15
    rng = np.random.default_rng(5)
    for _ in range(50):
        rv = rng.normal(size=3)
        rv *= rng.uniform(0.01, 3.1) / np.linalg.norm(rv)
        R = oracle_mod.rodrigues(rv)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
        assert np.allclose(oracle_mod.rodrigues(R), rv, rtol=0, atol=1e-13)
    assert np.array_equal(oracle_mod.rodrigues(np.zeros(3)), np.eye(3))
    assert np.array_equal(oracle_mod.rodrigues(np.eye(3)), np.zeros(3))


def test_epnp_recovers_exact_pose(oracle_mod, pkg):
    for n in (5, 6, 12, 300):
        obj, img, K, rvec, tvec, _ = pkg.synth.make_pnp_set(n, seed=100 + n, inlier_frac=1.1, noise=0.0)
        rc, r, t = oracle_mod.solve_pnp_epnp(obj, img, K)
        assert rc == 1
        assert np.allclose(r, rvec, rtol=0, atol=1e-7), (n, np.abs(r - rvec).max())
        assert np.allclose(t, tvec, rtol=1e-7, atol=1e-5), (n, np.abs(t - tvec).max())
    assert oracle_mod.solve_pnp_epnp(obj[:3], img[:3], K)[0] == -215


def test_pnp_ransac_recovers_planted_pose(oracle_mod, pkg):
    obj, img, K, rvec, tvec, inl = pkg.synth.make_pnp_set(3000, inlier_frac=0.6, noise=0.5)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, 1000, 3.0, 0.99)
    assert rc == 1
    got = np.zeros(len(obj), bool)
    got[idx] = True
    assert (got & inl).sum() > 0.97 * inl.sum()            # nearly every true inlier (0.5 px noise, 3 px gate)
    assert (got & ~inl).sum() <= 0.002 * len(obj) + 2      # chance hits of uniformly random outliers only
    assert np.allclose(r, rvec, atol=2e-3) and np.allclose(t, tvec, rtol=2e-3, atol=0.5)
    # SOLVEPNP_P3P: the RANSAC kernel is Gao's P3P on 4 points, the final pose EPnP over the inliers
    rc3, r3, t3, idx3 = oracle_mod.solve_pnp_ransac(obj, img, K, 1000, 3.0, 0.99, method=2)
    got3 = np.zeros(len(obj), bool)
    got3[idx3] = True
    assert rc3 == 1 and (got3 & inl).sum() > 0.97 * inl.sum() and (got3 & ~inl).sum() <= 0.002 * len(obj) + 2
    assert np.allclose(r3, rvec, atol=2e-3) and np.allclose(t3, tvec, rtol=2e-3, atol=0.5)
    # n == 4 goes through P3P directly (all four are inliers)
    rc4, r4, t4, idx4 = oracle_mod.solve_pnp_ransac(obj[inl][:4], img[inl][:4], K)
    assert rc4 == 1 and list(idx4) == [0, 1, 2, 3] and np.isfinite(r4).all()
    # SOLVEPNP_AP3P: Ke and Roumeliotis' P3P as the RANSAC kernel
    rc5, r5, t5, idx5 = oracle_mod.solve_pnp_ransac(obj, img, K, 1000, 3.0, 0.99, method=5)
    got5 = np.zeros(len(obj), bool)
    got5[idx5] = True
    assert rc5 == 1 and (got5 & inl).sum() > 0.97 * inl.sum() and (got5 & ~inl).sum() <= 0.002 * len(obj) + 2
    assert np.allclose(r5, rvec, atol=2e-3) and np.allclose(t5, tvec, rtol=2e-3, atol=0.5)
    # SOLVEPNP_SQPNP: EPnP's RANSAC (the same inliers as the default method), SQPnP over the inliers at the end
    rc8, r8, t8, idx8 = oracle_mod.solve_pnp_ransac(obj, img, K, 1000, 3.0, 0.99, method=8)
    assert rc8 == 1 and np.array_equal(idx8, idx) and not np.array_equal(r8, r)
    assert np.allclose(r8, rvec, atol=2e-3) and np.allclose(t8, tvec, rtol=2e-3, atol=0.5)
    assert oracle_mod.solve_pnp_ransac(obj, img, K, method=9)[0] == -213   # past the last cv::SolvePnPMethod
    assert oracle_mod.solve_pnp_ransac(obj, img, K, 1000, 3.0, 0.99, method=6)[0] == 0   # IPPE: these object points are not coplanar - no pose
    assert oracle_mod.solve_pnp_ransac(obj, img, K, method=7)[0] == -215   # IPPE_SQUARE: the final solvePnP's CV_Assert(npoints == 4)
    # SOLVEPNP_ITERATIVE: EPnP's RANSAC (hence the same inliers as the default method), then the Levenberg-Marquardt refinement
    rc0, r0, t0, idx0 = oracle_mod.solve_pnp_ransac(obj, img, K, 1000, 3.0, 0.99, method=0)
    assert rc0 == 1 and np.array_equal(idx0, idx)
    assert np.allclose(r0, rvec, atol=2e-3) and np.allclose(t0, tvec, rtol=2e-3, atol=0.5) and not np.array_equal(r0, r)
    # all outliers: no model gathers more than the 4 points that define it ... or only by chance; the call must not fail
    rc2, _, _, idx2 = oracle_mod.solve_pnp_ransac(obj[~inl][:200], img[~inl][:200], K, 200, 1.0, 0.99)
    assert rc2 in (0, 1) and (rc2 == 0) == (len(idx2) == 0)


def test_world_coordinates_closed_form(oracle_mod):
    # feature_database/src/elevationdb.rs:64-104; no runnable reference KAT (its tests need Postgres + a DEM file): checked against
    # the WGS 84 geodetic -> ECEF closed form in numpy, incl. the row-id lookup of elevationdb.rs:240 and both hemispheres
    rng = np.random.default_rng(9)
    el = rng.uniform(-20, 2500, (400, 400))
    for dgt, egt in (([9.0, 1e-4, 0, 57.0, 0, -1e-4], [8.99, 3e-4, 0, 57.01, 0, -3e-4]),
                     ([-70.6, 2e-4, 1e-6, -33.3, -2e-6, -2e-4], [-70.7, 1e-3, 0, -33.2, 0, -1e-3])):
        xy = rng.uniform(0, 900, (500, 2))
        rc, xyz = oracle_mod.world_coordinates(xy, dgt, egt, el)
        assert rc == 0
        lon = dgt[0] + xy[:, 0] * dgt[1] + xy[:, 1] * dgt[2]
        lat = dgt[3] + xy[:, 0] * dgt[4] + xy[:, 1] * dgt[5]
        inv = np.linalg.inv(np.array([[egt[1], egt[2]], [egt[4], egt[5]]]))
        p = (np.stack([lon - egt[0], lat - egt[3]], 1) @ inv.T)
        ix, iy = np.floor(np.abs(p[:, 0]) + 0.5) * np.sign(p[:, 0]), np.floor(np.abs(p[:, 1]) + 0.5) * np.sign(p[:, 1])
        h = el.ravel()[(iy * 400 + ix).astype(int)]
        a, f = 6378137.0, 1 / 298.257223563
        es = f * (2 - f)
        ph, la = np.radians(lat), np.radians(lon)
        N = a / np.sqrt(1 - es * np.sin(ph) ** 2)
        ref = np.stack([(N + h) * np.cos(ph) * np.cos(la), (N + h) * np.cos(ph) * np.sin(la), (N * (1 - es) + h) * np.sin(ph)], 1)
        assert np.abs(xyz - ref).max() < 5e-8                      # metres, at |xyz| ~ 6.4e6
    rc, xyz = oracle_mod.world_coordinates([[0, 0]], [9.0, 1e-4, 0, 57.0, 0, -1e-4])          # no elevation data -> height 0
    assert rc == 0 and abs(np.linalg.norm(xyz[0]) - 6363.2e3) < 1e3
    rc, xyz = oracle_mod.world_coordinates([[1e7, 1e7]], [9.0, 1e-4, 0, 57.0, 0, -1e-4], [8.99, 3e-4, 0, 57.01, 0, -3e-4], el)
    assert rc == -211 and np.isnan(xyz).all()
    assert oracle_mod.world_coordinates([[0, 0]], [9.0, 1e-4, 0, 57.0, 0, -1e-4], [0, 0, 0, 0, 0, 0], el)[0] == -5


def test_rho_oracle_on_the_reference_kat_and_on_planted_data(pkg, oracle_mod):
    # HomographyMethod::RHO in the oracle: the reference's homography_success data (mod.rs:437-472) gives the identity, and a planted
    # homography with 40 % inliers is recovered; deterministic (fixed xorshift128+ seed, as rho.cpp)
    pts = np.array([(i, j) for i in range(1, 11) for j in range(1, 11)], np.float32)
    ok, H, mask = oracle_mod.find_homography(pts, pts, 16, 1.0)
    assert ok and mask.all() and np.allclose(H.reshape(3, 3), np.eye(3), atol=1e-5)
    src, dst, H_true, flag = pkg.synth.make_ransac_set(3000, seed=99, inlier_frac=0.4, noise=0.5)
    ok, H, mask = oracle_mod.find_homography(src, dst, 16, 3.0)
    ok2, H2, mask2 = oracle_mod.find_homography(src, dst, 16, 3.0)
    assert ok and ok2 and np.array_equal(H, H2) and np.array_equal(mask, mask2)
    assert (mask.astype(bool) & flag).sum() >= 0.9 * flag.sum() and np.allclose(H.reshape(3, 3), H_true, rtol=2e-2, atol=1.0)
    # exactly four pairs: cv::findHomography takes the plain 4-point solve whatever the method (`method == 0 || npoints == 4` comes first)
    s4, d4, _, _ = pkg.synth.make_ransac_set(4, seed=77, inlier_frac=0.2, noise=1.0)
    ok0, H0, _ = oracle_mod.find_homography(s4, d4, 0, 5.0)
    for method in (4, 8, 16):
        ok, H, mask = oracle_mod.find_homography(s4, d4, method, 5.0)
        assert ok == ok0 and np.array_equal(H, H0) and mask.all()


def test_warp_generic_types_kat(oracle_mod):
    # mod.rs:683-707 for every element type the generic function admits: the identity warp is idempotent; the generic restatement equals the
    # dedicated 8UC4 one on 8UC4; a whole-pixel shift moves pixels and fills the uncovered part with the border value 1
    rng = np.random.default_rng(4)
    for dt in (np.uint8, np.float32):
        for ch in (1, 3, 4):
            shape = (23, 31) if ch == 1 else (23, 31, ch)
            img = rng.integers(0, 256, shape).astype(dt)
            assert np.array_equal(oracle_mod.warp_perspective(img, np.eye(3)), img), (dt, ch)
            sh = oracle_mod.warp_perspective(img, np.array([[1, 0, 3.0], [0, 1, 2.0], [0, 0, 1]]))
            assert np.array_equal(sh[2:, 3:], img[:-2, :-3]) and np.all(sh[:2] == 1) and np.all(sh[:, :3] == 1), (dt, ch)
    img4 = rng.integers(0, 256, (40, 52, 4)).astype(np.uint8)
    M = np.array([[0.9, -0.2, 6.0], [0.25, 1.1, -3.0], [1e-3, -2e-3, 1.0]])
    assert np.array_equal(oracle_mod.warp_perspective(img4, M), oracle_mod.warp_perspective_generic(img4, M))
    # f32: a half-pixel shift is the mean of two neighbours (weights 0.5 / 0.5 exactly)
    f = rng.normal(0, 10, (9, 12)).astype(np.float32)
    half = oracle_mod.warp_perspective(f, np.array([[1, 0, 0.5], [0, 1, 0.0], [0, 0, 1]]))
    assert np.array_equal(half[:, 1:], (f[:, :-1] * np.float32(0.5) + f[:, 1:] * np.float32(0.5)).astype(np.float32))


def test_oracle_hamming_against_a_numpy_popcount(oracle_mod):
    """The oracle's brute-force matcher against an implementation that shares no text with it: popcount(xor) via numpy.unpackbits and a
    stable argsort (distance, then lower train index - BFMatcher's order). tests/test_match_gpu.py holds the GPU matchers to the same."""
    rng = np.random.default_rng(22)
    db = rng.integers(0, 256, (2500, 61), dtype=np.uint8)
    db[:, 60] &= 0x3F
    q = rng.integers(0, 256, (400, 61), dtype=np.uint8)
    q[:, 60] &= 0x3F
    q[:150] = db[rng.integers(0, 2500, 150)]
    db[1200:1300] = db[50:150]
    q[150:200] = db[50:100]
    d = np.unpackbits(q[:, None, :] ^ db[None, :, :], axis=2).sum(2).astype(np.int32)
    order = np.argsort(d, axis=1, kind="stable")[:, :3]
    oi, od = oracle_mod.knn_hamming(q, db, 3)
    assert np.array_equal(oi, order.astype(np.int32)) and np.array_equal(od, np.take_along_axis(d, order, axis=1))
