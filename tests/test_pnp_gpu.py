"""GPU: pnp_solver_ransac parity, HIP path through the C ABI vs the oracle.
Bar: every 5-point hypothesis (rvec, tvec) bit-identical (both sides evaluate the same IEEE double operations in the same order); inlier
index lists identical; final pose bit-identical (the all-inlier solve runs on the host in index order). Tolerance against the planted pose
is stated per test. PARITY UNPINNED against OpenCV itself (oracle/pnp_oracle.cpp header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hyp(pkg, obj, img, K, idx, model_points=None):
    idx = np.ascontiguousarray(idx, np.int32)
    out = np.zeros((len(idx), 6))
    K = np.ascontiguousarray(K, np.float64)
    rc = pkg.lib().apds_pnp_hypotheses(pkg._lib.ptr(obj), pkg._lib.ptr(img), len(obj), pkg._lib.ptr(K), pkg._lib.ptr(idx), len(idx),
                                       model_points or idx.shape[1], pkg._lib.ptr(out))
    assert rc == 0, pkg.lib().apds_last_error()
    return out


def _solve(pkg, obj, img, K, iters, thr, conf, method=None):
    hg = pkg.homographier
    corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
    return hg.pnp_solver_ransac(corr, hg.Cmat(np.ascontiguousarray(K, np.float64), np.float64), iters, thr, conf, None, method)


def test_reference_kat_fewer_than_4_points(gpu_pkg):
    # /root/reference/homographier/src/homographier/mod.rs:627-638
    hg = gpu_pkg.homographier
    corr = [hg.ImgObjCorrespondence((1, 2, 3), (1, 2)), hg.ImgObjCorrespondence((4, 5, 6), (4, 5))]
    with pytest.raises(hg.MatError) as e:
        hg.pnp_solver_ransac(corr, hg.Cmat.zeros(3, 3, np.float64), 50, 2.0, 0.99, None, None)
    assert e.value.kind == "Opencv" and e.value.inner.code == -215


def test_hypotheses_bit_identical_to_oracle(gpu_pkg, oracle_mod):
    obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(2000, inlier_frac=0.6, noise=0.5)
    idx = oracle_mod.pnp_ransac_samples(len(obj), 300)
    got = _hyp(gpu_pkg, obj, img, K, idx)
    for b in range(len(idx)):
        r, t = oracle_mod.pnp_hypothesis(obj, img, idx[b], K)
        want = np.concatenate([r, t])
        assert np.array_equal(got[b], want, equal_nan=True), (b, got[b], want)


@pytest.mark.parametrize("n,frac,noise,iters,thr,conf", [(2000, 0.6, 0.5, 1000, 3.0, 0.99), (50000, 0.4, 0.5, 2000, 2.0, 0.995),
                                                        (300, 0.9, 0.2, 100, 8.0, 0.99), (6, 1.1, 0.0, 50, 2.0, 0.99),
                                                        (5, 1.1, 0.0, 50, 2.0, 0.99)])
def test_ransac_matches_oracle(gpu_pkg, oracle_mod, n, frac, noise, iters, thr, conf):
    obj, img, K, rvec, tvec, inl = gpu_pkg.synth.make_pnp_set(n, seed=7 + n, inlier_frac=frac, noise=noise)
    sol = _solve(gpu_pkg, obj, img, K, iters, thr, conf)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf)
    assert (sol is not None) == (rc == 1)
    assert np.array_equal(sol.inliers.mat.ravel(), idx)
    assert np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
    # and the answer is the planted pose (pixel-noise limited)
    assert np.allclose(r, rvec, atol=3e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)


def test_p3p_hypotheses_bit_identical_to_oracle(gpu_pkg, oracle_mod):
    obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(2000, inlier_frac=0.6, noise=0.5)
    idx = oracle_mod.pnp_ransac_samples4(len(obj), 400)
    got = _hyp(gpu_pkg, obj, img, K, idx)
    n_found = 0
    for b in range(len(idx)):
        found, r, t = oracle_mod.pnp_p3p_hypothesis(obj, img, idx[b], K)
        if found:
            n_found += 1
            assert np.array_equal(got[b], np.concatenate([r, t]), equal_nan=True), (b, got[b], r, t)
        else:
            assert np.isnan(got[b]).all()
    assert n_found > 300


@pytest.mark.parametrize("n,frac,noise,iters,thr,conf", [(2000, 0.6, 0.5, 1000, 3.0, 0.99), (50000, 0.4, 0.5, 2000, 2.0, 0.995),
                                                        (300, 0.9, 0.2, 100, 8.0, 0.99), (5, 1.1, 0.0, 50, 2.0, 0.99)])
def test_p3p_ransac_matches_oracle(gpu_pkg, oracle_mod, n, frac, noise, iters, thr, conf):
    hg = gpu_pkg.homographier
    obj, img, K, rvec, tvec, inl = gpu_pkg.synth.make_pnp_set(n, seed=17 + n, inlier_frac=frac, noise=noise)
    sol = _solve(gpu_pkg, obj, img, K, iters, thr, conf, hg.SolvePnPMethod.SOLVEPNP_P3P)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=2)
    assert (sol is not None) == (rc == 1) and rc == 1
    assert np.array_equal(sol.inliers.mat.ravel(), idx)
    assert np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
    assert np.allclose(r, rvec, atol=3e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)


def test_ap3p_hypotheses_bit_identical_to_oracle(gpu_pkg, oracle_mod):
    """SOLVEPNP_AP3P's kernel (ap3p.cpp restated: csrc/pnp_core.h ap3p_best_pose / oracle/pnp_oracle.cpp ap3p_solve): one thread per 4-point
    sample, the pose of every sample equal to the oracle's bit for bit (apds_pnp_hypotheses with model_points = 40)."""
    obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(2000, inlier_frac=0.6, noise=0.5)
    idx = oracle_mod.pnp_ransac_samples4(len(obj), 400)
    got = _hyp(gpu_pkg, obj, img, K, idx, model_points=40)
    n_found = 0
    for b in range(len(idx)):
        found, r, t = oracle_mod.pnp_ap3p_hypothesis(obj, img, idx[b], K)
        if found:
            n_found += 1
            assert np.array_equal(got[b], np.concatenate([r, t]), equal_nan=True), (b, got[b], r, t)
        else:
            assert np.isnan(got[b]).all()
    assert n_found > 300


@pytest.mark.parametrize("n,frac,noise,iters,thr,conf", [(2000, 0.6, 0.5, 1000, 3.0, 0.99), (50000, 0.4, 0.5, 2000, 2.0, 0.995),
                                                        (300, 0.9, 0.2, 100, 8.0, 0.99), (5, 1.1, 0.0, 50, 2.0, 0.99), (4, 1.1, 0.0, 50, 2.0, 0.99)])
def test_ap3p_ransac_matches_oracle(gpu_pkg, oracle_mod, n, frac, noise, iters, thr, conf):
    hg = gpu_pkg.homographier
    obj, img, K, rvec, tvec, inl = gpu_pkg.synth.make_pnp_set(n, seed=27 + n, inlier_frac=frac, noise=noise)
    sol = _solve(gpu_pkg, obj, img, K, iters, thr, conf, hg.SolvePnPMethod.SOLVEPNP_AP3P)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=5)
    assert (sol is not None) == (rc == 1) and rc == 1
    assert np.array_equal(sol.inliers.mat.ravel(), idx)
    assert np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
    assert np.allclose(r, rvec, atol=3e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)


def test_four_points_and_unbuilt_methods(gpu_pkg, oracle_mod):
    hg = gpu_pkg.homographier
    obj, img, K, _, _, inl = gpu_pkg.synth.make_pnp_set(400, inlier_frac=0.5, noise=0.5)
    with pytest.raises(hg.MatError) as e:   # a value past cv::SolvePnPMethod's last member
        _solve(gpu_pkg, obj, img, K, 100, 3.0, 0.99, 9)
    assert e.value.inner.code == -213
    # SOLVEPNP_IPPE on object points that are not coplanar: OpenCV's plane solver refuses, solvePnPRansac returns false -> Ok(None)
    assert _solve(gpu_pkg, obj, img, K, 100, 3.0, 0.99, hg.SolvePnPMethod.SOLVEPNP_IPPE) is None
    assert oracle_mod.solve_pnp_ransac(obj, img, K, 100, 3.0, 0.99, method=6)[0] == 0
    # SOLVEPNP_IPPE_SQUARE: the final solvePnP over the >= 5 inliers asserts npoints == 4; solvePnPRansac rethrows -> Err(MatError::Opencv)
    with pytest.raises(hg.MatError) as e:
        _solve(gpu_pkg, obj, img, K, 100, 3.0, 0.99, hg.SolvePnPMethod.SOLVEPNP_IPPE_SQUARE)
    assert e.value.kind == "Opencv" and e.value.inner.code == -215
    assert oracle_mod.solve_pnp_ransac(obj, img, K, 100, 3.0, 0.99, method=7)[0] == -215
    # exactly four correspondences: one direct P3P solve, whatever the method (OpenCV switches kernels)
    o4, i4 = obj[inl][:4], img[inl][:4]
    for method in (None, hg.SolvePnPMethod.SOLVEPNP_P3P, hg.SolvePnPMethod.SOLVEPNP_IPPE_SQUARE, hg.SolvePnPMethod.SOLVEPNP_SQPNP):
        sol = _solve(gpu_pkg, o4, i4, K, 100, 3.0, 0.99, method)
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(o4, i4, K, 100, 3.0, 0.99, method=int(method) if method else 1)
        assert (sol is not None) == (rc == 1)
        if sol is not None:
            assert list(sol.inliers.mat.ravel()) == [0, 1, 2, 3] and np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
    # outliers only: same outcome (usually Ok(None)) as the oracle
    o, i = obj[~inl][:150], img[~inl][:150]
    sol = _solve(gpu_pkg, o, i, K, 150, 0.5, 0.99)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(o, i, K, 150, 0.5, 0.99)
    assert (sol is not None) == (rc == 1)
    if sol is not None:
        assert np.array_equal(sol.inliers.mat.ravel(), idx)


@pytest.mark.parametrize("n,frac,noise,iters,thr", [(600, 0.7, 0.5, 300, 3.0), (5000, 0.4, 0.8, 1000, 4.0), (40, 1.1, 0.3, 100, 3.0), (6, 1.1, 0.0, 50, 2.0)])
def test_iterative_method_equals_oracle(gpu_pkg, oracle_mod, n, frac, noise, iters, thr):
    # SolvePnPMethod::SOLVEPNP_ITERATIVE (mod.rs:327,359-360): the RANSAC stage is EPnP's, the final pose is the Levenberg-Marquardt
    # refinement from the best RANSAC model over the inliers - inliers and pose bit-identical to the oracle's restatement
    hg = gpu_pkg.homographier
    obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(n, seed=0x1750 + n, inlier_frac=frac, noise=noise)
    sol = _solve(gpu_pkg, obj, img, K, iters, thr, 0.99, hg.SolvePnPMethod.SOLVEPNP_ITERATIVE)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, 0.99, method=0)
    assert rc == 1 and sol is not None
    assert np.array_equal(sol.inliers.mat.ravel(), idx)
    assert np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
    # and it differs from EPnP's final pose (the refinement did something) unless the data are noise-free
    sol_e = _solve(gpu_pkg, obj, img, K, iters, thr, 0.99, None)
    if noise > 0:
        assert not np.array_equal(sol_e.rvec.mat.ravel(), r)


@pytest.mark.parametrize("n,frac,noise,iters,thr", [(600, 0.7, 0.5, 300, 3.0), (5000, 0.4, 0.8, 1000, 4.0), (40, 1.1, 0.3, 100, 3.0), (6, 1.1, 0.0, 50, 2.0),
                                                   (5, 1.1, 0.0, 50, 2.0)])
def test_sqpnp_method_equals_oracle(gpu_pkg, oracle_mod, n, frac, noise, iters, thr):
    # SolvePnPMethod::SOLVEPNP_SQPNP (mod.rs:327,359-360): the RANSAC stage is EPnP's, the final pose is SQPnP over the inliers
    # (csrc/sqpnp_core.h against the oracle's separate restatement) - inliers and pose bit-identical
    hg = gpu_pkg.homographier
    obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(n, seed=0x5190 + n, inlier_frac=frac, noise=noise)
    sol = _solve(gpu_pkg, obj, img, K, iters, thr, 0.99, hg.SolvePnPMethod.SOLVEPNP_SQPNP)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, 0.99, method=8)
    assert rc == 1 and sol is not None
    assert np.array_equal(sol.inliers.mat.ravel(), idx)
    assert np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
    sol_e = _solve(gpu_pkg, obj, img, K, iters, thr, 0.99, None)
    assert np.array_equal(sol_e.inliers.mat.ravel(), idx)          # same RANSAC kernel, same consensus set
    if noise > 0 and n > 5:
        assert not np.array_equal(sol_e.rvec.mat.ravel(), r)       # another final solver ran
    # planar terrain (every null vector of SQPnP's 9 x 9 matrix is rank one there)
    flat = obj.copy()
    flat[:, 2] = 12.5
    sol = _solve(gpu_pkg, flat, img, K, iters, thr, 0.99, hg.SolvePnPMethod.SOLVEPNP_SQPNP)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(flat, img, K, iters, thr, 0.99, method=8)
    assert (sol is not None) == (rc == 1)
    if sol is not None:
        assert np.array_equal(sol.inliers.mat.ravel(), idx) and np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)


def test_dls_and_upnp_run_epnp(gpu_pkg, oracle_mod):
    # OpenCV 4's solvePnPGeneric sends SOLVEPNP_DLS and SOLVEPNP_UPNP to EPnP ("broken implementation"): identical answers
    hg = gpu_pkg.homographier
    obj, img, K, _, _, _ = gpu_pkg.synth.make_pnp_set(1500, seed=77, inlier_frac=0.6, noise=0.5)
    want = _solve(gpu_pkg, obj, img, K, 400, 3.0, 0.99, None)
    for method in (hg.SolvePnPMethod.SOLVEPNP_DLS, hg.SolvePnPMethod.SOLVEPNP_UPNP):
        got = _solve(gpu_pkg, obj, img, K, 400, 3.0, 0.99, method)
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, 400, 3.0, 0.99, method=int(method))
        assert rc == 1 and np.array_equal(got.inliers.mat.ravel(), idx) and np.array_equal(got.rvec.mat.ravel(), r) and np.array_equal(got.tvec.mat.ravel(), t)
        assert np.array_equal(got.rvec.mat.ravel(), want.rvec.mat.ravel()) and np.array_equal(got.tvec.mat.ravel(), want.tvec.mat.ravel())


def _planar_set(synth, n, seed, frac, noise, tilt):
    """make_pnp_set's correspondences with the object points pressed onto a plane (optionally tilted and off the origin) and the inliers'
    pixels re-projected from the planted pose."""
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(n, seed=seed, inlier_frac=frac, noise=noise)
    rng = np.random.default_rng(seed)
    obj = obj.copy()
    obj[:, 2] = 0.0
    if tilt:
        a = rng.normal(size=3)
        th = np.linalg.norm(a)
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        obj = obj @ (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)).T + np.array([3.0, -2.0, 5.0])
    th = np.linalg.norm(rvec)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    cam = obj[flag] @ R.T + tvec
    img = img.copy()
    img[flag] = cam[:, :2] / cam[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, noise, (int(flag.sum()), 2))
    return obj, img, K, rvec, tvec, flag


@pytest.mark.parametrize("n,frac,noise,iters,thr,tilt", [(600, 0.7, 0.5, 300, 3.0, False), (5000, 0.4, 0.8, 1000, 4.0, True), (40, 1.1, 0.3, 100, 3.0, True),
                                                        (6, 1.1, 0.0, 50, 2.0, False)])
def test_ippe_method_equals_oracle(gpu_pkg, oracle_mod, n, frac, noise, iters, thr, tilt):
    # SolvePnPMethod::SOLVEPNP_IPPE (mod.rs:327,359-360) on a planar target: EPnP's RANSAC, IPPE over the inliers (csrc/ippe_core.h against
    # the oracle's separate restatement) - inliers and pose bit-identical
    hg = gpu_pkg.homographier
    obj, img, K, rvec, tvec, _ = _planar_set(gpu_pkg.synth, n, 0x1BBE + n, frac, noise, tilt)
    sol = _solve(gpu_pkg, obj, img, K, iters, thr, 0.99, hg.SolvePnPMethod.SOLVEPNP_IPPE)
    rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, 0.99, method=6)
    assert (sol is not None) == (rc == 1)
    if sol is not None:
        assert np.array_equal(sol.inliers.mat.ravel(), idx)
        assert np.array_equal(sol.rvec.mat.ravel(), r) and np.array_equal(sol.tvec.mat.ravel(), t)
