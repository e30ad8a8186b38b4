"""Anchors that do not come from this repo's oracle: closed-form cases with a known answer, optimality conditions and an independent
numpy solver, applied to BOTH the oracle (CPU tests) and the product (GPU tests).

Why: oracle/pnp_oracle.cpp and csrc/pnp_core.h (and the homography refit on either side) restate the same OpenCV routines and share
much of their text, so "GPU == oracle bit for bit" alone would also pass on a shared misreading. Nothing below
looks at the other implementation: a pose or a homography is checked against (a) the construction that generated the data,
(b) first-order optimality of the reprojection error over the reported inliers, evaluated with numpy formulas written here, and
(c) an independent linear solver (DLT via numpy SVD + Gauss-Newton to convergence). Tolerances are stated where they are used.
"""
import numpy as np
import pytest


# ---------------------------------------------------------------------------------------------------------------- helpers (numpy only)
def rodrigues(rvec):
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def project(obj, R, t, K):
    cam = obj @ R.T + t
    return cam[:, :2] / cam[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])


def dlt_pnp(obj, img, K):
    """Independent pose solver: projection matrix by the direct linear transform (SVD of the 2n x 12 system on normalised image
    coordinates), then the nearest rotation (SVD) and the scale that goes with it."""
    xn = (img - np.array([K[0, 2], K[1, 2]])) / np.array([K[0, 0], K[1, 1]])
    n = len(obj)
    A = np.zeros((2 * n, 12))
    X = np.hstack([obj, np.ones((n, 1))])
    A[0::2, 0:4] = X
    A[0::2, 8:12] = -xn[:, 0:1] * X
    A[1::2, 4:8] = X
    A[1::2, 8:12] = -xn[:, 1:2] * X
    P = np.linalg.svd(A)[2][-1].reshape(3, 4)
    if np.linalg.det(P[:, :3]) < 0:
        P = -P
    U, S, Vt = np.linalg.svd(P[:, :3])
    R = U @ Vt
    t = P[:, 3] / S.mean()
    return R, t


def h_apply(H, p):
    q = np.hstack([p, np.ones((len(p), 1))]) @ H.T
    return q[:, :2] / q[:, 2:3]


def h_residual_jacobian(h8, src, dst):
    """residuals (2n) and Jacobian (2n x 8) of the reprojection error for H = [h8, 1]"""
    x, y = src[:, 0], src[:, 1]
    w = h8[6] * x + h8[7] * y + 1.0
    u = (h8[0] * x + h8[1] * y + h8[2]) / w
    v = (h8[3] * x + h8[4] * y + h8[5]) / w
    r = np.empty(2 * len(src))
    r[0::2], r[1::2] = u - dst[:, 0], v - dst[:, 1]
    J = np.zeros((2 * len(src), 8))
    J[0::2, 0], J[0::2, 1], J[0::2, 2] = x / w, y / w, 1 / w
    J[0::2, 6], J[0::2, 7] = -x * u / w, -y * u / w
    J[1::2, 3], J[1::2, 4], J[1::2, 5] = x / w, y / w, 1 / w
    J[1::2, 6], J[1::2, 7] = -x * v / w, -y * v / w
    return r, J


def gauss_newton_h(src, dst, iters=50):
    """Independent homography estimate: normalised DLT (numpy SVD) then Gauss-Newton on the reprojection error until it stops moving."""
    def norm(p):
        c = p.mean(0)
        s = np.sqrt(2) / np.mean(np.linalg.norm(p - c, axis=1))
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
    T1, T2 = norm(src), norm(dst)
    a, b = h_apply(T1, src), h_apply(T2, dst)
    A = np.zeros((2 * len(a), 9))
    A[0::2, 0:2], A[0::2, 2] = a, 1
    A[0::2, 6:8], A[0::2, 8] = -b[:, 0:1] * a, -b[:, 0]
    A[1::2, 3:5], A[1::2, 5] = a, 1
    A[1::2, 6:8], A[1::2, 8] = -b[:, 1:2] * a, -b[:, 1]
    Hn = np.linalg.svd(A)[2][-1].reshape(3, 3)
    H = np.linalg.inv(T2) @ Hn @ T1
    h = (H / H[2, 2]).ravel()[:8]
    for _ in range(iters):
        r, J = h_residual_jacobian(h, src, dst)
        step = np.linalg.lstsq(J, r, rcond=None)[0]
        h = h - step
        if np.abs(step).max() < 1e-13:
            break
    return np.append(h, 1.0).reshape(3, 3)


# ---------------------------------------------------------------------------------------------------------------- the checks
def check_homography_solver(find, synth, methods=(0, 4, 8, 16)):
    """find(src, dst, method, thr) -> (found, H 3x3, mask or None)"""
    # (a) noise-free data generated from a known H: every method must return it (f64 paths to 1e-8, the binary32 RHO path to 1e-4)
    src, dst, H_true, _ = synth.make_ransac_set(400, seed=11, inlier_frac=1.0, noise=0.0)
    for m in methods:
        ok, H, _ = find(src, dst, m, 3.0)
        assert ok, m
        tol = 1e-4 if m == 16 else 1e-8
        # src / dst are f32 roundings of the exact points: compare transfer, which is what those roundings allow (1e-3 px), and H itself loosely
        assert np.abs(h_apply(H / H[2, 2], src.astype(np.float64)) - dst).max() < 2e-3, m
        assert np.allclose(H / H[2, 2], H_true / H_true[2, 2], rtol=1e-4, atol=max(tol, 2e-3)), m
    # (b) + (c) noisy inliers + outliers: first-order optimality over the reported inliers and agreement with the independent solver
    src, dst, H_true, flag = synth.make_ransac_set(3000, seed=12, inlier_frac=0.7, noise=0.4)   # (LMEDS needs a clear majority of inliers)
    for m in (8, 4):
        ok, H, mask = find(src, dst, m, 3.0)
        assert ok and mask is not None
        sel = mask.astype(bool)
        # (LMEDS derives its own, wider threshold from the median residual: a few percent of the uniform outliers fall inside it)
        assert (sel & flag).sum() >= 0.97 * flag.sum() and (sel & ~flag).sum() <= (0.02 if m == 8 else 0.10) * sel.sum()
        s64, d64 = src[sel].astype(np.float64), dst[sel].astype(np.float64)
        h8 = (H / H[2, 2]).ravel()[:8]
        r, J = h_residual_jacobian(h8, s64, d64)
        g = J.T @ r
        # the refined H is a stationary point of the inliers' squared reprojection error: |J^T r| is tiny against |J| |r| (LM stops at 1e-7 steps)
        assert np.abs(g).max() <= 1e-5 * np.linalg.norm(J, axis=0).max() * np.linalg.norm(r), (m, np.abs(g).max())
        # against the independent optimum (DLT + Gauss-Newton to convergence): OpenCV stops its LM after 10 iterations, i.e. next to the
        # minimum rather than at it, so: the objective within 1e-6 relative of the optimum's, the entries of H within 1e-4 relative
        Hi = gauss_newton_h(s64, d64)
        ri, _ = h_residual_jacobian(Hi.ravel()[:8], s64, d64)
        assert r @ r <= (ri @ ri) * (1 + 1e-6), (m, r @ r, ri @ ri)
        assert np.allclose(H / H[2, 2], Hi, rtol=1e-4, atol=1e-3), (m, np.abs(H / H[2, 2] - Hi).max())
        assert np.allclose(H / H[2, 2], H_true / H_true[2, 2], rtol=5e-3, atol=0.3 if m == 8 else 1.0)


def check_pnp_solver(solve, synth):
    """solve(obj, img, K, iters, thr, conf) -> (found, rvec, tvec, inlier indices)"""
    # (a) noise-free projections of a known pose: recovered to solver precision
    obj, img, K, rvec, tvec, _ = synth.make_pnp_set(500, seed=21, inlier_frac=1.1, noise=0.0)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99)
    assert ok and len(idx) == 500
    assert np.allclose(rodrigues(r), rodrigues(rvec), atol=1e-7) and np.allclose(t, tvec, rtol=1e-7, atol=1e-5)
    assert np.abs(project(obj, rodrigues(r), t, K) - img).max() < 1e-4      # (EPnP is a linearisation + 5 Gauss-Newton steps: 1e-5 px here)
    # (b) noise + outliers: inlier set, reprojection error at the noise level, agreement with the independent DLT pose
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(4000, seed=22, inlier_frac=0.6, noise=0.5)
    ok, r, t, idx = solve(obj, img, K, 1000, 3.0, 0.99)
    assert ok
    sel = np.zeros(len(obj), bool)
    sel[idx] = True
    assert (sel & flag).sum() >= 0.97 * flag.sum() and (sel & ~flag).sum() <= 0.01 * sel.sum()
    R = rodrigues(r)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    err = np.linalg.norm(project(obj[sel], R, t, K) - img[sel], axis=1)
    rms = np.sqrt(np.mean(err ** 2))
    assert 0.5 < rms < 1.0                        # pixel noise N(0, 0.5^2) per axis: rms ~ 0.71
    Rd, td = dlt_pnp(obj[sel], img[sel], K)
    rms_d = np.sqrt(np.mean(np.linalg.norm(project(obj[sel], Rd, td, K) - img[sel], axis=1) ** 2))
    assert rms <= 1.05 * rms_d + 1e-3             # EPnP over the inliers fits them at least as well as the plain DLT does
    ang = np.arccos(np.clip((np.trace(R.T @ Rd) - 1) / 2, -1, 1))
    assert ang < 2e-3 and np.linalg.norm(t - td) < 5e-3 * np.linalg.norm(td)
    assert np.allclose(R, rodrigues(rvec), atol=2e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)


def pose_residual_jacobian(rvec, tvec, obj, img, K, eps=1e-6):
    """reprojection residuals (2n) of a pose and their Jacobian wrt (rvec, tvec) by central differences - nothing analytic shared with
    either implementation"""
    def res(p):
        return (project(obj, rodrigues(p[:3]), p[3:], K) - img).ravel()
    p0 = np.concatenate([rvec, tvec]).astype(np.float64)
    J = np.empty((2 * len(obj), 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = eps * max(1.0, abs(p0[k]))
        J[:, k] = (res(p0 + d) - res(p0 - d)) / (2 * d[k])
    return res(p0), J


def check_iterative_pnp_solver(solve, synth):
    """SOLVEPNP_ITERATIVE: the final pose is a Levenberg-Marquardt refinement over the inliers, so it must sit at a minimum of the
    reprojection error: (a) the gradient vanishes, (b) an independent Gauss-Newton iteration (numerical Jacobian, run to convergence from
    the reported pose) cannot lower the objective by more than 1e-6 relative and stays at the same pose, (c) it fits the inliers at
    least as well as EPnP's pose does."""
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(3000, seed=31, inlier_frac=0.7, noise=0.4)
    ok, r, t, idx = solve(obj, img, K, 500, 3.0, 0.99, 0)          # 0 = SOLVEPNP_ITERATIVE
    ok_e, re, te, idx_e = solve(obj, img, K, 500, 3.0, 0.99, 1)    # 1 = SOLVEPNP_EPNP: same RANSAC kernel, hence the same inliers
    assert ok and ok_e and np.array_equal(idx, idx_e)
    o, i = obj[idx].astype(np.float32).astype(np.float64), img[idx].astype(np.float32).astype(np.float64)   # the solver sees float copies
    res, J = pose_residual_jacobian(r, t, o, i, K)
    f = res @ res
    g = J.T @ res
    scale = np.sqrt(np.diag(J.T @ J)) * np.sqrt(f)
    assert np.all(np.abs(g) <= 1e-5 * scale), (np.abs(g) / scale)                       # (a) first-order optimality
    p = np.concatenate([r, t]).astype(np.float64)
    for _ in range(30):                                                               # (b) independent Gauss-Newton
        rr, JJ = pose_residual_jacobian(p[:3], p[3:], o, i, K)
        step = np.linalg.lstsq(JJ, -rr, rcond=None)[0]
        p = p + step
        if np.linalg.norm(step) < 1e-13 * max(1.0, np.linalg.norm(p)):
            break
    rr, _ = pose_residual_jacobian(p[:3], p[3:], o, i, K)
    assert f <= (rr @ rr) * (1 + 1e-6)
    assert np.allclose(np.concatenate([r, t]), p, rtol=1e-5, atol=1e-5)
    res_e, _ = pose_residual_jacobian(re, te, o, i, K)
    assert f <= res_e @ res_e * (1 + 1e-9)                                              # (c) no worse than EPnP over the same inliers
    assert np.allclose(rodrigues(r), rodrigues(rvec), atol=2e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)
    # noise-free data: the refinement must not move a perfect pose away
    obj, img, K, rvec, tvec, _ = synth.make_pnp_set(400, seed=33, inlier_frac=1.1, noise=0.0)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99, 0)
    assert ok and len(idx) == 400 and np.abs(project(obj, rodrigues(r), t, K) - img).max() < 1e-4
    # PLANAR object points (terrain without relief): the reference passes use_extrinsic_guess = false (mod.rs:354), so the final solvePnP
    # starts from a homography plane -> image instead of the DLT. Whatever the start, the result must sit at the minimum of the
    # reprojection error over its inliers and at the pose the data was made with.
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(1500, seed=35, inlier_frac=0.75, noise=0.3)
    obj = obj.copy()
    obj[:, 2] = 0.0
    rng = np.random.default_rng(35)
    img = img.copy()
    img[flag] = project(obj[flag], rodrigues(rvec), tvec, K) + rng.normal(0, 0.3, (int(flag.sum()), 2))
    ok, r, t, idx = solve(obj, img, K, 500, 3.0, 0.99, 0)
    assert ok and len(idx) >= 0.95 * flag.sum()
    o, i = obj[idx].astype(np.float32).astype(np.float64), img[idx].astype(np.float32).astype(np.float64)
    res, J = pose_residual_jacobian(r, t, o, i, K)
    g = J.T @ res
    scale = np.sqrt(np.diag(J.T @ J)) * np.sqrt(res @ res)
    assert np.all(np.abs(g) <= 1e-5 * scale), (np.abs(g) / scale)
    assert np.allclose(rodrigues(r), rodrigues(rvec), atol=3e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.5)
    # six noise-free points in general position: RANSAC's five-point kernel, every point an inlier, the DLT start from exactly six
    # points (the fewest it accepts), and a pose that reprojects them exactly
    obj, img, K, rvec, tvec, _ = synth.make_pnp_set(6, seed=37, inlier_frac=1.1, noise=0.0)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99, 0)
    assert ok and len(idx) == 6 and np.abs(project(obj, rodrigues(r), t, K) - img).max() < 1e-3


def check_ap3p_solver(solve, synth):
    """SOLVEPNP_AP3P (4-point algebraic kernel in the RANSAC loop, EPnP over the inliers at the end): what the construction of the data says -
    noise-free correspondences give back the pose they were projected with, every point an inlier; with pixel noise and 40 % outliers the
    consensus set is the planted one and the pose is the planted pose to the noise level."""
    obj, img, K, rvec, tvec, _ = synth.make_pnp_set(400, seed=43, inlier_frac=1.1, noise=0.0)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99, 5)          # 5 = SOLVEPNP_AP3P
    assert ok and len(idx) == 400 and np.abs(project(obj, rodrigues(r), t, K) - img).max() < 2e-2
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(3000, seed=45, inlier_frac=0.6, noise=0.4)
    ok, r, t, idx = solve(obj, img, K, 500, 3.0, 0.99, 5)
    assert ok
    planted = np.flatnonzero(flag)
    assert len(np.intersect1d(idx, planted)) >= 0.97 * len(planted) and len(np.setdiff1d(idx, planted)) <= 0.01 * len(planted)
    assert np.allclose(rodrigues(r), rodrigues(rvec), atol=2e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)
    # exactly four correspondences with the AP3P flag: one direct solve that reprojects all four
    o4, i4 = obj[planted[:4]], project(obj[planted[:4]], rodrigues(rvec), tvec, K)
    ok, r, t, idx = solve(o4, i4, K, 50, 2.0, 0.99, 5)
    assert ok and len(idx) == 4 and np.abs(project(o4, rodrigues(r), t, K) - i4).max() < 2e-2


# ---------------------------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_ap3p_against_external_anchors(pkg, oracle_mod):
    def solve(obj, img, K, iters, thr, conf, method):
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=method)
        return rc == 1, r, t, idx
    check_ap3p_solver(solve, pkg.synth)


def test_oracle_iterative_pnp_against_external_anchors(pkg, oracle_mod):
    def solve(obj, img, K, iters, thr, conf, method):
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=method)
        return rc == 1, r, t, idx
    check_iterative_pnp_solver(solve, pkg.synth)


def test_oracle_homography_against_external_anchors(pkg, oracle_mod):
    def find(src, dst, method, thr):
        ok, H, mask = oracle_mod.find_homography(src, dst, method, thr)
        return ok, H.reshape(3, 3), (mask if method in (4, 8) else None)
    check_homography_solver(find, pkg.synth)


def test_oracle_pnp_against_external_anchors(pkg, oracle_mod):
    def solve(obj, img, K, iters, thr, conf):
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf)
        return rc == 1, r, t, idx
    check_pnp_solver(solve, pkg.synth)


# ---------------------------------------------------------------------------------------------------------------- product (GPU)
@pytest.mark.gpu
def test_gpu_homography_against_external_anchors(gpu_pkg):
    hg = gpu_pkg.homographier

    def find(src, dst, method, thr):
        try:
            H, mask = hg.find_homography_mat(src, dst, hg.HomographyMethod(method), thr)
        except hg.MatError:
            return False, None, None
        return True, H.mat, (mask.mat.ravel() if mask is not None else None)
    check_homography_solver(find, gpu_pkg.synth)


@pytest.mark.gpu
def test_gpu_pnp_against_external_anchors(gpu_pkg):
    hg = gpu_pkg.homographier

    def solve(obj, img, K, iters, thr, conf):
        corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
        sol = hg.pnp_solver_ransac(corr, hg.Cmat(np.ascontiguousarray(K, np.float64), np.float64), iters, thr, conf, None, None)
        if sol is None:
            return False, None, None, None
        return True, sol.rvec.mat.ravel(), sol.tvec.mat.ravel(), sol.inliers.mat.ravel()
    check_pnp_solver(solve, gpu_pkg.synth)


@pytest.mark.gpu
def test_gpu_ap3p_against_external_anchors(gpu_pkg):
    hg = gpu_pkg.homographier

    def solve(obj, img, K, iters, thr, conf, method):
        corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
        sol = hg.pnp_solver_ransac(corr, hg.Cmat(np.ascontiguousarray(K, np.float64), np.float64), iters, thr, conf, None, hg.SolvePnPMethod(method))
        if sol is None:
            return False, None, None, None
        return True, sol.rvec.mat.ravel(), sol.tvec.mat.ravel(), sol.inliers.mat.ravel()
    check_ap3p_solver(solve, gpu_pkg.synth)


@pytest.mark.gpu
def test_gpu_iterative_pnp_against_external_anchors(gpu_pkg):
    hg = gpu_pkg.homographier

    def solve(obj, img, K, iters, thr, conf, method):
        corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
        sol = hg.pnp_solver_ransac(corr, hg.Cmat(np.ascontiguousarray(K, np.float64), np.float64), iters, thr, conf, None, hg.SolvePnPMethod(method))
        if sol is None:
            return False, None, None, None
        return True, sol.rvec.mat.ravel(), sol.tvec.mat.ravel(), sol.inliers.mat.ravel()
    check_iterative_pnp_solver(solve, gpu_pkg.synth)


# ---------------------------------------------------------------------------------------------------------------- SQPnP (round 4)
def sqpnp_cost(R, t, obj, img, K):
    """SQPnP's own objective, written from the paper's definition and not from either implementation: the squared distance between each
    camera-frame point and its image ray scaled to the point's depth, sum_i |(Xc - x Zc, Yc - y Zc)|^2 with (x, y) the normalised pixel."""
    xn = (img - np.array([K[0, 2], K[1, 2]])) / np.array([K[0, 0], K[1, 1]])
    cam = obj @ R.T + t
    return float(np.sum((cam[:, 0] - xn[:, 0] * cam[:, 2]) ** 2 + (cam[:, 1] - xn[:, 1] * cam[:, 2]) ** 2))


def small_rotation(w):
    return rodrigues(np.asarray(w, np.float64))


def check_sqpnp_pose(r, t, obj, img, K, others=()):
    """A pose SQPnP reports must be (a) a proper rotation, (b) a stationary point of sqpnp_cost over SO(3) x R^3 (central differences along
    the six generators, written here), and (c) the GLOBAL minimiser: no worse than any other pose offered (EPnP's, the planted one) or 300
    random rotations about it with their own best translation."""
    R = rodrigues(r)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
    f = sqpnp_cost(R, t, obj, img, K)
    spread = float(np.sum((obj - obj.mean(0)) ** 2)) + 1e-12
    h = 1e-6
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        fp = sqpnp_cost(small_rotation(d[:3]) @ R, t + d[3:], obj, img, K)
        fm = sqpnp_cost(small_rotation(-d[:3]) @ R, t - d[3:], obj, img, K)
        assert abs(fp - fm) / (2 * h) <= 1e-5 * spread, (k, (fp - fm) / (2 * h))                 # (b)
    for Ro, to in others:
        assert f <= sqpnp_cost(Ro, to, obj, img, K) * (1 + 1e-9) + 1e-12                       # (c) against the poses of other solvers
    xn = (img - np.array([K[0, 2], K[1, 2]])) / np.array([K[0, 0], K[1, 1]])
    rng = np.random.default_rng(5)
    for _ in range(300):                                                                      # (c) against rotations around it
        Rp = small_rotation(rng.normal(0, rng.choice([1e-3, 0.05, 1.0]), 3)) @ R
        # the translation that is optimal for Rp: linear least squares on the two residual rows per point
        A = np.zeros((2 * len(obj), 3))
        A[0::2, 0] = 1
        A[0::2, 2] = -xn[:, 0]
        A[1::2, 1] = 1
        A[1::2, 2] = -xn[:, 1]
        rot = obj @ Rp.T
        b = np.empty(2 * len(obj))
        b[0::2] = -(rot[:, 0] - xn[:, 0] * rot[:, 2])
        b[1::2] = -(rot[:, 1] - xn[:, 1] * rot[:, 2])
        tp = np.linalg.lstsq(A, b, rcond=None)[0]
        if np.mean((rot + tp)[:, 2] > 0) < 0.5:
            continue                                                                          # behind the camera: not a candidate
        assert f <= sqpnp_cost(Rp, tp, obj, img, K) * (1 + 1e-9) + 1e-12


def check_sqpnp_direct(direct, synth):
    """solvePnP(SOLVEPNP_SQPNP) alone: the construction of the data comes back on noise-free points (general position, a plane, a
    nearly flat cloud, from three points up to thousands), and with pixel noise the pose is the global minimum of SQPnP's objective."""
    rng = np.random.default_rng(91)
    K = np.array([[900.0, 0, 640], [0, 880, 360], [0, 0, 1]])
    for case, n in enumerate([4, 5, 6, 12, 100, 5000, 40, 40]):
        rvec = rng.normal(0, [0.2, 0.8, 1.5, 2.5][case % 4], 3)
        tvec = np.array([0.3, -0.2, 7.0]) + rng.normal(0, 0.4, 3)
        obj = rng.uniform(-1, 1, (n, 3))
        if case == 6:
            obj[:, 2] = 0.25                      # a plane that does not pass through the origin
        if case == 7:
            obj[:, 2] *= 1e-3                     # nearly flat
        img = project(obj, rodrigues(rvec), tvec, K)
        ok, r, t = direct(obj, img, K)
        assert ok and np.allclose(rodrigues(r), rodrigues(rvec), atol=1e-7) and np.allclose(t, tvec, atol=1e-6), (case, n)
        noisy = img + rng.normal(0, 0.7, img.shape)
        ok, r, t = direct(obj, noisy, K)
        assert ok
        check_sqpnp_pose(r, t, obj, noisy, K, others=[(rodrigues(rvec), tvec)])
        if n >= 12:
            assert np.allclose(rodrigues(r), rodrigues(rvec), atol=0.05) and np.allclose(t, tvec, atol=0.3)
    # points behind the camera only (the data of a camera looking away): no pose
    obj = rng.uniform(-1, 1, (30, 3))
    img = project(obj, np.eye(3), np.array([0, 0, 5.0]), K)
    ok, r, t = direct(obj, img, K)
    assert ok and t[2] > 0      # SQPnP keeps the solution in front of the camera (the mirrored one is rejected)


def check_sqpnp_ransac(solve, synth):
    """pnp_solver_ransac with SOLVEPNP_SQPNP: EPnP stays the RANSAC kernel (same inliers as the default method), the final pose is
    SQPnP's global minimum over those inliers; DLS and UPNP are EPnP."""
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(3000, seed=61, inlier_frac=0.65, noise=0.4)
    ok, r, t, idx = solve(obj, img, K, 500, 3.0, 0.99, 8)            # 8 = SOLVEPNP_SQPNP
    ok_e, re, te, idx_e = solve(obj, img, K, 500, 3.0, 0.99, 1)      # 1 = SOLVEPNP_EPNP
    assert ok and ok_e and np.array_equal(idx, idx_e)
    o, i = obj[idx].astype(np.float32).astype(np.float64), img[idx].astype(np.float32).astype(np.float64)   # the solver sees float copies
    check_sqpnp_pose(r, t, o, i, K, others=[(rodrigues(re), te), (rodrigues(rvec), tvec)])
    assert np.allclose(rodrigues(r), rodrigues(rvec), atol=2e-3) and np.allclose(t, tvec, rtol=3e-3, atol=1.0)
    assert not np.array_equal(r, re)                                  # a different final solver ran
    for alias in (3, 4):                                              # SOLVEPNP_DLS, SOLVEPNP_UPNP
        ok_a, ra, ta, idx_a = solve(obj, img, K, 500, 3.0, 0.99, alias)
        assert ok_a and np.array_equal(ra, re) and np.array_equal(ta, te) and np.array_equal(idx_a, idx_e)
    # noise-free, planar terrain, six points
    obj, img, K, rvec, tvec, _ = synth.make_pnp_set(400, seed=63, inlier_frac=1.1, noise=0.0)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99, 8)
    assert ok and len(idx) == 400 and np.abs(project(obj, rodrigues(r), t, K) - img).max() < 1e-3
    obj = obj.copy()
    obj[:, 2] = 0.0
    img = project(obj, rodrigues(rvec), tvec, K)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99, 8)
    assert ok and len(idx) == 400 and np.abs(project(obj, rodrigues(r), t, K) - img).max() < 1e-3
    obj, img, K, rvec, tvec, _ = synth.make_pnp_set(6, seed=65, inlier_frac=1.1, noise=0.0)
    ok, r, t, idx = solve(obj, img, K, 100, 2.0, 0.99, 8)
    # (solvePnPRansac hands the solver float copies of the points: pixel coordinates near 2800 carry 2.4e-4 of rounding each)
    assert ok and len(idx) == 6 and np.abs(project(obj, rodrigues(r), t, K) - img).max() < 5e-3


def test_oracle_sqpnp_against_external_anchors(pkg, oracle_mod):
    def direct(obj, img, K):
        rc, r, t = oracle_mod.solve_pnp_sqpnp(obj, img, K)
        return rc == 1, r, t

    def solve(obj, img, K, iters, thr, conf, method):
        rc, r, t, idx = oracle_mod.solve_pnp_ransac(obj, img, K, iters, thr, conf, method=method)
        return rc == 1, r, t, idx
    check_sqpnp_direct(direct, pkg.synth)
    check_sqpnp_ransac(solve, pkg.synth)


def product_sqpnp(pkg, obj, img, K):
    """apds_pnp_sqpnp through the C ABI: the product's own text (csrc/sqpnp_core.h), host arithmetic, no device needed."""
    import ctypes as C
    obj, img, K = (np.ascontiguousarray(a, np.float64) for a in (obj, img, K))
    r, t, found = np.zeros(3), np.zeros(3), C.c_int(0)
    rc = pkg.lib().apds_pnp_sqpnp(pkg._lib.ptr(obj), pkg._lib.ptr(img), len(obj), pkg._lib.ptr(K), pkg._lib.ptr(r), pkg._lib.ptr(t), C.byref(found))
    assert rc == 0, pkg.lib().apds_last_error()
    return found.value == 1, r, t


def test_product_sqpnp_against_external_anchors(pkg):
    check_sqpnp_direct(lambda obj, img, K: product_sqpnp(pkg, obj, img, K), pkg.synth)


@pytest.mark.gpu
def test_gpu_sqpnp_against_external_anchors(gpu_pkg):
    hg = gpu_pkg.homographier

    def solve(obj, img, K, iters, thr, conf, method):
        corr = [hg.ImgObjCorrespondence(o, i) for o, i in zip(obj, img)]
        sol = hg.pnp_solver_ransac(corr, hg.Cmat(np.ascontiguousarray(K, np.float64), np.float64), iters, thr, conf, None, hg.SolvePnPMethod(method))
        if sol is None:
            return False, None, None, None
        return True, sol.rvec.mat.ravel(), sol.tvec.mat.ravel(), sol.inliers.mat.ravel()
    check_sqpnp_ransac(solve, gpu_pkg.synth)


# ---------------------------------------------------------------------------------------------------------------- IPPE (round 4)
def check_ippe_direct(direct):
    """solvePnP(SOLVEPNP_IPPE) alone. IPPE is a closed-form method (a homography, its Jacobian at the centroid, two candidate rotations), not
    a minimiser, so the anchors are the construction of the data: noise-free points of ANY plane (through the origin, off it, tilted, with the
    first three points collinear) give back the pose they were projected with; with pixel noise the pose is a proper rotation within the
    noise's reach of the planted one and reprojects to the noise level; a cloud with relief has no IPPE pose."""
    rng = np.random.default_rng(17)
    K = np.array([[900.0, 0, 640], [0, 880, 360], [0, 0, 1]])
    for case, n in enumerate([4, 5, 12, 100, 3000, 60, 60]):
        rvec = rng.normal(0, [0.2, 0.6, 1.0][case % 3], 3)
        tvec = np.array([0.3, -0.2, 7.0]) + rng.normal(0, 0.4, 3)
        obj = rng.uniform(-1, 1, (n, 3))
        obj[:, 2] = 0.0
        if case in (1, 3):
            obj[:, 2] = -0.4
        if case in (2, 4, 6):
            obj = obj @ rodrigues(rng.normal(0, 1.0, 3)).T + rng.normal(0, 1.0, 3)
        if case == 5:
            obj[1] = 0.5 * (obj[0] + obj[2])
            obj = obj @ rodrigues(np.array([0.4, -0.3, 0.2])).T + 0.1
        img = project(obj, rodrigues(rvec), tvec, K)
        ok, r, t = direct(obj, img, K)
        assert ok and np.allclose(rodrigues(r), rodrigues(rvec), atol=1e-8) and np.allclose(t, tvec, atol=1e-7), (case, n)
        if n >= 12:
            noisy = img + rng.normal(0, 0.5, img.shape)
            ok, r, t = direct(obj, noisy, K)
            R = rodrigues(r)
            assert ok and np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
            assert np.allclose(R, rodrigues(rvec), atol=0.05) and np.allclose(t, tvec, atol=0.3), (case, n)
            assert np.sqrt(np.mean((project(obj, R, t, K) - noisy) ** 2)) < 1.0      # 0.5 px of noise per coordinate
    obj = rng.uniform(-1, 1, (50, 3))                                                # relief of +-1 against IPPE's 1e-3
    ok, r, t = direct(obj, project(obj, np.eye(3), np.array([0, 0, 6.0]), K), K)
    assert not ok


def test_oracle_ippe_against_external_anchors(oracle_mod):
    def direct(obj, img, K):
        rc, r, t = oracle_mod.solve_pnp_ippe(obj, img, K)
        return rc == 1, r, t
    check_ippe_direct(direct)


def test_product_ippe_against_external_anchors(pkg):
    import ctypes as C

    def direct(obj, img, K):
        obj, img, K = (np.ascontiguousarray(a, np.float64) for a in (obj, img, K))
        r, t, found = np.zeros(3), np.zeros(3), C.c_int(0)
        rc = pkg.lib().apds_pnp_ippe(pkg._lib.ptr(obj), pkg._lib.ptr(img), len(obj), pkg._lib.ptr(K), pkg._lib.ptr(r), pkg._lib.ptr(t), C.byref(found))
        assert rc == 0, pkg.lib().apds_last_error()
        return found.value == 1, r, t
    check_ippe_direct(direct)
