"""ctypes front end of oracle/liboracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (cubesat-apds_amd/) never imports this module. See oracle/oracle.h for what is
restated, which reference lines it follows and the parity status (AKAZE/match: "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("img_idx", "<i4"), ("distance", "<f4")])

PLANE_LT, PLANE_LSMOOTH, PLANE_LX, PLANE_LY, PLANE_LDET, PLANE_LFLOW, PLANE_MASK0, PLANE_MASK1 = range(8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("akaze_oracle.cpp", "match_oracle.cpp", "homography_oracle.cpp", "rho_oracle.cpp", "ingest_oracle.cpp", "pnp_oracle.cpp",
                                             "oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_NATIVE = False


def use_native():
    """Switch to liboracle_native.so (-O3 -march=native), built here and now on this host (bench.py's cpu_baseline
    leg). Returns the flags string reported with the baseline; on any build/load failure keeps the portable library."""
    global _LIB, _NATIVE
    so = os.path.join(_HERE, "liboracle_native.so")
    try:
        if os.path.exists(so):
            os.remove(so)             # never trust a copy built on another host
        subprocess.check_call(["make", "-C", _HERE, "liboracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        C.CDLL(so)
    except Exception:
        return "-O2 (portable build; the -O3 -march=native build failed on this host)"
    _LIB, _NATIVE = None, True
    return "-O3 -march=native -fopenmp -ffp-contract=off"


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("APDS_ORACLE_LIB") or os.path.join(_HERE, "liboracle_native.so" if _NATIVE else "liboracle.so")   # (override: the sanitizer build, tests)
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        u8p, f32p, i32p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double))
        L.oracle_akaze_run.restype = C.c_void_p
        L.oracle_akaze_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int]
        for name, rt in (("oracle_akaze_num_keypoints", C.c_int), ("oracle_akaze_keypoints", C.c_void_p),
                         ("oracle_akaze_descriptors", C.c_void_p), ("oracle_akaze_desc_bytes", C.c_int),
                         ("oracle_akaze_num_levels", C.c_int), ("oracle_akaze_kcontrast", C.c_float),
                         ("oracle_akaze_gray", C.c_void_p)):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.oracle_akaze_level_info.restype = None
        L.oracle_akaze_level_info.argtypes = [C.c_void_p, C.c_int, i32p, f32p]
        L.oracle_akaze_level_tau.restype = C.c_int
        L.oracle_akaze_level_tau.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int]
        L.oracle_akaze_plane.restype = C.c_void_p
        L.oracle_akaze_plane.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_akaze_free.restype = None
        L.oracle_akaze_free.argtypes = [C.c_void_p]
        L.oracle_knn_hamming.restype = None
        L.oracle_knn_hamming.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]
        L.oracle_get_knn_matches.restype = C.c_int
        L.oracle_get_knn_matches.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_int,
                                             C.c_int, C.c_float, C.c_void_p]
        L.oracle_get_bruteforce_matches.restype = C.c_int
        L.oracle_get_bruteforce_matches.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t,
                                                    C.c_int, C.c_void_p]
        L.oracle_get_points_from_matches.restype = C.c_int
        L.oracle_get_points_from_matches.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p]
        L.oracle_find_homography.restype = C.c_int
        L.oracle_find_homography.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                             C.c_void_p, C.c_void_p]
        L.oracle_solve_pnp_ransac.restype = C.c_int
        L.oracle_solve_pnp_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_solve_pnp_epnp.restype = C.c_int
        L.oracle_solve_pnp_epnp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_solve_pnp_ippe.restype = C.c_int
        L.oracle_solve_pnp_ippe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_solve_pnp_sqpnp.restype = C.c_int
        L.oracle_solve_pnp_sqpnp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_pnp_ransac_samples.restype = C.c_int
        L.oracle_pnp_ransac_samples.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.oracle_pnp_hypothesis.restype = C.c_int
        L.oracle_pnp_hypothesis.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_pnp_ransac_samples4.restype = C.c_int
        L.oracle_pnp_ransac_samples4.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.oracle_pnp_ap3p_hypothesis.restype = C.c_int
        L.oracle_pnp_ap3p_hypothesis.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_pnp_p3p_hypothesis.restype = C.c_int
        L.oracle_pnp_p3p_hypothesis.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_rodrigues.restype = None
        L.oracle_rodrigues.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_det_acos.restype = None
        L.oracle_det_acos.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_world_coordinates.restype = C.c_int
        L.oracle_world_coordinates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_homography_4pt.restype = C.c_int
        L.oracle_homography_4pt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_ransac_samples.restype = C.c_int
        L.oracle_ransac_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_raster_to_mat.restype = C.c_int
        L.oracle_raster_to_mat.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.oracle_knn_l2.restype = None
        L.oracle_knn_l2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_band_merger.restype = None
        L.oracle_band_merger.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_gamma_correction.restype = C.c_float
        L.oracle_gamma_correction.argtypes = [C.c_float, C.POINTER(C.c_int)]
        L.oracle_f32_to_u8.restype = C.c_int
        L.oracle_f32_to_u8.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int)]
        L.oracle_warp_perspective_any.restype = C.c_int
        L.oracle_warp_perspective_any.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_warp_perspective_8uc4.restype = C.c_int
        L.oracle_warp_perspective_8uc4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_get_threads.restype = C.c_int
        _LIB = L
    return _LIB


def set_threads(n):
    lib().oracle_set_threads(int(n))


def host_threads():
    """(threads to use, nproc): nproc = the CPUs this process may run on (what `nproc` prints); the thread count is the one of
    8, 16, 32, 64, ..., nproc that runs a small detect+describe fastest, because a container's CPU share can be far below its
    affinity mask (256 visible CPUs with a 16-CPU quota made 256 OpenMP threads 25x slower than 32)."""
    import time
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({min(nproc, c) for c in (8, 16, 32, 64, 128, nproc)})
    rng = np.random.default_rng(7)
    img = (rng.random((512, 512)) * 255).astype(np.uint8)
    best, best_t = cands[0], None
    for c in cands:
        set_threads(c)
        akaze(img)
        t0 = time.perf_counter()
        akaze(img)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t * 0.9:      # more threads only if they pay clearly
            best, best_t = c, dt
    set_threads(best)
    return best, nproc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class AkazeResult:
    """Keypoints, descriptors and (optionally) every intermediate plane of one oracle AKAZE run."""

    def __init__(self, handle, keep):
        L = lib()
        self._h = handle
        n = L.oracle_akaze_num_keypoints(handle)
        self.desc_bytes = L.oracle_akaze_desc_bytes(handle)
        kp = np.zeros(n, KEYPOINT_DTYPE)
        desc = np.zeros((n, self.desc_bytes), np.uint8)
        if n:
            C.memmove(_ptr(kp), L.oracle_akaze_keypoints(handle), kp.nbytes)
            C.memmove(_ptr(desc), L.oracle_akaze_descriptors(handle), desc.nbytes)
        self.keypoints, self.descriptors = kp, desc
        self.kcontrast = L.oracle_akaze_kcontrast(handle)
        self.levels = []
        for i in range(L.oracle_akaze_num_levels(handle)):
            info = (C.c_int32 * 8)()
            finfo = (C.c_float * 4)()
            L.oracle_akaze_level_info(handle, i, info, finfo)
            tau = (C.c_float * 64)()
            nt = L.oracle_akaze_level_tau(handle, i, tau, 64)
            self.levels.append(dict(w=info[0], h=info[1], octave=info[2], sublevel=info[3], sigma_size=info[4],
                                    border=info[5], nsteps=info[6], esigma=finfo[0], etime=finfo[1], ratio=finfo[2],
                                    kcontrast=finfo[3], tau=np.array(tau[:nt], np.float32)))
        self._keep = keep

    def plane(self, level, which):
        L = lib()
        lv = self.levels[level]
        p = L.oracle_akaze_plane(self._h, level, which)
        if not p:
            return None
        dt = np.uint8 if which >= PLANE_MASK0 else np.float32
        out = np.zeros((lv["h"], lv["w"]), dt)
        C.memmove(_ptr(out), p, out.nbytes)
        return out

    def gray(self):
        L = lib()
        p = L.oracle_akaze_gray(self._h)
        if not p:
            return None
        lv = self.levels[0]
        out = np.zeros((lv["h"], lv["w"]), np.float32)
        C.memmove(_ptr(out), p, out.nbytes)
        return out

    def close(self):
        if self._h:
            lib().oracle_akaze_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def akaze(img, max_points=(1 << 18) - 1, keep_planes=False):
    """img: HxW (gray) or HxWx{3,4} uint8 (BGR / BGRA). Mirrors feature_extraction/src/lib.rs:61-92."""
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    h = lib().oracle_akaze_run(_ptr(img), img.shape[0], img.shape[1], ch, img.strides[0], int(max_points), int(keep_planes))
    if not h:
        raise ValueError("oracle_akaze_run rejected the arguments")
    return AkazeResult(h, keep_planes)


def knn_hamming(q, t, k):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((q.shape[0], k), np.int32)
    dist = np.zeros((q.shape[0], k), np.int32)
    lib().oracle_knn_hamming(_ptr(q), q.shape[0], q.strides[0], _ptr(t), t.shape[0], t.strides[0] if t.shape[0] else q.shape[1],
                             q.shape[1], k, _ptr(idx), _ptr(dist))
    return idx, dist


def get_knn_matches(q, t, k, filter_strength):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros(max(q.shape[0], 1), DMATCH_DTYPE)
    n = lib().oracle_get_knn_matches(_ptr(q), q.shape[0], q.strides[0] if q.shape[0] else 0, _ptr(t), t.shape[0],
                                     t.strides[0] if t.shape[0] else 0, q.shape[1], k, filter_strength, _ptr(out))
    if n < 0:
        raise RuntimeError(f"oracle error {n}")
    return out[:n].copy()


def get_bruteforce_matches(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros(max(q.shape[0], 1), DMATCH_DTYPE)
    n = lib().oracle_get_bruteforce_matches(_ptr(q), q.shape[0], q.strides[0] if q.shape[0] else 0, _ptr(t), t.shape[0],
                                            t.strides[0] if t.shape[0] else 0, q.shape[1], _ptr(out))
    if n < 0:
        raise RuntimeError(f"oracle error {n}")
    return out[:n].copy()


def get_points_from_matches(kp1, kp2, matches, bug_compatible=False):
    kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE)
    kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    p1 = np.zeros((len(matches), 2), np.float32)
    p2 = np.zeros((len(matches), 2), np.float32)
    rc = lib().oracle_get_points_from_matches(_ptr(kp1), len(kp1), _ptr(kp2), len(kp2), _ptr(matches), len(matches),
                                              int(bug_compatible), _ptr(p1), _ptr(p2))
    if rc < 0:
        raise RuntimeError(f"oracle error {rc}")
    return p1, p2


def find_homography(src, dst, method=0, thr=3.0, max_iters=2000, confidence=0.995):
    """Returns (found, H 3x3 f64, mask n u8)."""
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
    H = np.zeros(9, np.float64)
    mask = np.zeros(len(src), np.uint8)
    rc = lib().oracle_find_homography(_ptr(src), _ptr(dst), len(src), method, thr, max_iters, confidence, _ptr(H), _ptr(mask))
    if rc < 0:
        raise RuntimeError(f"oracle error {rc}")
    return rc == 1, H.reshape(3, 3), mask


def homography_4pt(src, dst):
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
    H = np.zeros(9, np.float64)
    rc = lib().oracle_homography_4pt(_ptr(src), _ptr(dst), len(src), _ptr(H))
    return rc, H.reshape(3, 3)


def ransac_samples(src, dst, iters):
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
    idx = np.zeros((iters, 4), np.int32)
    n = lib().oracle_ransac_samples(_ptr(src), _ptr(dst), len(src), iters, _ptr(idx))
    return idx[:n]


def raster_to_mat(rgba, w, h):
    rgba = np.ascontiguousarray(rgba, np.uint8).reshape(-1, 4)
    out = np.zeros((max(h, 0), max(w, 0), 4), np.uint8)
    rc = lib().oracle_raster_to_mat(_ptr(rgba), rgba.shape[0], w, h, _ptr(out))
    if rc < 0:
        raise ValueError("MatError::Unknown (len != w*h)")
    return out


def gamma_correction(v):
    """geotiff_extractor mod.rs:402-408: returns the corrected value or None (PixelConversion::GammaOutOfRange)."""
    ok = C.c_int(0)
    r = lib().oracle_gamma_correction(float(v), C.byref(ok))
    return np.float32(r) if ok.value else None


def f32_to_u8(v, mn, mx):
    """mod.rs:410-422: u8 or None (Err)."""
    ok = C.c_int(0)
    r = lib().oracle_f32_to_u8(float(v), float(mn), float(mx), C.byref(ok))
    return r if ok.value else None


def band_merger(red, green, blue, minmax):
    """mod.rs:346-378: minmax = (red_min, red_max, green_min, green_max, blue_min, blue_max) as f64. Returns n x 4 RGBA u8."""
    r, g, b = (np.ascontiguousarray(a, np.float32).ravel() for a in (red, green, blue))
    mm = np.ascontiguousarray(minmax, np.float64)
    out = np.zeros((len(r), 4), np.uint8)
    lib().oracle_band_merger(_ptr(r), _ptr(g), _ptr(b), len(r), _ptr(mm), _ptr(out))
    return out


def warp_perspective(src, M, size=None):
    """homographier mod.rs:271-300 on an HxW[xC] image of u8 or f32 elements (C = 1, 3, 4); size = (width, height) or None (source size).
    HxWx4 u8 goes through the dedicated 8UC4 restatement, everything else through the generic one (tests compare the two on 8UC4)."""
    src = np.asarray(src)
    dt = np.float32 if src.dtype == np.float32 else np.uint8
    src = np.ascontiguousarray(src, dt)
    h, w = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = (w, h) if size is None else size
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((dh, dw) if src.ndim == 2 else (dh, dw, ch), dt)
    if dt == np.uint8 and ch == 4:
        rc = lib().oracle_warp_perspective_8uc4(_ptr(src), h, w, _ptr(M), dh, dw, _ptr(out))
    else:
        rc = lib().oracle_warp_perspective_any(_ptr(src), h, w, ch, np.dtype(dt).itemsize, _ptr(M), dh, dw, _ptr(out))
    if rc != 0:
        raise RuntimeError("singular matrix" if rc == -1 else "unsupported element type")
    return out


def warp_perspective_generic(src, M, size=None):
    """the generic restatement on any supported type, 8UC4 included (cross-check of the two texts)"""
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = (w, h) if size is None else size
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((dh, dw) if src.ndim == 2 else (dh, dw, ch), src.dtype)
    if lib().oracle_warp_perspective_any(_ptr(src), h, w, ch, src.dtype.itemsize, _ptr(M), dh, dw, _ptr(out)) != 0:
        raise RuntimeError("singular matrix")
    return out


def knn_l2(q, t, k):
    """BFMatcher(NORM_L2).knnMatch semantics for float descriptors (BASELINE config 3; no reference call site)."""
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    idx = np.zeros((q.shape[0], k), np.int32)
    dist = np.zeros((q.shape[0], k), np.float32)
    lib().oracle_knn_l2(_ptr(q), q.shape[0], _ptr(t), t.shape[0], q.shape[1], k, _ptr(idx), _ptr(dist))
    return idx, dist


def solve_pnp_ransac(obj, img, K, iterations=100, reproj_thr=8.0, confidence=0.99, method=1):
    """Returns (rc, rvec[3], tvec[3], inliers int32[]); rc 1 found / 0 none / <0 cv error code."""
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    inl = np.zeros(max(len(obj), 1), np.int32)
    n_inl = C.c_int(0)
    rc = lib().oracle_solve_pnp_ransac(_ptr(obj), _ptr(img), len(obj), _ptr(K), iterations, reproj_thr, confidence, method, _ptr(rvec),
                                   _ptr(tvec), _ptr(inl), C.cast(C.byref(n_inl), C.c_void_p))
    return rc, rvec, tvec, inl[:n_inl.value].copy()


def solve_pnp_epnp(obj, img, K):
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    rc = lib().oracle_solve_pnp_epnp(_ptr(obj), _ptr(img), len(obj), _ptr(K), _ptr(rvec), _ptr(tvec))
    return rc, rvec, tvec


def solve_pnp_ippe(obj, img, K):
    """cv::solvePnP(..., SOLVEPNP_IPPE) on double points: (1 = a pose, rvec, tvec); 0 for object points that are not coplanar."""
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    rc = lib().oracle_solve_pnp_ippe(_ptr(obj), _ptr(img), len(obj), _ptr(K), _ptr(rvec), _ptr(tvec))
    return rc, rvec, tvec


def solve_pnp_sqpnp(obj, img, K):
    """cv::solvePnP(..., SOLVEPNP_SQPNP) on double points: (1 = a pose, rvec, tvec)."""
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    rc = lib().oracle_solve_pnp_sqpnp(_ptr(obj), _ptr(img), len(obj), _ptr(K), _ptr(rvec), _ptr(tvec))
    return rc, rvec, tvec


def pnp_ransac_samples(n, iters):
    idx = np.zeros((iters, 5), np.int32)
    lib().oracle_pnp_ransac_samples(n, iters, _ptr(idx))
    return idx


def pnp_hypothesis(obj, img, idx5, K):
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    idx5 = np.ascontiguousarray(idx5, np.int32)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    lib().oracle_pnp_hypothesis(_ptr(obj), _ptr(img), _ptr(idx5), _ptr(K), _ptr(rvec), _ptr(tvec))
    return rvec, tvec


def rodrigues(x):
    x = np.ascontiguousarray(x, np.float64)
    if x.size == 9:
        out = np.zeros(3)
        lib().oracle_rodrigues(_ptr(x), 1, _ptr(out))
        return out
    out = np.zeros(9)
    lib().oracle_rodrigues(_ptr(x), 0, _ptr(out))
    return out.reshape(3, 3)


def det_acos(c):
    c = np.ascontiguousarray(c, np.float64).reshape(-1)
    out = np.zeros_like(c)
    lib().oracle_det_acos(_ptr(c), len(c), _ptr(out))
    return out


def world_coordinates(xy, dataset_gt, elevation_gt=None, elevation=None):
    """elevationdb.rs:64-104, batched. Returns (rc, xyz n x 3)."""
    xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
    dgt = np.ascontiguousarray(dataset_gt, np.float64)
    out = np.zeros((len(xy), 3))
    if elevation_gt is None:
        rc = lib().oracle_world_coordinates(_ptr(xy), len(xy), _ptr(dgt), None, None, 0, 0, _ptr(out))
    else:
        egt = np.ascontiguousarray(elevation_gt, np.float64)
        el = np.ascontiguousarray(elevation, np.float64)
        rc = lib().oracle_world_coordinates(_ptr(xy), len(xy), _ptr(dgt), _ptr(egt), _ptr(el), el.shape[1], el.shape[0], _ptr(out))
    return rc, out


def pnp_ransac_samples4(n, iters):
    idx = np.zeros((iters, 4), np.int32)
    lib().oracle_pnp_ransac_samples4(n, iters, _ptr(idx))
    return idx


def pnp_p3p_hypothesis(obj, img, idx4, K):
    """Returns (found, rvec, tvec) of solvePnP(P3P) on the four correspondences idx4 (converted to float as in the RANSAC loop)."""
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    idx4 = np.ascontiguousarray(idx4, np.int32)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    ok = lib().oracle_pnp_p3p_hypothesis(_ptr(obj), _ptr(img), _ptr(idx4), _ptr(K), _ptr(rvec), _ptr(tvec))
    return ok == 1, rvec, tvec


def pnp_ap3p_hypothesis(obj, img, idx4, K):
    """SOLVEPNP_AP3P's kernel on one 4-point sample: the first three points solve, the fourth ranks. Returns (ok, rvec, tvec)."""
    obj = np.ascontiguousarray(obj, np.float64)
    img = np.ascontiguousarray(img, np.float64)
    idx4 = np.ascontiguousarray(idx4, np.int32)
    K = np.ascontiguousarray(K, np.float64)
    rvec, tvec = np.zeros(3), np.zeros(3)
    ok = lib().oracle_pnp_ap3p_hypothesis(_ptr(obj), _ptr(img), _ptr(idx4), _ptr(K), _ptr(rvec), _ptr(tvec))
    return bool(ok), rvec, tvec
