"""Host-side mirror of the reference crate `homographier` (/root/reference/homographier/src/homographier/mod.rs)
for the rows on the hot path: HomographyMethod, MatError, Cmat, raster_to_mat, find_homography_mat, and the crate's remaining
public functions warp_image_perspective and pnp_solver_ransac."""
import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import ApdsError, check, lib, ptr


class HomographyMethod(enum.IntEnum):
    """mod.rs:25-31"""
    Default = 0
    LMEDS = 4
    RANSAC = 8
    RHO = 16


class MatError(Exception):
    """mod.rs:33-44 — kind in {"Opencv", "Empty", "Jagged", "Unknown"}; `inner` carries the ApdsError for Opencv."""

    def __init__(self, kind, inner=None):
        super().__init__(kind if inner is None else f"{kind}({inner})")
        self.kind = kind
        self.inner = inner


class Cmat:
    """mod.rs:70-146 — checked matrix: guarantees the wrapped array is non-empty and of the declared element type.
    `mat` is a numpy array (rows x cols [x channels])."""

    def __init__(self, mat, dtype, channels=1):
        mat = np.asarray(mat)
        if mat.dtype != np.dtype(dtype) or (mat.ndim == 3 and mat.shape[2] != channels) or (mat.ndim == 2 and channels != 1):
            raise MatError("Empty")           # mod.rs:115-118: type mismatch -> MatError::Empty
        if mat.ndim < 2 or mat.size == 0:
            raise MatError("Empty")           # mod.rs:85-90: dims()==0
        self.mat = mat
        self.dtype = np.dtype(dtype)
        self.channels = channels

    @classmethod
    def from_2d_slice(cls, rows, dtype, channels=1):
        """mod.rs:94-100; a jagged slice is an OpenCV error there."""
        lens = {len(r) for r in rows}
        if len(lens) > 1:
            raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "jagged 2-D slice"))
        return cls(np.array(rows, dtype=dtype), dtype, channels)

    @classmethod
    def zeros(cls, rows, cols, dtype, channels=1):
        shape = (rows, cols) if channels == 1 else (rows, cols, channels)
        return cls(np.zeros(shape, dtype), dtype, channels)

    def at_2d(self, row, col):
        """mod.rs:130-137 — note the reference compares row with WIDTH and col with HEIGHT, using '>'."""
        height, width = self.mat.shape[0], self.mat.shape[1]
        if row > width or col > height:
            raise MatError("Opencv", ApdsError(_lib.ERR_OUT_OF_RANGE, ""))
        if not (0 <= row < height and 0 <= col < width):
            raise MatError("Opencv", ApdsError(_lib.ERR_OUT_OF_RANGE, "index out of range"))
        return self.mat[row, col]


def raster_to_mat(pixels, w, h):
    """mod.rs:183-197 — RGBA8 slice -> Cmat<Vec4b> (BGRA rows). MatError::Unknown if len(pixels) != w*h."""
    px = np.ascontiguousarray(pixels, np.uint8).reshape(-1, 4)
    if w <= 0 or h <= 0 or px.shape[0] != w * h:
        raise MatError("Unknown")
    out = np.zeros((h, w, 4), np.uint8)
    try:
        check(lib().apds_raster_to_mat(ptr(px), px.shape[0], int(w), int(h), ptr(out)))
    except ApdsError as e:
        raise MatError("Unknown" if e.code == _lib.ERR_BAD_ARG else "Opencv", e)
    return Cmat(out, np.uint8, 4)


def find_homography_mat(input_pts, reference_pts, method=None, reproj_threshold=None):
    """mod.rs:231-259 — returns (Cmat<f64> 3x3, Optional[Cmat<u8>] n x 1). The mask is returned for RANSAC and LMEDS only."""
    src = np.ascontiguousarray(input_pts, np.float32).reshape(-1, 2)
    dst = np.ascontiguousarray(reference_pts, np.float32).reshape(-1, 2)
    method_i = int(HomographyMethod.Default if method is None else method)   # mod.rs:241
    thr = 3.0 if reproj_threshold is None else float(reproj_threshold)          # mod.rs:248
    H = np.zeros(9, np.float64)
    mask = np.zeros(max(len(src), 1), np.uint8)
    if len(src) != len(dst):
        raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "point lists differ in length"))
    try:
        check(lib().apds_find_homography(ptr(src), ptr(dst), len(src), method_i, thr, ptr(H), ptr(mask)))
    except ApdsError as e:
        if e.code == _lib.ERR_EMPTY:
            raise MatError("Empty")            # empty H -> Cmat::new fails (mod.rs:258,114-119)
        raise MatError("Opencv", e)
    out_mask = None
    if method is not None and int(method) in (HomographyMethod.RANSAC, HomographyMethod.LMEDS):
        out_mask = Cmat(mask[:len(src)].reshape(-1, 1), np.uint8)
    return Cmat(H.reshape(3, 3), np.float64), out_mask


def warp_image_perspective(src, m, size=None):
    """mod.rs:271-300 — warpPerspective(INTER_LINEAR, BORDER_CONSTANT (1,1,1,1)), generic over the element type as the reference
    (`warp_image_perspective<T: DataType>`): src is a Cmat of HxW or HxWxC (C = 1, 3, 4) u8 or f32 elements, m: Cmat<f64> 3x3,
    size: (width, height) or None (= source size). Returns a Cmat of the same element type."""
    a = np.asarray(src.mat)
    if a.dtype not in (np.uint8, np.float32) or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3, 4)):
        raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "warp_image_perspective serves u8 and f32 images of 1, 3 or 4 channels"))
    img = np.ascontiguousarray(a)
    M = np.ascontiguousarray(m.mat, np.float64)
    if M.shape != (3, 3):
        raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "m must be 3x3"))
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    dw, dh = (w, h) if size is None else (int(size[0]), int(size[1]))
    out = np.zeros((dh, dw) if img.ndim == 2 else (dh, dw, ch), img.dtype)
    try:
        fn = lib().apds_warp_perspective if img.dtype == np.uint8 else lib().apds_warp_perspective_f32
        check(fn(ptr(img), h, w, ch, ptr(M), dh, dw, ptr(out)))
    except ApdsError as e:
        raise MatError("Opencv", e)
    return Cmat(out, img.dtype.type, ch if img.ndim == 3 else 1)


class SolvePnPMethod(enum.IntEnum):
    """opencv::calib3d::SolvePnPMethod values the reference can pass (mod.rs:4,327)."""
    SOLVEPNP_ITERATIVE = 0
    SOLVEPNP_EPNP = 1
    SOLVEPNP_P3P = 2
    SOLVEPNP_DLS = 3       # OpenCV 4 runs EPnP for DLS and UPNP
    SOLVEPNP_UPNP = 4
    SOLVEPNP_AP3P = 5
    SOLVEPNP_IPPE = 6          # planar targets: inliers that are not coplanar give no pose (None)
    SOLVEPNP_IPPE_SQUARE = 7   # solvePnPRansac: P3P for four points, otherwise the final solve's npoints == 4 assertion (an error)
    SOLVEPNP_SQPNP = 8


class ImgObjCorrespondence:
    """mod.rs:52-65 — a 3D object point and the 2D image point it projects to."""

    def __init__(self, obj_point, img_point):
        self.obj_point = tuple(float(v) for v in obj_point)
        self.img_point = tuple(float(v) for v in img_point)
        if len(self.obj_point) != 3 or len(self.img_point) != 2:
            raise ValueError("obj_point is (x, y, z), img_point is (x, y)")


class PNPRANSACSolution:
    """mod.rs:46-51 — rvec, tvec: Cmat<f64> 3x1; inliers: Cmat<i32> n_inliers x 1 (indices into point_correspondences)."""

    def __init__(self, rvec, tvec, inliers):
        self.rvec, self.tvec, self.inliers = rvec, tvec, inliers


def pnp_solver_ransac(point_correspondences, camera_intrinsic, iter_count, reproj_thres, confidence, dist_coeffs=None, method=None):
    """mod.rs:320-369 — solvePnPRansac(useExtrinsicGuess = false). Returns a PNPRANSACSolution, or None when no pose was found
    (Ok(None)); raises MatError("Opencv") for fewer than 4 correspondences (mod.rs:627-638). `dist_coeffs` is accepted and ignored,
    as in the reference, which shadows it with zeros(4,1) before the call (mod.rs:344). Built: SOLVEPNP_EPNP (the default),
    SOLVEPNP_P3P, SOLVEPNP_AP3P, SOLVEPNP_ITERATIVE (EPnP RANSAC, then solvePnP(ITERATIVE) over the inliers without an extrinsic guess),
    SOLVEPNP_SQPNP (EPnP RANSAC, then SQPnP over the inliers) and SOLVEPNP_DLS / SOLVEPNP_UPNP (EPnP, as in OpenCV 4); four correspondences go through P3P
    whatever the method, as in OpenCV."""
    del dist_coeffs
    n = len(point_correspondences)
    obj = np.ascontiguousarray([p.obj_point for p in point_correspondences], np.float64).reshape(n, 3)
    img = np.ascontiguousarray([p.img_point for p in point_correspondences], np.float64).reshape(n, 2)
    K = np.ascontiguousarray(camera_intrinsic.mat, np.float64)
    if K.shape != (3, 3):
        raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "camera_intrinsic must be 3x3"))
    method_i = int(SolvePnPMethod.SOLVEPNP_EPNP if method is None else method)      # mod.rs:360
    rvec, tvec = np.zeros((3, 1), np.float64), np.zeros((3, 1), np.float64)
    inliers = np.zeros(max(n, 1), np.int32)
    n_inl, found = C.c_int(0), C.c_int(0)
    try:
        check(lib().apds_pnp_solver_ransac(ptr(obj), ptr(img), n, ptr(K), int(iter_count), float(reproj_thres), float(confidence), method_i,
                                           ptr(rvec), ptr(tvec), ptr(inliers), C.byref(n_inl), C.byref(found)))
    except ApdsError as e:
        raise MatError("Opencv", e)
    if not found.value:
        return None
    return PNPRANSACSolution(Cmat(rvec, np.float64), Cmat(tvec, np.float64), Cmat(inliers[:n_inl.value].reshape(-1, 1).copy(), np.int32))
