"""Host-side mirror of the reference crate `homographier` (/root/reference/homographier/src/homographier/mod.rs)
for the rows on the hot path: HomographyMethod, MatError, Cmat, raster_to_mat, find_homography_mat."""
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
    """mod.rs:271-300 — warpPerspective(INTER_LINEAR, BORDER_CONSTANT (1,1,1,1)). src: Cmat of HxWx4 u8, m: Cmat<f64> 3x3,
    size: (width, height) or None (= source size). Returns a Cmat of the same element type."""
    img = np.ascontiguousarray(src.mat, np.uint8)
    if img.ndim != 3 or img.shape[2] != 4:
        raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "warp_image_perspective is implemented for Vec4b images"))
    M = np.ascontiguousarray(m.mat, np.float64)
    if M.shape != (3, 3):
        raise MatError("Opencv", ApdsError(_lib.ERR_ASSERT, "m must be 3x3"))
    h, w = img.shape[:2]
    dw, dh = (w, h) if size is None else (int(size[0]), int(size[1]))
    out = np.zeros((dh, dw, 4), np.uint8)
    try:
        check(lib().apds_warp_perspective(ptr(img), h, w, 4, ptr(M), dh, dw, ptr(out)))
    except ApdsError as e:
        raise MatError("Opencv", e)
    return Cmat(out, np.uint8, 4)
