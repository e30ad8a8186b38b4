"""Host-side mirror of the reference crate `feature_extraction` (/root/reference/feature_extraction/src/lib.rs).

Same public names, argument order and error behaviour; `Mat`/`Vector<...>` become numpy arrays (u8 images and
descriptor matrices, structured arrays with cv::KeyPoint / cv::DMatch layout). Every function forwards to the C ABI
of libapds_hip.so; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DMATCH_DTYPE, KEYPOINT_DTYPE, ApdsError, check, lib, ptr, take

MAX_POINTS_SHIFT = 18                       # lib.rs:12
MAX_POINTS = (1 << MAX_POINTS_SHIFT) - 1    # lib.rs:13


class DbKeypoints:
    """lib.rs:20-31 — one row of the `keypoint` table."""
    __slots__ = ("x_coord", "y_coord", "size", "angle", "response", "octave", "class_id", "descriptor", "image_id")

    def __init__(self, x_coord, y_coord, size, angle, response, octave, class_id, descriptor, image_id):
        self.x_coord, self.y_coord, self.size, self.angle, self.response = x_coord, y_coord, size, angle, response
        self.octave, self.class_id, self.descriptor, self.image_id = octave, class_id, descriptor, image_id

    def __repr__(self):
        return (f"DbKeypoints(x_coord={self.x_coord}, y_coord={self.y_coord}, size={self.size}, angle={self.angle}, "
                f"response={self.response}, octave={self.octave}, class_id={self.class_id}, image_id={self.image_id})")


class ExtractedKeyPoint:
    """lib.rs:15-18 — keypoints (K x cv::KeyPoint) and descriptors (K x 61 u8). Fields are private in the
    reference; `to_db_type` is the only way out there, so both spellings are offered here."""

    def __init__(self, keypoints, descriptors):
        self._keypoints = keypoints
        self._descriptors = descriptors

    @property
    def keypoints(self):
        return self._keypoints

    @property
    def descriptors(self):
        return self._descriptors

    def to_db_type(self, image_id):
        """lib.rs:34-58"""
        kp, d = self._keypoints, self._descriptors
        return [DbKeypoints(float(kp["x"][i]), float(kp["y"][i]), float(kp["size"][i]), float(kp["angle"][i]),
                            float(kp["response"][i]), int(kp["octave"][i]), int(kp["class_id"][i]), bytes(d[i]), image_id)
                for i in range(len(kp))]


def akaze_keypoint_descriptor_extraction_def(img, max_points=None):
    """lib.rs:61-92 — AKAZE(MLDB, 0, 3, 0.001, 4 octaves, 4 layers, PM_G2, max_points or MAX_POINTS).detectAndCompute.
    img: HxW, HxWx3 (BGR) or HxWx4 (BGRA) uint8."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or img.size == 0:
        raise ApdsError(_lib.ERR_ASSERT, "image must be a non-empty uint8 HxW[xC] array")
    if not img.flags["C_CONTIGUOUS"] and not (img.ndim == 3 and img.strides[2] == 1 and img.strides[1] == img.shape[2]):
        img = np.ascontiguousarray(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    kps, desc = C.c_void_p(), C.c_void_p()
    n, nb = C.c_int(0), C.c_int(0)
    check(lib().apds_akaze_extract(ptr(img), img.shape[0], img.shape[1], ch, img.strides[0],
                                   MAX_POINTS if max_points is None else int(max_points),
                                   C.byref(kps), C.byref(desc), C.byref(n), C.byref(nb)))
    k = take(kps, n.value, KEYPOINT_DTYPE)
    d = take(desc, n.value * nb.value, np.uint8).reshape(n.value, nb.value)
    return ExtractedKeyPoint(k, d)


def akaze_keypoint_descriptor_extraction_batch(imgs, max_points=None):
    """lib.rs:61-92 for a batch of equal-sized images (array [B, H, W] or [B, H, W, C] uint8, or a list of such images) in ONE library
    call: the batch goes through every kernel's grid together. Returns a list of ExtractedKeyPoint, each exactly what
    akaze_keypoint_descriptor_extraction_def returns for that image. This is how a caller that extracts one tile per task
    (preprocessor/src/main.rs:227-245,277) should hand its tiles over: a single small tile is launch-latency-bound."""
    a = np.ascontiguousarray(np.stack([np.asarray(i) for i in imgs]) if isinstance(imgs, (list, tuple)) else np.asarray(imgs))
    if a.dtype != np.uint8 or a.ndim not in (3, 4) or a.size == 0:
        raise ApdsError(_lib.ERR_ASSERT, "images must be a non-empty uint8 [B, H, W[, C]] array")
    b, h, w = a.shape[0], a.shape[1], a.shape[2]
    ch = 1 if a.ndim == 3 else a.shape[3]
    kps, desc = C.c_void_p(), C.c_void_p()
    counts, nb = (C.c_int * b)(), C.c_int(0)
    check(lib().apds_akaze_extract_batch(ptr(a), b, a.strides[0], h, w, ch, a.strides[1], MAX_POINTS if max_points is None else int(max_points),
                                         C.byref(kps), C.byref(desc), counts, C.byref(nb)))
    total = sum(counts)
    k = take(kps, total, KEYPOINT_DTYPE)
    d = take(desc, total * nb.value, np.uint8).reshape(total, nb.value)
    out, off = [], 0
    for c in counts:
        out.append(ExtractedKeyPoint(k[off:off + c].copy(), d[off:off + c].copy()))
        off += c
    return out


def tile_keypoint_descriptor_extraction(red, green, blue, min_max, max_points=None):
    """One preprocessor tile (preprocessor/src/main.rs:258-277): to_rgb's band_merger, raster_to_mat and the extraction above in one
    library call — red/green/blue are equal-shape 2-D float32 views (row-strided views into the mosaic are taken as they are), the
    RGBA/BGRA image exists on the device only. Same result as the three separate calls."""
    bands = [np.asarray(b) for b in (red, green, blue)]
    h, w = bands[0].shape
    if any(b.dtype != np.float32 or b.ndim != 2 or b.shape != (h, w) for b in bands) or h == 0 or w == 0:
        raise ApdsError(_lib.ERR_ASSERT, "bands must be equal, non-empty 2-D float32 arrays")
    stride = bands[0].strides[0]
    if any(b.strides[1] != 4 or b.strides[0] != stride or stride % 4 for b in bands):
        bands = [np.ascontiguousarray(b) for b in bands]
        stride = w * 4
    mm = min_max.as_array()
    kps, desc = C.c_void_p(), C.c_void_p()
    n, nb = C.c_int(0), C.c_int(0)
    check(lib().apds_tile_extract(bands[0].ctypes.data, bands[1].ctypes.data, bands[2].ctypes.data, h, w, stride // 4, ptr(mm),
                                  MAX_POINTS if max_points is None else int(max_points), C.byref(kps), C.byref(desc), C.byref(n), C.byref(nb)))
    k = take(kps, n.value, KEYPOINT_DTYPE)
    d = take(desc, n.value * nb.value, np.uint8).reshape(n.value, nb.value)
    return ExtractedKeyPoint(k, d)


def tiles_keypoint_descriptor_extraction(windows, min_max, max_points=None):
    """A batch of preprocessor tiles in ONE library call: `windows` = list of [3, h, w] float32 band windows of one size (strided views
    into the mosaic are taken as they are). Returns a list of ExtractedKeyPoint, each exactly what tile_keypoint_descriptor_extraction
    returns for that tile."""
    wins = [np.asarray(w) for w in windows]
    if not wins:
        return []
    _, h, w = wins[0].shape
    stride = wins[0].strides[1]
    ok = all(x.dtype == np.float32 and x.ndim == 3 and x.shape == (3, h, w) and x.strides[2] == 4 and x.strides[1] == stride and stride % 4 == 0 for x in wins)
    if not ok:
        wins = [np.ascontiguousarray(x, np.float32) for x in wins]
        if any(x.shape != (3, h, w) for x in wins):
            raise ApdsError(_lib.ERR_ASSERT, "tile windows must all be [3, h, w] float32 of one size")
        stride = w * 4
    b = len(wins)
    ptrs = [(C.c_void_p * b)(*[x[band].ctypes.data for x in wins]) for band in range(3)]
    mm = min_max.as_array()
    kps, desc = C.c_void_p(), C.c_void_p()
    counts, nb = (C.c_int * b)(), C.c_int(0)
    check(lib().apds_tile_extract_batch(ptrs[0], ptrs[1], ptrs[2], b, h, w, stride // 4, ptr(mm), MAX_POINTS if max_points is None else int(max_points),
                                        C.byref(kps), C.byref(desc), counts, C.byref(nb)))
    total = sum(counts)
    k = take(kps, total, KEYPOINT_DTYPE)
    d = take(desc, total * nb.value, np.uint8).reshape(total, nb.value)
    out, off = [], 0
    for c in counts:
        out.append(ExtractedKeyPoint(k[off:off + c].copy(), d[off:off + c].copy()))
        off += c
    return out


def _desc(a):
    a = np.ascontiguousarray(a, np.uint8)
    if a.ndim != 2:
        raise ApdsError(_lib.ERR_ASSERT, "descriptor matrix must be 2-D uint8")
    return a


def get_knn_matches(origin_desc, target_desc, k, filter_strength):
    """lib.rs:94-114 — BFMatcher(NORM_HAMMING).knnMatch(origin -> target, k) + Lowe ratio test."""
    q, t = _desc(origin_desc), _desc(target_desc)
    if q.shape[0] and t.shape[0] and q.shape[1] != t.shape[1]:
        raise ApdsError(_lib.ERR_ASSERT, "descriptor lengths differ")
    out, n = C.c_void_p(), C.c_int(0)
    nb = q.shape[1] if q.shape[0] else t.shape[1]
    check(lib().apds_get_knn_matches(ptr(q), q.shape[0], ptr(t), t.shape[0], nb, int(k), float(filter_strength), C.byref(out), C.byref(n)))
    return take(out, n.value, DMATCH_DTYPE)


def get_bruteforce_matches(origin_desc, target_desc):
    """lib.rs:116-126 — BFMatcher(NORM_HAMMING, crossCheck=true).match(origin -> target)."""
    q, t = _desc(origin_desc), _desc(target_desc)
    if q.shape[0] and t.shape[0] and q.shape[1] != t.shape[1]:
        raise ApdsError(_lib.ERR_ASSERT, "descriptor lengths differ")
    out, n = C.c_void_p(), C.c_int(0)
    nb = q.shape[1] if q.shape[0] else t.shape[1]
    check(lib().apds_get_bruteforce_matches(ptr(q), q.shape[0], ptr(t), t.shape[0], nb, C.byref(out), C.byref(n)))
    return take(out, n.value, DMATCH_DTYPE)


def knn_match(query_desc, train_desc, k):
    """BFMatcher.knnMatch itself (lib.rs:103): (idx, dist) int32 arrays of shape (nq, k)."""
    q, t = _desc(query_desc), _desc(train_desc)
    idx = np.zeros((q.shape[0], k), np.int32)
    dist = np.zeros((q.shape[0], k), np.int32)
    check(lib().apds_knn_match(ptr(q), q.shape[0], ptr(t), t.shape[0], q.shape[1], int(k), ptr(idx), ptr(dist)))
    return idx, dist


def get_points_from_matches(img1_keypoints, img2_keypoints, matches, bug_compatible=False):
    """lib.rs:161-180. The reference indexes img1 with `img_idx` (always 0) and converts img1's points twice
    (:169, :176-177); pass bug_compatible=True for that exact output, the default is the intended gather."""
    k1 = np.ascontiguousarray(img1_keypoints, KEYPOINT_DTYPE)
    k2 = np.ascontiguousarray(img2_keypoints, KEYPOINT_DTYPE)
    m = np.ascontiguousarray(matches, DMATCH_DTYPE)
    p1 = np.zeros((len(m), 2), np.float32)
    p2 = np.zeros((len(m), 2), np.float32)
    check(lib().apds_get_points_from_matches(ptr(k1), len(k1), ptr(k2), len(k2), ptr(m), len(m), int(bug_compatible), ptr(p1), ptr(p2)))
    return p1, p2


def l2_knn_match(query_desc, train_desc, k):
    """NOT in the reference (it matches Hamming only, lib.rs:101,121): BFMatcher(NORM_L2).knnMatch for float descriptors
    (BASELINE config 3), computed as an MFMA distance GEMM. Returns (idx int32, dist float32) of shape (nq, k)."""
    q = np.ascontiguousarray(query_desc, np.float32)
    t = np.ascontiguousarray(train_desc, np.float32)
    if q.ndim != 2 or t.ndim != 2 or (q.shape[0] and t.shape[0] and q.shape[1] != t.shape[1]):
        raise ApdsError(_lib.ERR_ASSERT, "descriptor matrices must be 2-D float32 with equal row length")
    idx = np.zeros((q.shape[0], k), np.int32)
    dist = np.zeros((q.shape[0], k), np.float32)
    check(lib().apds_l2_knn_match(ptr(q), q.shape[0], ptr(t), t.shape[0], q.shape[1], int(k), ptr(idx), ptr(dist)))
    return idx, dist
