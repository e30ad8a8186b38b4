"""ctypes binding of libapds_hip.so (C ABI: include/apds.h). Fails loudly when the library is missing."""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APDS_LIB_PATH") or os.path.join(_HERE, "libapds_hip.so")   # (override: A/B runs of differently built libraries)
_LIB = None

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("img_idx", "<i4"), ("distance", "<f4")])
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16

ERR_INTERNAL, ERR_NOMEM, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_OUT_OF_RANGE, ERR_ASSERT, ERR_EMPTY = -2, -4, -5, -216, -211, -215, -1000
ERR_NOT_IMPLEMENTED = -213

# every symbol include/apds.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "apds_last_error", "apds_free", "apds_device_count", "apds_set_device", "apds_build_info", "apds_thread_release", "apds_live_contexts", "apds_release_cached_memory",
    "apds_akaze_extract", "apds_akaze_extract_batch", "apds_dev_akaze_extract_batch", "apds_tile_extract", "apds_tile_extract_batch", "apds_get_knn_matches", "apds_get_bruteforce_matches", "apds_knn_match",
    "apds_get_points_from_matches", "apds_find_homography", "apds_find_homography_ex", "apds_raster_to_mat",
    "apds_dev_pack_descriptors", "apds_dev_hamming_topk", "apds_dev_merge_topk", "apds_dev_match_lds_cap", "apds_dev_match_last_launch_lds", "apds_dev_match_backend", "apds_dev_hamming_topk_backend", "apds_dev_ratio_filter",
    "apds_dev_cross_check", "apds_dev_akaze_extract", "apds_dev_points_from_matches", "apds_dev_find_homography",
    "apds_dev_valu_popcount_peak", "apds_dev_valu_peak", "apds_dev_valu_peak_modes", "apds_dev_last_kernel_ms", "apds_dev_timing_enable", "apds_akaze_debug_plane", "apds_stream_create", "apds_stream_destroy",
    "apds_band_merger", "apds_dev_band_merger", "apds_warp_perspective", "apds_warp_perspective_f32", "apds_pnp_solver_ransac", "apds_pnp_hypotheses", "apds_pnp_sqpnp", "apds_pnp_ippe", "apds_get_world_coordinates", "apds_l2_knn_match", "apds_dev_l2_topk", "apds_dev_l2_topk_ex",
    "apds_db_create", "apds_db_destroy", "apds_db_rows", "apds_db_insert_image", "apds_db_select", "apds_db_view", "apds_db_view_download", "apds_db_knn_match",
    "apds_comm_id_create", "apds_shard_create", "apds_shard_destroy", "apds_shard_info", "apds_shard_counts", "apds_shard_knn", "apds_shard_knn_replicated", "apds_shard_slot_create",
    "apds_shard_slot_destroy", "apds_shard_gather", "apds_shard_scan", "apds_shard_exchange_merge", "apds_db_shard",
    "apds_dev_alloc", "apds_dev_release", "apds_dev_upload", "apds_dev_download", "apds_stream_synchronize",
    "apds_pipeline_create", "apds_pipeline_submit", "apds_pipeline_poll", "apds_pipeline_stats", "apds_pipeline_destroy",
    "apds_dev_topk_state_create", "apds_dev_topk_state_destroy", "apds_dev_topk_prepass", "apds_dev_topk_scan", "apds_dev_topk_merge",
]

# multi-GPU sharded matcher (include/apds.h: apds_comm_id, apds_host_transport)
TRANSPORT_RCCL, TRANSPORT_LOOPBACK, TRANSPORT_HOST, TRANSPORT_DEVICE = 0, 1, 2, 3
COMM_ID_BYTES = 128


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * COMM_ID_BYTES)]


HOST_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
HOST_ALL_TO_ALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                              C.POINTER(C.c_size_t))


class HostTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", HOST_ALL_GATHER), ("all_to_all", HOST_ALL_TO_ALL)]


# apds_device_transport: the same two callbacks on DEVICE buffers, with the stream they are ordered on
DEV_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
DEV_ALL_TO_ALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                             C.POINTER(C.c_size_t), C.c_void_p)


class DeviceTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", DEV_ALL_GATHER), ("all_to_all", DEV_ALL_TO_ALL)]


class PipelineParams(C.Structure):
    """apds_pipeline_params (include/apds.h)"""
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("channels", C.c_int), ("max_points", C.c_int), ("n_slots", C.c_int), ("extract_workers", C.c_int),
                ("filter_strength", C.c_float), ("homography_method", C.c_int), ("reproj_threshold", C.c_double), ("max_iters", C.c_int),
                ("confidence", C.c_double), ("timing", C.c_int), ("match_lds_cap", C.c_int), ("match_stream", C.c_void_p), ("debug_extract_delay_ms", C.c_double)]


class FrameResult(C.Structure):
    """apds_frame_result"""
    _fields_ = [("frame", C.c_int64), ("status", C.c_int), ("n_keypoints", C.c_int), ("n_matches", C.c_int), ("n_inliers", C.c_int), ("homography_found", C.c_int),
                ("H", C.c_double * 9)]


class PipelineCounters(C.Structure):
    """apds_pipeline_counters"""
    _fields_ = [("frames_submitted", C.c_int64), ("frames_done", C.c_int64),
                ("hamming_topk_ms", C.c_double), ("hamming_topk_sample_ms", C.c_double), ("akaze_extract_ms", C.c_double), ("ransac_score_ms", C.c_double),
                ("hamming_topk_launches", C.c_int), ("hamming_topk_sample_launches", C.c_int), ("akaze_extract_calls", C.c_int), ("ransac_score_launches", C.c_int),
                ("match_gap_mean_ms", C.c_double), ("match_gaps", C.c_int), ("match_gaps_first_ms", C.c_float * 16),
                ("match_lds_cap_bytes", C.c_int), ("match_lds_cap_set_at_frame", C.c_int), ("match_lds_cap_gaps_ms", C.c_float * 6),
                ("extract_workers", C.c_int), ("slots", C.c_int), ("split_scan", C.c_int), ("world", C.c_int)]


PIPELINE_NOT_READY = 1


class ApdsError(RuntimeError):
    """Mirror of opencv::Error{code, message} as surfaced by the reference crates."""

    def __init__(self, code, message):
        super().__init__(f"apds error {code}: {message}")
        self.code = code
        self.message = message


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libapds_hip.so (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ApdsError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        # torch ships a HIP runtime of its own: a process that loads this library first and torch afterwards ends up with two runtimes, and
        # the second one to initialise finds no device (-216 from the first call). Loading torch first makes both use the same one.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        vp, i, f, d, sz, i64, u32 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_int64, C.c_uint32
        pp, ip = C.POINTER(C.c_void_p), C.POINTER(C.c_int)
        sig = {
            "apds_last_error": (C.c_char_p, []),
            "apds_free": (None, [vp]),
            "apds_device_count": (i, []),
            "apds_set_device": (i, [i]),
            "apds_thread_release": (i, []),
            "apds_live_contexts": (i, []),
            "apds_release_cached_memory": (i, []),
            "apds_build_info": (C.c_char_p, []),
            "apds_akaze_extract": (i, [vp, i, i, i, sz, i, pp, pp, ip, ip]),
            "apds_akaze_extract_batch": (i, [vp, i, sz, i, i, i, sz, i, pp, pp, ip, ip]),
            "apds_dev_akaze_extract_batch": (i, [vp, i, sz, i, i, i, sz, i, vp, vp, i, ip, vp]),
            "apds_tile_extract_batch": (i, [vp, vp, vp, i, i, i, sz, vp, i, pp, pp, ip, ip]),
            "apds_tile_extract": (i, [vp, vp, vp, i, i, sz, vp, i, pp, pp, ip, ip]),
            "apds_get_knn_matches": (i, [vp, i, vp, i, i, i, f, pp, ip]),
            "apds_get_bruteforce_matches": (i, [vp, i, vp, i, i, pp, ip]),
            "apds_knn_match": (i, [vp, i, vp, i, i, i, vp, vp]),
            "apds_get_points_from_matches": (i, [vp, i, vp, i, vp, i, i, vp, vp]),
            "apds_find_homography": (i, [vp, vp, i, i, d, vp, vp]),
            "apds_find_homography_ex": (i, [vp, vp, i, i, d, i, d, vp, vp]),
            "apds_raster_to_mat": (i, [vp, sz, i, i, vp]),
            "apds_dev_pack_descriptors": (i, [vp, i64, i, i64, vp, vp]),
            "apds_dev_hamming_topk": (i, [vp, i, vp, i64, u32, i, vp, vp]),
            "apds_dev_merge_topk": (i, [vp, i, i, i, vp, vp]),
            "apds_dev_match_lds_cap": (i, [i, ip]),
            "apds_dev_match_last_launch_lds": (i, [ip]),
            "apds_dev_match_backend": (i, [ip]),
            "apds_dev_hamming_topk_backend": (i, [vp, i, vp, i64, u32, i, vp, i, vp]),
            "apds_dev_ratio_filter": (i, [vp, i, i, f, vp, ip, vp]),
            "apds_dev_cross_check": (i, [vp, i64, i, vp, ip, vp]),
            "apds_dev_akaze_extract": (i, [vp, i, i, i, sz, i, vp, vp, i, ip, vp]),
            "apds_dev_points_from_matches": (i, [vp, i, vp, i, vp, i, i, vp, vp, vp]),
            "apds_dev_find_homography": (i, [vp, vp, i, i, d, i, d, vp, vp, vp]),
            "apds_dev_valu_popcount_peak": (i, [C.POINTER(d)]),
            "apds_dev_valu_peak": (i, [i, i, C.POINTER(d), C.POINTER(d), C.POINTER(C.c_char_p)]),
            "apds_dev_valu_peak_modes": (i, []),
            "apds_dev_last_kernel_ms": (i, [C.c_char_p, C.POINTER(f), ip]),
            "apds_dev_timing_enable": (i, [i]),
            "apds_akaze_debug_plane": (i, [vp, i, i, i, sz, i, i, vp]),
            "apds_stream_create": (i, [i, vp, i, pp]),
            "apds_stream_destroy": (i, [vp]),
            "apds_band_merger": (i, [vp, vp, vp, sz, vp, i, vp]),
            "apds_dev_band_merger": (i, [vp, vp, vp, sz, vp, i, vp, vp]),
            "apds_warp_perspective": (i, [vp, i, i, i, vp, i, i, vp]),
            "apds_warp_perspective_f32": (i, [vp, i, i, i, vp, i, i, vp]),
            "apds_pnp_solver_ransac": (i, [vp, vp, i, vp, i, f, d, i, vp, vp, vp, ip, ip]),
            "apds_pnp_hypotheses": (i, [vp, vp, i, vp, vp, i, i, vp]),
            "apds_pnp_sqpnp": (i, [vp, vp, i, vp, vp, vp, vp]),
            "apds_pnp_ippe": (i, [vp, vp, i, vp, vp, vp, vp]),
            "apds_get_world_coordinates": (i, [vp, i, vp, vp, vp, i, i, vp]),
            "apds_l2_knn_match": (i, [vp, i, vp, i, i, i, vp, vp]),
            "apds_dev_l2_topk": (i, [vp, i, vp, i64, i, u32, i, vp, vp]),
            "apds_dev_l2_topk_ex": (i, [vp, i, vp, i64, i, u32, i, i, vp, vp, ip, C.POINTER(d)]),
            "apds_db_create": (i, [pp, i64]),
            "apds_db_destroy": (i, [vp]),
            "apds_db_rows": (i64, [vp]),
            "apds_db_insert_image": (i, [vp, vp, vp, i, i, i, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
            "apds_db_select": (i, [vp, i, i, f, f, f, f, ip]),
            "apds_db_view": (i, [vp, pp, pp, pp, pp, ip]),
            "apds_db_view_download": (i, [vp, vp, vp, vp, vp]),
            "apds_db_knn_match": (i, [vp, vp, i, i, i, vp, vp]),
            "apds_comm_id_create": (i, [i, C.POINTER(CommId)]),
            "apds_shard_create": (i, [pp, i, i, i, C.POINTER(CommId), vp, vp, i64, u32]),
            "apds_shard_destroy": (i, [vp]),
            "apds_shard_info": (i, [vp, ip, ip, C.POINTER(i64), C.POINTER(u32), C.POINTER(C.c_char_p), ip]),
            "apds_shard_counts": (i, [vp, i, ip, vp]),
            "apds_shard_knn": (i, [vp, vp, i, ip, i, vp, vp]),
            "apds_shard_knn_replicated": (i, [vp, vp, i, i, i, vp, vp]),
            "apds_shard_slot_create": (i, [vp, i, i, pp]),
            "apds_shard_slot_destroy": (i, [vp, vp]),
            "apds_shard_gather": (i, [vp, vp, vp, i, ip, vp]),
            "apds_shard_scan": (i, [vp, vp, i, vp]),
            "apds_shard_exchange_merge": (i, [vp, vp, i, vp, vp]),
            "apds_db_shard": (i, [vp, i, i, i, C.POINTER(CommId), C.POINTER(HostTransport), pp]),
            "apds_dev_alloc": (i, [sz, pp]),
            "apds_dev_release": (i, [vp]),
            "apds_dev_upload": (i, [vp, vp, sz, vp]),
            "apds_dev_download": (i, [vp, vp, sz, vp]),
            "apds_stream_synchronize": (i, [vp]),
            "apds_pipeline_create": (i, [pp, vp, i64, u32, vp, vp, i64, C.POINTER(PipelineParams)]),
            "apds_pipeline_submit": (i, [vp, vp, sz, i, C.POINTER(i64)]),
            "apds_pipeline_poll": (i, [vp, C.POINTER(FrameResult), i]),
            "apds_pipeline_stats": (i, [vp, C.POINTER(PipelineCounters), i]),
            "apds_pipeline_destroy": (i, [vp]),
            "apds_dev_topk_state_create": (i, [pp]),
            "apds_dev_topk_state_destroy": (i, [vp]),
            "apds_dev_topk_prepass": (i, [vp, vp, i, vp, i64, u32, i, vp]),
            "apds_dev_topk_scan": (i, [vp, vp, vp, vp]),
            "apds_dev_topk_merge": (i, [vp, u32, vp, vp]),
        }
        for name, (rt, at) in sig.items():
            fn = getattr(L, name)
            fn.restype = rt
            fn.argtypes = at
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise ApdsError(rc, lib().apds_last_error().decode("utf-8", "replace"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def take(pointer, count, dtype, shape=None):
    """Copy `count` items from a library-allocated host buffer into numpy and apds_free() it."""
    out = np.zeros(count if shape is None else shape, dtype)
    if pointer.value:
        if out.nbytes:
            C.memmove(ptr(out), pointer, out.nbytes)
        lib().apds_free(pointer)
    return out


def kernel_ms(name):
    ms, n = C.c_float(0), C.c_int(0)
    check(lib().apds_dev_last_kernel_ms(name.encode(), C.byref(ms), C.byref(n)))
    return ms.value, n.value
