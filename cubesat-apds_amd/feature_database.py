"""Host-side mirror of the keypoint part of the reference crate `feature_database`
(/root/reference/feature_database/src/keypointdb.rs, models.rs) on a GPU-resident table instead of Postgres.
The table is what fills the descriptor database the matcher scans; a selection is directly usable as a train set."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DMATCH_DTYPE, KEYPOINT_DTYPE, check, lib, ptr

OPENCV_KEYPOINT_LIMIT = 2 ** 18 - 1     # keypointdb.rs:12


class KeypointRows:
    """What `Vec<models::Keypoint>` holds (models.rs:27-41), column-wise."""

    def __init__(self, ids, keypoints, descriptors, image_ids):
        self.ids, self.keypoints, self.descriptors, self.image_ids = ids, keypoints, descriptors, image_ids

    def __len__(self):
        return len(self.ids)


class KeypointTable:
    def __init__(self, capacity):
        self._h = C.c_void_p()
        check(lib().apds_db_create(C.byref(self._h), int(capacity)))

    def close(self):
        if self._h:
            lib().apds_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib().apds_db_rows(self._h))

    def create_keypoints(self, extracted, image_id, level_of_detail=0, column=0, row=0, tile_size=(0, 0)):
        """preprocessor/src/main.rs:296-324: insert every keypoint of one tile (`keypoints.to_db_type(image_id)` with x, y
        lifted to level-of-detail-0 pixels) — one INSERT ... VALUES in the reference (keypointdb.rs:100-109)."""
        kp = np.ascontiguousarray(extracted.keypoints, KEYPOINT_DTYPE)
        d = np.ascontiguousarray(extracted.descriptors, np.uint8)
        check(lib().apds_db_insert_image(self._h, ptr(kp), ptr(d), len(kp), int(image_id), int(level_of_detail), int(column), int(row),
                                         int(tile_size[0]), int(tile_size[1])))

    def _select(self, mode, value, box=(0, 0, 0, 0)):
        n = C.c_int(0)
        check(lib().apds_db_select(self._h, mode, int(value), float(box[0]), float(box[1]), float(box[2]), float(box[3]), C.byref(n)))
        m = n.value
        kp = np.zeros(m, KEYPOINT_DTYPE)
        d = np.zeros((m, 61), np.uint8)
        ids = np.zeros(m, np.int32)
        img = np.zeros(m, np.int32)
        if m:
            check(lib().apds_db_view_download(self._h, ptr(kp), ptr(d), ptr(ids), ptr(img)))
        return KeypointRows(ids, kp, d, img)

    def read_keypoints_from_image_id(self, image_id):
        """keypointdb.rs:38-48"""
        return self._select(0, image_id)

    def read_keypoints_from_lod(self, level_of_detail):
        """keypointdb.rs:50-65"""
        return self._select(1, level_of_detail)

    def read_keypoints_from_coordinates(self, x_start, y_start, x_end, y_end, level_of_detail):
        """keypointdb.rs:67-90"""
        return self._select(2, level_of_detail, (x_start, y_start, x_end, y_end))

    def view_device_pointers(self):
        """(rows64, keypoints, row_ids, image_ids, n): device pointers of the last selection (a ready train set)."""
        r, k, i, g, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        check(lib().apds_db_view(self._h, C.byref(r), C.byref(k), C.byref(i), C.byref(g), C.byref(n)))
        return r.value, k.value, i.value, g.value, n.value

    def knn_match_view(self, query_desc, k=2):
        """BFMatcher.knnMatch of host query descriptors against the last selection, which stays resident on the device."""
        q = np.ascontiguousarray(query_desc, np.uint8)
        idx = np.zeros((q.shape[0], k), np.int32)
        dist = np.zeros((q.shape[0], k), np.int32)
        check(lib().apds_db_knn_match(self._h, ptr(q), q.shape[0], q.shape[1], int(k), ptr(idx), ptr(dist)))
        return idx, dist


class ElevationTable:
    """The `geotransform` and `elevation` tables of feature_database/src/elevationdb.rs (create_geotransform :20-45,
    add_elevation_data :196-232, get_world_coordinates :64-104) held in memory; the conversion runs on the GPU."""

    def __init__(self):
        self.transforms = {}
        self.elevation = None

    def create_geotransform(self, dataset_name, transform):
        """elevationdb.rs:20-45 — `transform`: GDAL's six coefficients."""
        t = np.ascontiguousarray(transform, np.float64)
        if t.shape != (6,):
            raise ValueError("a geotransform has six coefficients")
        self.transforms[dataset_name] = t

    def add_elevation_data(self, raster):
        """elevationdb.rs:196-232 — heights row by row (ids are 1-based row-major positions)."""
        self.elevation = np.ascontiguousarray(raster, np.float64)
        if self.elevation.ndim != 2:
            raise ValueError("elevation raster: H x W")

    def get_world_coordinates_batch(self, xy):
        """elevationdb.rs:64-104 for n pixel positions at once -> n x 3 ECEF metres. Raises ApdsError(ERR_OUT_OF_RANGE) where the
        reference returns Err (a lookup outside the elevation table)."""
        xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
        out = np.zeros((len(xy), 3), np.float64)
        dgt = self.transforms["dataset"]
        egt = self.transforms.get("elevation") if self.elevation is not None else None
        if egt is None:
            check(lib().apds_get_world_coordinates(ptr(xy), len(xy), ptr(dgt), None, None, 0, 0, ptr(out)))
        else:
            check(lib().apds_get_world_coordinates(ptr(xy), len(xy), ptr(dgt), ptr(egt), ptr(self.elevation), self.elevation.shape[1],
                                                   self.elevation.shape[0], ptr(out)))
        return out

    def get_world_coordinates(self, x, y):
        """elevationdb.rs:64 — one point, (x, y, z) as the reference returns it."""
        return tuple(self.get_world_coordinates_batch([[x, y]])[0])
