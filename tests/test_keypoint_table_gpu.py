"""GPU: the resident keypoint table — insert rescale (preprocessor/src/main.rs:296-304) and the three read
filters of feature_database/src/keypointdb.rs:38-90 (ORDER BY response DESC LIMIT 262143), against a numpy restatement
(the reference's own tests for these need a live Postgres). Exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Ex:
    def __init__(self, kp, d):
        self.keypoints, self.descriptors = kp, d


def _fake(pkg, n, seed):
    rng = np.random.default_rng(seed)
    kp = np.zeros(n, pkg._lib.KEYPOINT_DTYPE)
    kp["x"], kp["y"] = rng.random(n).astype(np.float32) * 512, rng.random(n).astype(np.float32) * 512
    kp["size"], kp["angle"] = 4.8, rng.random(n).astype(np.float32) * 360
    kp["response"] = (rng.random(n) * 0.05 + 0.001).astype(np.float32)
    kp["response"][:: max(n // 50, 1)] = np.float32(0.0123)          # ties in response
    kp["octave"], kp["class_id"] = rng.integers(0, 4, n), rng.integers(0, 16, n)
    d = pkg.synth.make_descriptor_db(n, seed=seed)
    return _Ex(kp, d)


def _want(rows, mask):
    ids = np.nonzero(mask)[0]
    order = np.lexsort((ids, -rows["response"][ids].astype(np.float64)))[:2 ** 18 - 1]
    return ids[order]


def test_insert_rescale_and_filters(gpu_pkg):
    fd = gpu_pkg.feature_database
    t = fd.KeypointTable(40000)
    all_kp, all_d, all_img, all_lod = [], [], [], []
    tile = (512, 512)
    for image_id, (lod, col, row, n) in enumerate([(0, 0, 0, 3000), (0, 1, 0, 2500), (1, 0, 0, 4000), (2, 3, 5, 1500), (1, 1, 2, 1)], start=1):
        ex = _fake(gpu_pkg, n, 100 + image_id)
        t.create_keypoints(ex, image_id, lod, col, row, tile)
        kp = ex.keypoints.copy()
        # preprocessor/src/main.rs:299-300 in f32
        kp["x"] = kp["x"] * np.float32(2.0 ** lod) + np.float32(col * tile[0] * 2 ** lod)
        kp["y"] = kp["y"] * np.float32(2.0 ** lod) + np.float32(row * tile[1] * 2 ** lod)
        all_kp.append(kp); all_d.append(ex.descriptors); all_img.append(np.full(n, image_id)); all_lod.append(np.full(n, lod))
    kp, d, img, lod = np.concatenate(all_kp), np.concatenate(all_d), np.concatenate(all_img), np.concatenate(all_lod)
    assert len(t) == len(kp)

    def check(rows, mask):
        want = _want(kp, mask)
        assert np.array_equal(rows.ids - 1, want)
        assert np.array_equal(rows.keypoints, kp[want]) and np.array_equal(rows.descriptors, d[want]) and np.array_equal(rows.image_ids, img[want])
        assert (np.diff(rows.keypoints["response"]) <= 0).all()

    check(t.read_keypoints_from_image_id(3), img == 3)
    check(t.read_keypoints_from_image_id(5), img == 5)
    assert len(t.read_keypoints_from_image_id(99)) == 0
    check(t.read_keypoints_from_lod(0), lod == 0)
    check(t.read_keypoints_from_lod(1), lod == 1)
    x0, y0, x1, y1 = 100.3, 50.7, 700.2, 400.9
    box = (lod == 0) & (kp["x"] >= np.floor(x0)) & (kp["x"] <= np.ceil(x1)) & (kp["y"] >= np.floor(y0)) & (kp["y"] <= np.ceil(y1))
    assert box.sum() > 100
    check(t.read_keypoints_from_coordinates(x0, y0, x1, y1, 0), box)
    t.close()


def test_limit_262143_keeps_strongest(gpu_pkg):
    fd = gpu_pkg.feature_database
    n = 300_000
    t = fd.KeypointTable(n)
    ex = _fake(gpu_pkg, n, 7)
    t.create_keypoints(ex, 1, 0)
    rows = t.read_keypoints_from_lod(0)
    assert len(rows) == fd.OPENCV_KEYPOINT_LIMIT
    want = _want(ex.keypoints, np.ones(n, bool))
    assert np.array_equal(rows.ids - 1, want)
    assert np.array_equal(rows.descriptors, ex.descriptors[want])
    t.close()


def test_selection_is_a_train_set(gpu_pkg, oracle_mod):
    """extract -> insert -> select by LOD -> match against the selection: train_idx indexes the returned rows."""
    fe, fd = gpu_pkg.feature_extraction, gpu_pkg.feature_database
    tile = gpu_pkg.synth.make_tile(512, 512, frame_index=31)
    ex = fe.akaze_keypoint_descriptor_extraction_def(tile, None)
    t = fd.KeypointTable(10000)
    t.create_keypoints(ex, 1, 0)
    t.create_keypoints(_fake(gpu_pkg, 3000, 9), 2, 1)
    rows = t.read_keypoints_from_lod(0)
    assert len(rows) == len(ex.keypoints)
    idx, dist = t.knn_match_view(ex.descriptors, 2)
    oi, od = oracle_mod.knn_hamming(ex.descriptors, rows.descriptors, 2)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert (dist[:, 0] == 0).all()                       # every descriptor finds itself in the table
    assert np.array_equal(rows.ids[idx[:, 0]] - 1, np.arange(len(ex.keypoints)))
    rn, kn, _, _, n = t.view_device_pointers()
    assert n == len(rows) and rn and kn
    t.close()
