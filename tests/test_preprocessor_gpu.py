"""GPU: the tile scheduler mirror (/root/reference/preprocessor/src/main.rs:175-327) end to end on a synthetic
three-band mosaic: every tile goes band_merger -> raster_to_mat -> AKAZE -> resident keypoint table. Each tile's rows must be
exactly what the oracle chain (band_merger, raster_to_mat, akaze) gives for that window, with the reference's coordinate lift.
(Level-of-detail > 0 windows are decimated by nearest neighbour on both sides: GDAL's Lanczos is not restated.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mosaic(pkg, size):
    t = pkg.synth.make_tile(size, size, frame_index=11, channels=3).astype(np.float32)
    bands = np.stack([t[:, :, 2] * 3.0 + 10.0, t[:, :, 1] * 2.0 - 5.0, t[:, :, 0] * 1.5])     # R, G, B with different ranges
    bands[0, 5:9, 7:12] = np.nan                                                            # a nodata patch in one band
    return bands


def test_process_lod_from_mosaic(gpu_pkg, oracle_mod, tmp_path):
    ge, pp, fd = gpu_pkg.geotiff_extractor, gpu_pkg.preprocessor, gpu_pkg.feature_database
    bands = _mosaic(gpu_pkg, 1024)
    path = tmp_path / "mosaic.tif"
    ge.write_tiff_f32(str(path), bands)
    ds = ge.MosaicedDataset.import_mosaic_dataset(str(path))
    assert np.array_equal(ds.bands, bands, equal_nan=True)
    table, images = fd.KeypointTable(200000), pp.ImageTable()
    out = pp.process_lod_from_mosaic(table, images, ds, 2)          # 2 levels: 4 tiles of 512 at lod 0, 1 tile at lod 1
    assert [len(level) for level in out] == [4, 1]
    assert [r["level_of_detail"] for r in images.rows] == [0, 0, 0, 0, 1]
    assert images.rows[1] == dict(id=2, level_of_detail=0, x_start=512, x_end=1023, y_start=0, y_end=511)
    assert images.rows[4] == dict(id=5, level_of_detail=1, x_start=0, x_end=1023, y_start=0, y_end=1023)
    assert len(table) == sum(n for level in out for _, n in level)
    mm = ds.datasets_min_max().as_array()
    oracle_mod.set_threads(8)
    for row in images.rows:
        lod, x0, y0 = row["level_of_detail"], row["x_start"], row["y_start"]
        span = 512 * 2 ** lod
        win = bands[:, y0:y0 + span, x0:x0 + span]
        if lod:
            idx = np.minimum(((np.arange(512) + 0.5) * (span / 512)).astype(np.int64), span - 1)
            win = win[:, idx][:, :, idx]
        rgba = oracle_mod.band_merger(win[0].ravel(), win[1].ravel(), win[2].ravel(), mm)
        bgra = oracle_mod.raster_to_mat(rgba, 512, 512)
        ref = oracle_mod.akaze(bgra)
        got = table.read_keypoints_from_image_id(row["id"])
        assert len(got) == len(ref.keypoints) > 100
        order = np.lexsort((np.arange(len(ref.keypoints)), -ref.keypoints["response"].astype(np.float64)))
        want = ref.keypoints[order].copy()
        want["x"] = want["x"] * np.float32(2.0 ** lod) + np.float32(x0)      # main.rs:299-300
        want["y"] = want["y"] * np.float32(2.0 ** lod) + np.float32(y0)
        assert np.array_equal(got.keypoints, want) and np.array_equal(got.descriptors, ref.descriptors[order])
        assert (got.image_ids == row["id"]).all()
    lod0 = table.read_keypoints_from_lod(0)
    assert len(lod0) == sum(n for _, n in out[0]) and lod0.keypoints["x"].max() > 512
    table.close()


def test_concurrent_tile_workers_give_the_same_tables(gpu_pkg):
    """The reference extracts tiles on a rayon pool (main.rs:233-243): with `workers` threads calling the C ABI concurrently the image
    ids, row order and every stored value must equal the serial run's."""
    ge, pp, fd = gpu_pkg.geotiff_extractor, gpu_pkg.preprocessor, gpu_pkg.feature_database
    ds = ge.MosaicedDataset(_mosaic(gpu_pkg, 2048))
    runs = []
    # fused: band_merger -> BGRA -> AKAZE inside one library call; batch: that call for several tiles at once (apds_tile_extract_batch)
    for workers, fused, batch in ((1, False, 1), (4, True, 1), (1, True, 1), (1, True, 5), (1, True, 16)):
        table, images = fd.KeypointTable(400000), pp.ImageTable()
        out = pp.process_lod_from_mosaic(table, images, ds, 3, workers=workers, fused=fused, batch=batch)     # 16 + 4 + 1 tiles of 512
        assert [len(level) for level in out] == [16, 4, 1]
        allk = table.read_keypoints_from_lod(0), table.read_keypoints_from_lod(1), table.read_keypoints_from_lod(2)
        runs.append((out, images.rows, [(k.keypoints.copy(), k.descriptors.copy(), k.image_ids.copy()) for k in allk]))
        table.close()
    for other in runs[1:]:
        assert runs[0][0] == other[0] and runs[0][1] == other[1]
        for a, b in zip(runs[0][2], other[2]):
            assert len(a[0]) > 100
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
