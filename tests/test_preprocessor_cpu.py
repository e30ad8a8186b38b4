"""CPU: the host logic of the preprocessor mirror that needs no GPU: TIFF round trip, tile grid arithmetic
(preprocessor/src/main.rs:212-216), image-table rows (main.rs:280-293), nearest-neighbour window indexing of MosaicedDataset."""
import numpy as np
import pytest


def test_tiff_round_trip(pkg, tmp_path):
    ge = pkg.geotiff_extractor
    rng = np.random.default_rng(3)
    for shape in ((3, 17, 29), (4, 8, 8), (1, 5, 3)):
        b = rng.normal(size=shape).astype(np.float32)
        b[0, 1, 2] = np.nan
        p = tmp_path / f"t{shape[0]}.tif"
        ge.write_tiff_f32(str(p), b)
        r = ge.read_tiff_f32(str(p))
        assert r.shape == shape and np.array_equal(r, b, equal_nan=True)
    (tmp_path / "bad.tif").write_bytes(b"MM\x00*" + b"\0" * 16)
    with pytest.raises(ValueError):
        ge.read_tiff_f32(str(tmp_path / "bad.tif"))


def test_tile_grid_and_image_rows(pkg):
    pp = pkg.preprocessor
    # main.rs:212-216 with image 4096 x 4096 and 3 levels: 1024-pixel tiles; 4x4, 2x2, 1x1 tiles
    assert pp.tile_grid((4096, 4096), 3, 0) == ((1024, 1024), 4, 4)
    assert pp.tile_grid((4096, 4096), 3, 1) == ((1024, 1024), 2, 2)
    assert pp.tile_grid((4096, 4096), 3, 2) == ((1024, 1024), 1, 1)
    assert pp.tile_grid((1000, 600), 2, 0) == ((500, 300), 2, 2)      # integer division as in the reference
    t = pp.ImageTable()
    assert t.create_image(1, 2048, 4095, 0, 2047) == 1 and t.create_image(0, 0, 1023, 0, 1023) == 2
    assert t.rows[0] == dict(id=1, level_of_detail=1, x_start=2048, x_end=4095, y_start=0, y_end=2047)


def test_window_indexing(pkg):
    ge = pkg.geotiff_extractor
    bands = np.stack([np.arange(64, dtype=np.float32).reshape(8, 8) + 100 * c for c in range(3)])
    ds = ge.MosaicedDataset(bands)
    assert ds.raster_size() == (8, 8)
    mm = ds.datasets_min_max()
    assert (mm.red_min, mm.red_max, mm.blue_min, mm.blue_max) == (0.0, 63.0, 200.0, 263.0)
    with pytest.raises(pkg.ApdsError):
        ds.to_rgb((4, 4), (8, 8), (4, 4))
