"""The reference's only AKAZE / match known answers (count assertions of feature_extraction/src/lib.rs:251-316):

    keypoints_count   30.tif -> 9079 keypoints, 31.tif -> 9357              (lib.rs:273)
    knn_matches_count k = 2, ratio 0.3, 30 -> 31: 27 matches                (lib.rs:295)
    bf_matches_count  cross-check 1-NN, 30 -> 31: 3228 matches              (lib.rs:314)

The two GeoTIFFs (resources/test/Geotiff/30.tif, 31.tif) are git-ignored upstream and absent here, so these tests are
FIXTURE-GATED: point APDS_REFERENCE_TIFF_DIR at a directory holding 30.tif and 31.tif and they run (oracle on the CPU,
product path on the GPU); without it they skip. The day the files exist, this is what pins the oracle (and the kernels)
to the reference's OpenCV build. Images are read as cv::imread(IMREAD_COLOR) does (lib.rs:157-159): 8-bit, 3 channels, BGR.
"""
import os

import numpy as np
import pytest

DIR = os.environ.get("APDS_REFERENCE_TIFF_DIR", "")
have = bool(DIR) and all(os.path.exists(os.path.join(DIR, f)) for f in ("30.tif", "31.tif"))
needs_fixture = pytest.mark.skipif(not have, reason="set APDS_REFERENCE_TIFF_DIR to a directory with the reference's 30.tif and 31.tif")


def imread_color(path):
    """cv::imread(path, IMREAD_COLOR): 8-bit BGR, alpha dropped, 16-bit samples scaled by 1/256, grey replicated."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("I;16", "I;16L", "I;16B", "I"):
        a = (np.asarray(im).astype(np.uint32) >> 8).clip(0, 255).astype(np.uint8)
        rgb = np.dstack([a, a, a])
    elif im.mode == "F":
        a = np.asarray(im)
        a = np.clip(np.rint(a), 0, 255).astype(np.uint8)
        rgb = np.dstack([a, a, a])
    else:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


def test_loader_reads_8bit_tiff_as_bgr(tmp_path):
    from PIL import Image
    rgb = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    Image.fromarray(rgb).save(tmp_path / "x.tif")
    got = imread_color(str(tmp_path / "x.tif"))
    assert got.dtype == np.uint8 and got.shape == (5, 7, 3) and np.array_equal(got, rgb[..., ::-1])
    Image.fromarray(rgb[..., 0]).save(tmp_path / "g.tif")
    g = imread_color(str(tmp_path / "g.tif"))
    assert np.array_equal(g[..., 0], rgb[..., 0]) and np.array_equal(g[..., 1], g[..., 2])


@needs_fixture
def test_oracle_reproduces_the_reference_counts(oracle_mod):
    a, b = (oracle_mod.akaze(imread_color(os.path.join(DIR, f))) for f in ("30.tif", "31.tif"))
    assert (len(a.keypoints), len(b.keypoints)) == (9079, 9357)                                  # lib.rs:273
    assert len(oracle_mod.get_knn_matches(a.descriptors, b.descriptors, 2, 0.3)) == 27           # lib.rs:287-295
    assert len(oracle_mod.get_bruteforce_matches(a.descriptors, b.descriptors)) == 3228          # lib.rs:310-314


@needs_fixture
@pytest.mark.gpu
def test_gpu_reproduces_the_reference_counts(gpu_pkg):
    fe = gpu_pkg.feature_extraction
    a, b = (fe.akaze_keypoint_descriptor_extraction_def(imread_color(os.path.join(DIR, f)), None) for f in ("30.tif", "31.tif"))
    assert (len(a.keypoints), len(b.keypoints)) == (9079, 9357)
    assert len(fe.get_knn_matches(a.descriptors, b.descriptors, 2, 0.3)) == 27
    assert len(fe.get_bruteforce_matches(a.descriptors, b.descriptors)) == 3228
