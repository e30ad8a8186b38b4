"""GPU: AKAZE parity, HIP path (through the C ABI) vs the oracle on the same seeded tiles.
Bar: keypoint set/order, octave/class_id and descriptors bit-exact; float keypoint fields bit-exact too
(the kernels mirror the oracle's float contract), with the documented tolerance as the fallback bar."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PLANES = {"Lt": 0, "Lx": 2, "Ly": 3, "Ldet": 4}


def _plane(pkg, tile, level, which, shape, dtype=np.float32):
    out = np.zeros(shape, dtype)
    ch = 1 if tile.ndim == 2 else tile.shape[2]
    pkg._lib.check(pkg.lib().apds_akaze_debug_plane(pkg._lib.ptr(tile), tile.shape[0], tile.shape[1], ch, tile.strides[0], level, which,
                                                    pkg._lib.ptr(out)))
    return out


def _assert_same_extraction(got, ref):
    gk, rk = got.keypoints, ref.keypoints
    assert len(gk) == len(rk), (len(gk), len(rk))
    assert np.array_equal(gk["class_id"], rk["class_id"]) and np.array_equal(gk["octave"], rk["octave"])
    for f in ("x", "y", "size", "response", "angle"):
        # tolerance stated in SURVEY §8c (1e-4 px / 1e-3 deg) is the fallback bar; bitwise is expected
        assert np.array_equal(gk[f], rk[f]), (f, np.abs(gk[f] - rk[f]).max())
    assert np.array_equal(got.descriptors, ref.descriptors)


@pytest.mark.parametrize("h,w,ch,frame", [(256, 256, 4, 0), (200, 333, 3, 1), (512, 512, 4, 2), (96, 640, 1, 3), (1024, 1024, 4, 4)])
def test_extraction_equals_oracle(gpu_pkg, oracle_mod, h, w, ch, frame):
    tile = gpu_pkg.synth.make_tile(h, w, frame_index=frame, channels=ch)
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None)
    oracle_mod.set_threads(8)
    ref = oracle_mod.akaze(tile)
    assert len(ref.keypoints) > 0
    _assert_same_extraction(got, ref)


def test_planes_equal_oracle(gpu_pkg, oracle_mod):
    tile = gpu_pkg.synth.make_tile(320, 448, frame_index=7)
    ref = oracle_mod.akaze(tile, keep_planes=True)
    k = _plane(gpu_pkg, tile, 0, 8, (1,))
    assert k[0] == np.float32(ref.kcontrast)
    for level in range(len(ref.levels)):
        lv = ref.levels[level]
        for name, which in PLANES.items():
            got = _plane(gpu_pkg, tile, level, which, (lv["h"], lv["w"]))
            want = ref.plane(level, which)
            assert np.array_equal(got, want), (level, name, np.abs(got - want).max())
        m = _plane(gpu_pkg, tile, level, 7, (lv["h"], lv["w"]), np.uint8)
        assert np.array_equal(m != 0, ref.plane(level, oracle_mod.PLANE_MASK1) != 0), level


def test_odd_sizes_use_general_area_resize(gpu_pkg, oracle_mod):
    tile = gpu_pkg.synth.make_tile(301, 407, frame_index=9, channels=1)
    _assert_same_extraction(gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None), oracle_mod.akaze(tile))


def test_dense_texture_exercises_suppression_rounds(gpu_pkg, oracle_mod):
    # high-contrast noise gives thousands of neighbouring extrema: long dependency chains in the cross-level pass
    rng = np.random.default_rng(4)
    base = rng.integers(0, 256, (96, 96), dtype=np.uint8)
    tile = np.kron(base, np.ones((4, 4), np.uint8))
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None)
    ref = oracle_mod.akaze(tile)
    assert len(ref.keypoints) > 500
    _assert_same_extraction(got, ref)


def test_max_points_keeps_strongest(gpu_pkg, oracle_mod):
    tile = gpu_pkg.synth.make_tile(512, 512, frame_index=2)
    for mp in (1, 17, 100):
        got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, mp)
        ref = oracle_mod.akaze(tile, max_points=mp)
        assert len(got.keypoints) == mp
        _assert_same_extraction(got, ref)


def test_blank_and_tiny_images(gpu_pkg, oracle_mod):
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(np.full((128, 128, 3), 77, np.uint8), None)
    assert len(got.keypoints) == 0 and got.descriptors.shape == (0, 61)
    with pytest.raises(gpu_pkg.ApdsError) as e:     # AKAZE asserts img_width > 2 && img_height > 2
        gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(np.zeros((2, 2), np.uint8), None)
    assert e.value.code == -215
    small = gpu_pkg.synth.make_tile(64, 64, frame_index=5)
    _assert_same_extraction(gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(small, None), oracle_mod.akaze(small))


def test_to_db_type(gpu_pkg):
    tile = gpu_pkg.synth.make_tile(256, 256)
    ex = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None)
    rows = ex.to_db_type(42)                     # lib.rs:34-58
    assert len(rows) == len(ex.keypoints)
    r = rows[0]
    assert r.image_id == 42 and len(r.descriptor) == 61 and r.descriptor == bytes(ex.descriptors[0])
    assert r.x_coord == float(ex.keypoints["x"][0]) and r.class_id == int(ex.keypoints["class_id"][0])


def test_noise_image_many_keypoints_and_large_max_points(gpu_pkg, oracle_mod):
    """Pure noise: tens of thousands of keypoints per megapixel, dense cross-level interaction, and a max_points cut that
    goes through the rank-select kernel at scale (ties in response are broken by detection order)."""
    rng = np.random.default_rng(12)
    tile = rng.integers(0, 256, (768, 1024), dtype=np.uint8)
    oracle_mod.set_threads(8)
    ref = oracle_mod.akaze(tile)
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None)
    assert len(ref.keypoints) > 5000
    _assert_same_extraction(got, ref)
    cut = len(ref.keypoints) // 3
    _assert_same_extraction(gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, cut), oracle_mod.akaze(tile, max_points=cut))


def test_strided_rows_and_channel_orders(gpu_pkg, oracle_mod):
    """Mat rows may be padded (step > cols * channels); BGR weights differ per channel."""
    rng = np.random.default_rng(13)
    base = gpu_pkg.synth.make_tile(200, 264, frame_index=21, channels=3).astype(np.int32)
    base[..., 0] = np.clip(base[..., 0] + rng.integers(-40, 40, base.shape[:2]), 0, 255)
    base[..., 2] = np.clip(255 - base[..., 2], 0, 255)
    img = base.astype(np.uint8)
    padded = np.zeros((200, 264 + 9, 3), np.uint8)
    padded[:, :264] = img
    view = padded[:, :264]                       # non-contiguous rows: stride 273*3
    assert not view.flags["C_CONTIGUOUS"]
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(view, None)
    ref = oracle_mod.akaze(img)
    _assert_same_extraction(got, ref)
    bgra = np.dstack([img, np.full(img.shape[:2], 255, np.uint8)])
    _assert_same_extraction(gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(bgra, None), ref)


def test_full_size_4096_frame_equals_oracle(gpu_pkg, oracle_mod):
    # BASELINE config 2 size: one 4096 x 4096 BGRA tile, every keypoint field and descriptor bit-equal to the oracle
    # (the oracle needs a few seconds on the host cores)
    tile = gpu_pkg.synth.make_tile(4096, 4096, frame_index=0, channels=4)
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None)
    oracle_mod.set_threads(16)
    ref = oracle_mod.akaze(tile)
    assert len(ref.keypoints) > 20000
    _assert_same_extraction(got, ref)
    # extraction is a pure function of the image: a second run gives the same bytes
    again = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None)
    assert np.array_equal(again.descriptors, got.descriptors) and np.array_equal(again.keypoints, got.keypoints)


def test_2048_frame_equals_oracle(gpu_pkg, oracle_mod):
    # a size of the reference's bench sweep (benchmarks/benches/feature_extraction.rs:14-15) that no other test touches; 3 channels as there
    tile = gpu_pkg.synth.make_tile(2048, 2048, frame_index=11, channels=3)
    oracle_mod.set_threads(16)
    _assert_same_extraction(gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(tile, None), oracle_mod.akaze(tile))


@pytest.mark.parametrize("h,w", [(2891, 3333), (5003, 2049)])
def test_odd_large_frames_equal_oracle(gpu_pkg, oracle_mod, h, w):
    # Frames of 8 Mpx and more take the streaming Hessian / level kernels on their own (csrc/akaze_doh_strips.hip, akaze_level_stream.hip);
    # the square power-of-two sizes above never leave a partial 62-column strip, a partial row band or an odd half-size behind. Odd sizes do.
    t = gpu_pkg.synth.make_tile(4096, 4096, frame_index=7, channels=1)
    img = np.ascontiguousarray(np.block([[t], [t[::-1]]])[:h, :w]) if h > 4096 else np.ascontiguousarray(t[:h, :w])
    assert img.shape == (h, w) and h * w >= 1 << 23
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(img, None)
    oracle_mod.set_threads(16)
    ref = oracle_mod.akaze(img)
    assert len(ref.keypoints) > 10000
    _assert_same_extraction(got, ref)


def test_8192_frame_equals_oracle(gpu_pkg, oracle_mod):
    # the largest size of the reference's bench sweep: 8192 x 8192. The image is a 2 x 2 mosaic of one synthetic 4096^2 tile and its three
    # flips (the tile generator needs 15 s per 4096^2 tile). More than 100 k keypoints: also the largest keypoint set of the suite
    t = gpu_pkg.synth.make_tile(4096, 4096, frame_index=5, channels=1)
    img = np.ascontiguousarray(np.block([[t, t[:, ::-1]], [t[::-1], t[::-1, ::-1]]]))
    assert img.shape == (8192, 8192)
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(img, None)
    oracle_mod.set_threads(16)
    ref = oracle_mod.akaze(img)
    assert len(ref.keypoints) > 100000
    _assert_same_extraction(got, ref)


@pytest.mark.parametrize("size,ch,batch", [(256, 4, 5), (512, 3, 3), (1024, 4, 4), (200, 1, 7)])
def test_batched_extraction_equals_oracle_per_image(gpu_pkg, oracle_mod, size, ch, batch):
    # every image of a batch (one launch per kernel for all of them) against the ORACLE, not only against the unbatched call
    imgs = np.stack([gpu_pkg.synth.make_tile(size, size + (40 if ch == 1 else 0), frame_index=20 + i, channels=ch) for i in range(batch)])
    if batch > 2:
        imgs[1] = 128                      # a flat image in the middle of the batch: no keypoints, its neighbours unaffected
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_batch(imgs, None)
    assert len(got) == batch
    oracle_mod.set_threads(8)
    for i in range(batch):
        ref = oracle_mod.akaze(imgs[i])
        _assert_same_extraction(got[i], ref)
        one = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(imgs[i], None)
        assert np.array_equal(one.keypoints, got[i].keypoints) and np.array_equal(one.descriptors, got[i].descriptors)
    assert len(got[1].keypoints) == 0 if batch > 2 else True
    assert sum(len(g.keypoints) for g in got) > 0


def test_batched_extraction_max_points_path(gpu_pkg, oracle_mod):
    # an image of the batch has more keypoints than max_points: the rank-selection path runs image by image
    imgs = np.stack([gpu_pkg.synth.make_tile(384, 384, frame_index=40 + i, channels=4) for i in range(3)])
    oracle_mod.set_threads(8)
    full = [len(oracle_mod.akaze(im).keypoints) for im in imgs]
    cap = max(8, min(full) // 2)
    got = gpu_pkg.feature_extraction.akaze_keypoint_descriptor_extraction_batch(imgs, cap)
    for i in range(3):
        ref = oracle_mod.akaze(imgs[i], max_points=cap)
        assert len(ref.keypoints) == min(cap, full[i])
        _assert_same_extraction(got[i], ref)
