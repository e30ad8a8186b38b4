"""GPU: the "next" rows either side of the hot path through the C ABI vs the oracle and the
reference's own KATs. Byte-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_band_merger_reference_kats(gpu_pkg):
    ge = gpu_pkg.geotiff_extractor
    mm = ge.BandsMinMax(-1.0, 2.0, -1.0, 2.0, -1.0, 2.0)
    out = ge.band_merger([[0.0, 0.5, 1.0]] * 3, mm)            # mod.rs:626-646 merging_bands
    assert out.shape == (3, 4) and out[0, 0] == 155
    assert ge.f32_to_u8(0.2, 0.1, 0.3) == 186                   # mod.rs:547-555
    assert ge.f32_to_u8(float("nan"), 0.1, 0.3) is None         # mod.rs:557-566
    assert ge.f32_to_u8(0.5, 0.0, 1.0) == round(float(np.float32(0.7297401)) * 255)   # gamma_correct_input, mod.rs:517-525


def test_band_merger_equals_oracle(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(3)
    n = 300_000
    bands = [rng.normal(0.4, 0.3, n).astype(np.float32) for _ in range(3)]
    for b in bands:
        b[rng.random(n) < 0.01] = np.nan
    bands[0][:50] = np.nan
    bands[1][:50] = np.nan
    bands[2][:50] = np.nan
    mm = gpu_pkg.geotiff_extractor.BandsMinMax(0.0017, 0.93, -0.2, 1.1, 0.05, 0.8)
    got = gpu_pkg.geotiff_extractor.band_merger(bands, mm)
    want = oracle_mod.band_merger(*bands, mm.as_array())
    assert np.array_equal(got, want)
    assert (got[:50, 3] == 0).all() and (got[50:, 3] == 255).all()
    bgra = gpu_pkg.geotiff_extractor.band_merger(bands, mm, bgra=True)   # fused raster_to_mat
    assert np.array_equal(bgra, want[:, [2, 1, 0, 3]])
    # every u8 boundary: a dense sweep of the normalised value
    sweep = np.linspace(0, 1, 2_000_001, dtype=np.float32)
    one = gpu_pkg.geotiff_extractor.BandsMinMax(0, 1, 0, 1, 0, 1)
    assert np.array_equal(gpu_pkg.geotiff_extractor.band_merger([sweep] * 3, one), oracle_mod.band_merger(sweep, sweep, sweep, one.as_array()))


def test_warp_image_perspective(gpu_pkg, oracle_mod):
    hg = gpu_pkg.homographier
    n = 4                                                        # mod.rs:683-707 warp_image_empty
    px = np.array([[1, (i % n) + 1, (i // n) + 1, 1] for i in range(n * n)], np.uint8)
    image = hg.raster_to_mat(px, n, n)
    eye = hg.Cmat.from_2d_slice([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]], np.float64)
    warped = hg.warp_image_perspective(image, eye, None)
    assert np.array_equal(warped.mat, image.mat)
    rng = np.random.default_rng(5)
    big = hg.Cmat(rng.integers(0, 256, (301, 407, 4), dtype=np.uint8), np.uint8, 4)
    for M in (np.eye(3), [[1, 0, 7.5], [0, 1, -3.25], [0, 0, 1]], [[0.9, -0.2, 30], [0.25, 1.1, -12], [1e-4, -2e-4, 1.0]],
              [[1.7, 0.1, -80], [-0.3, 1.4, 55], [3e-4, 1e-4, 0.9]]):
        m = hg.Cmat(np.array(M, np.float64), np.float64)
        got = hg.warp_image_perspective(big, m, None).mat
        assert np.array_equal(got, oracle_mod.warp_perspective(big.mat, M))
    got = hg.warp_image_perspective(big, hg.Cmat(np.eye(3), np.float64), (128, 64)).mat
    assert got.shape == (64, 128, 4) and np.array_equal(got, big.mat[:64, :128])
    with pytest.raises(hg.MatError):
        hg.warp_image_perspective(big, hg.Cmat(np.zeros((3, 3)), np.float64), None)


def test_warp_image_perspective_is_generic_over_the_element_type(gpu_pkg, oracle_mod):
    """mod.rs:271-300 is `warp_image_perspective<T: DataType>`: u8 and f32 elements with 1, 3 or 4 channels. u8: byte-equal to the oracle's
    fixed-point restatement; f32: bit-equal to its float restatement (the same four products and three sums, no contraction)."""
    hg = gpu_pkg.homographier
    rng = np.random.default_rng(7)
    Ms = [np.array([[0.93, -0.21, 9.0], [0.18, 1.07, -6.5], [4e-4, -7e-4, 1.0]]), np.array([[1, 0, 3.25], [0, 1, -2.75], [0, 0, 1.0]]),
          np.array([[1.6, 0.1, -40.0], [-0.2, 1.5, 25.0], [1e-3, 2e-3, 1.0]])]
    for dt in (np.uint8, np.float32):
        for ch in (1, 3, 4):
            shape = (61, 83) if ch == 1 else (61, 83, ch)
            img = rng.integers(0, 256, shape).astype(dt) if dt == np.uint8 else rng.normal(0, 50, shape).astype(np.float32)
            c = hg.Cmat(img, dt, ch)
            # the reference's own KAT shape, for every type: the identity warp changes nothing (mod.rs:683-707)
            assert np.array_equal(hg.warp_image_perspective(c, hg.Cmat(np.eye(3), np.float64), None).mat, img), (dt, ch)
            for M in Ms:
                for size in (None, (97, 45)):
                    got = hg.warp_image_perspective(c, hg.Cmat(M, np.float64), size).mat
                    want = oracle_mod.warp_perspective(img, M, size)
                    assert got.dtype == dt and got.shape == want.shape
                    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dt, ch, size)
    with pytest.raises(hg.MatError):
        hg.warp_image_perspective(hg.Cmat(np.zeros((8, 8, 2), np.uint8), np.uint8, 2), hg.Cmat(np.eye(3), np.float64), None)
