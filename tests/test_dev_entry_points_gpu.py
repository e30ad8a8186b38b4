"""The device-resident entry points that no other test calls directly (include/apds.h "device-resident API", "measurement helpers"):
each against the oracle, or against the host-pointer entry that the other tests pin to the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(gpu_pkg):
    import torch
    return torch, torch.device("cuda:0"), gpu_pkg._lib.lib(), gpu_pkg._lib.check


def test_pack_descriptors_strided_rows(gpu_pkg, dev):
    torch, d, L, check = dev
    rng = np.random.default_rng(3)
    for n, nbytes, stride in ((1000, 61, 61), (257, 61, 80), (64, 32, 32), (5, 64, 64)):
        src = rng.integers(0, 256, (n, stride), dtype=np.uint8)
        t = torch.from_numpy(src).to(d)
        out = torch.full((n, 64), 0xAB, dtype=torch.uint8, device=d)
        check(L.apds_dev_pack_descriptors(t.data_ptr(), n, nbytes, stride, out.data_ptr(), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :nbytes], src[:, :nbytes]) and not got[:, nbytes:].any()


def test_cross_check_on_device_equals_host_entry(gpu_pkg, oracle_mod, dev):
    """lib.rs:116-126 on resident rows: nearest query of every train row (top-1, roles swapped) -> apds_dev_cross_check."""
    torch, d, L, check = dev
    db = gpu_pkg.synth.make_descriptor_db(4000, seed=21)
    q, _ = gpu_pkg.synth.make_queries(db, 1500, seed=22)
    q[7] = q[3]                                   # equal queries: the lower query index wins a train row
    ref = oracle_mod.get_bruteforce_matches(q, db)
    host = gpu_pkg.feature_extraction.get_bruteforce_matches(q, db)
    assert np.array_equal(host, ref)
    pad = lambda a: np.ascontiguousarray(np.pad(a, ((0, 0), (0, 64 - a.shape[1]))))   # noqa: E731
    tq, tt = torch.from_numpy(pad(q)).to(d), torch.from_numpy(pad(db)).to(d)
    tbest = torch.empty((len(db),), dtype=torch.int64, device=d)
    check(L.apds_dev_hamming_topk(tt.data_ptr(), len(db), tq.data_ptr(), len(q), 0, 1, tbest.data_ptr(), None))
    out = torch.empty((len(q), 4), dtype=torch.int32, device=d)
    n = C.c_int(0)
    check(L.apds_dev_cross_check(tbest.data_ptr(), len(db), len(q), out.data_ptr(), C.byref(n), None))
    got = out.cpu().numpy()[: n.value].copy().view(gpu_pkg._lib.DMATCH_DTYPE).ravel()
    assert n.value == len(ref) and np.array_equal(got, ref)


def test_band_merger_on_device_equals_host_entry(gpu_pkg, dev):
    torch, d, L, check = dev
    rng = np.random.default_rng(5)
    n = 100003
    bands = [rng.uniform(-0.2, 1.3, n).astype(np.float32) for _ in range(3)]
    bands[0][::17] = np.nan
    bands[1][::17] = np.nan
    bands[2][::34] = np.nan                      # some pixels lose all three bands (alpha 0), some only two
    mm = np.array([0.0, 1.0, 0.1, 0.9, -0.1, 1.1], np.float64)
    for bgra in (0, 1):
        host = np.zeros((n, 4), np.uint8)
        check(L.apds_band_merger(bands[0].ctypes.data, bands[1].ctypes.data, bands[2].ctypes.data, n, mm.ctypes.data, bgra, host.ctypes.data))
        tb = [torch.from_numpy(b).to(d) for b in bands]
        out = torch.zeros((n, 4), dtype=torch.uint8, device=d)
        check(L.apds_dev_band_merger(tb[0].data_ptr(), tb[1].data_ptr(), tb[2].data_ptr(), n, mm.ctypes.data, bgra, out.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), host)
        assert (host[:, 3] == 0).any() and (host[:, 3] == 255).any()


def test_batched_device_extraction_equals_single_calls(gpu_pkg, dev):
    torch, d, L, check = dev
    B, T, cap = 5, 384, 4096
    imgs = np.stack([gpu_pkg.synth.make_tile(T, T, frame_index=40 + i, channels=4) for i in range(B)])
    t = torch.from_numpy(imgs).to(d)
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=d)
    desc = torch.zeros((B, cap, 64), dtype=torch.uint8, device=d)
    counts = (C.c_int * B)()
    check(L.apds_dev_akaze_extract_batch(t.data_ptr(), B, T * T * 4, T, T, 4, T * 4, cap, kps.data_ptr(), desc.data_ptr(), cap, counts, None))
    torch.cuda.synchronize()
    k1 = torch.zeros((cap, 7), dtype=torch.float32, device=d)
    d1 = torch.zeros((cap, 64), dtype=torch.uint8, device=d)
    for i in range(B):
        n = C.c_int(0)
        check(L.apds_dev_akaze_extract(t[i].data_ptr(), T, T, 4, T * 4, cap, k1.data_ptr(), d1.data_ptr(), cap, C.byref(n), None))
        torch.cuda.synchronize()
        assert n.value == counts[i] and n.value > 50
        assert torch.equal(kps[i, : n.value].view(torch.int32), k1[: n.value].view(torch.int32)) and torch.equal(desc[i, : n.value], d1[: n.value])


def test_measurement_and_housekeeping_entries(gpu_pkg, dev):
    torch, d, L, check = dev
    assert b"gfx950" in L.apds_build_info()
    db = gpu_pkg.synth.make_descriptor_db(40000, seed=1)
    q, _ = gpu_pkg.synth.make_queries(db, 2000, seed=2)
    check(L.apds_dev_timing_enable(1))
    try:
        gpu_pkg.feature_extraction.knn_match(q, db, 2)
        ms, launches = C.c_float(0), C.c_int(0)
        check(L.apds_dev_last_kernel_ms(b"hamming_topk", C.byref(ms), C.byref(launches)))
        assert launches.value >= 1 and 0.0 < ms.value < 50.0
    finally:
        check(L.apds_dev_timing_enable(0))
    assert L.apds_live_contexts() >= 1
    peak = C.c_double(0)
    check(L.apds_dev_valu_popcount_peak(C.byref(peak)))
    assert 2e13 < peak.value < 1e14               # 256 CUs x 64 lanes x ~2.4 GHz x (a pair per ~3 issue cycles): tens of T lane-ops/s
    nm = L.apds_dev_valu_peak_modes()
    assert nm >= 4
    rate, cyc, name = C.c_double(0), C.c_double(0), C.c_char_p()
    check(L.apds_dev_valu_peak(0, 4, C.byref(rate), C.byref(cyc), C.byref(name)))
    assert rate.value > 1e13 and cyc.value > 0 and name.value
    # parked workspaces can be given back at any time; the next call allocates again and still answers correctly
    check(L.apds_thread_release())
    check(L.apds_release_cached_memory())
    idx, dist = gpu_pkg.feature_extraction.knn_match(q[:10], db[:100], 2)
    assert idx.shape == (10, 2) and (dist[:, 0] <= dist[:, 1]).all()
