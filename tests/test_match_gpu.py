"""GPU: Hamming match parity, HIP path (through the C ABI) vs the oracle. Bit-exact: indices, distances, order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a, b):
    return np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)


@pytest.mark.parametrize("nq,nt,k", [(1, 2, 2), (3, 5, 2), (64, 64, 2), (65, 4097, 2), (1000, 10000, 2), (257, 33333, 1),
                                     (5000, 20000, 2), (20000, 3001, 2),
                                     (9000, 70000, 2), (20000, 50000, 2)])   # 2 / 4 queries per lane with a threshold pass and many chunks
def test_knn_match_equals_oracle(gpu_pkg, oracle_mod, nq, nt, k):
    db = gpu_pkg.synth.make_descriptor_db(nt, seed=0x44420001 + nt)
    q, _ = gpu_pkg.synth.make_queries(db, nq, seed=0x51550001 + nq)
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, k)
    oi, od = oracle_mod.knn_hamming(q, db, k)
    assert np.array_equal(dist, od)
    assert np.array_equal(idx, oi)


def _numpy_top2(q, db):
    """Brute force in numpy, independent of oracle/ and of the kernels: popcount by unpackbits, stable argsort = ties to the lower row."""
    d = _bits(q, db).astype(np.int32)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    return order.astype(np.int32), np.take_along_axis(d, order, axis=1)


def test_knn_match_against_a_numpy_popcount(gpu_pkg):
    """An anchor that shares no text with the oracle: distances are popcount(xor) by definition and BFMatcher's order is by distance with
    ties to the lower train index (a stable sort). Both matchers (the default runs here, the other one in
    test_strip_kernels_gpu.py::test_match_parity_on_the_vector_alu_matcher) must give exactly that, planted duplicates included."""
    rng = np.random.default_rng(21)
    db = rng.integers(0, 256, (3000, 61), dtype=np.uint8)
    db[:, 60] &= 0x3F
    q = rng.integers(0, 256, (700, 61), dtype=np.uint8)
    q[:, 60] &= 0x3F
    q[:200] = db[rng.integers(0, 3000, 200)]                 # exact hits
    db[1500:1600] = db[100:200]                              # duplicate rows: ties between rows 100.. and 1500..
    q[200:260] = db[100:160]
    q[260:300, 7] ^= 0x11                                    # near misses of random rows
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, 2)
    wi, wd = _numpy_top2(q, db)
    assert np.array_equal(dist, wd) and np.array_equal(idx, wi)
    i1, d1 = gpu_pkg.feature_extraction.knn_match(q, db, 1)
    assert np.array_equal(i1[:, 0], wi[:, 0]) and np.array_equal(d1[:, 0], wd[:, 0])


@pytest.mark.parametrize("nq,nt,k", [(1, 1, 2), (5, 2, 2), (777, 129, 1), (3000, 70000, 2), (40000, 300000, 2), (260, 65536 + 127, 1)])
def test_both_backends_give_the_same_keys(gpu_pkg, nq, nt, k):
    """apds_dev_hamming_topk_backend: the vector-ALU kernel (xor + popcount) and the matrix-core kernel (bits as FP4 operands) in ONE process on
    the same device buffers: every key (distance << 32 | row) identical, empty slots included (fewer train rows than k), with and without
    the matrix-core matcher's threshold launch (from 65 536 rows up), duplicate rows planted."""
    import ctypes as C
    import torch
    L, check = gpu_pkg.lib(), gpu_pkg._lib.check
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(nq * 7919 + nt)
    db = torch.randint(0, 256, (nt, 64), dtype=torch.uint8, device=dev, generator=g)
    q = torch.randint(0, 256, (nq, 64), dtype=torch.uint8, device=dev, generator=g)
    for t in (db, q):
        t[:, 60] &= 0x3F
        t[:, 61:] = 0
    m = min(nq, nt, 500)
    q[:m] = db[torch.randint(0, nt, (m,), device=dev, generator=g)]
    if nt >= 2000:
        db[nt // 2: nt // 2 + 300] = db[:300]
    outs = []
    for backend in (1, 2):
        out = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()   # the library call runs on the library's own stream: torch's writes above must have landed
        check(L.apds_dev_hamming_topk_backend(q.data_ptr(), nq, db.data_ptr(), nt, 1000, k, out.data_ptr(), backend, None))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert (outs[0] != -7).all()
    if nt < k:
        assert (outs[0][:, nt:] == -1).all()          # EMPTY_KEY = ~0
    # a named backend refuses what it cannot do
    big = torch.empty((nq, 3), dtype=torch.int64, device=dev)
    assert L.apds_dev_hamming_topk_backend(q.data_ptr(), nq, db.data_ptr(), nt, 0, 3, big.data_ptr(), 2, None) == -215


def test_ties_prefer_lower_train_index(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(5)
    # few distinct rows repeated many times: every query has massive ties, across chunk borders too
    base = rng.integers(0, 256, (7, 61), dtype=np.uint8)
    base[:, 60] &= 0x3F
    db = base[rng.integers(0, 7, 150000)]
    q = base[rng.integers(0, 7, 300)].copy()
    q[::3, 5] ^= 1
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, 2)
    oi, od = oracle_mod.knn_hamming(q, db, 2)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)


def test_large_db_uses_sample_thresholds(gpu_pkg, oracle_mod):
    # >= 131072 rows switches on the threshold pre-pass; planted neighbours sit inside and outside the sample
    db = gpu_pkg.synth.make_descriptor_db(200000)
    q, src = gpu_pkg.synth.make_queries(db, 3000)
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, 2)
    oracle_mod.set_threads(8)
    oi, od = oracle_mod.knn_hamming(q, db, 2)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert ((src >= 0) & (src < 16384)).any() and (src >= 16384).any()


def test_ties_across_the_threshold_launch(gpu_pkg, oracle_mod):
    """The matrix-core matcher ranks the leading rows first (a sixteenth of the set, at most 16 384) and starts the launch over the rest from
    each query's second-best value of that sample: a later row enters only with a strictly smaller value. Planted here, for every query:
    its best and second best inside the sample, and rows in the rest that tie with the best, tie with the second best (must lose to the
    lower index), beat both, or are exact copies."""
    rng = np.random.default_rng(11)
    nt, nq = 90000, 400                      # sample = 5504 rows
    db = rng.integers(0, 256, (nt, 61), dtype=np.uint8)
    db[:, 60] &= 0x3F
    q = rng.integers(0, 256, (nq, 61), dtype=np.uint8)
    q[:, 60] &= 0x3F

    def flipped(row, nbits, seed):
        r = row.copy()
        for b in np.random.default_rng(seed).choice(480, nbits, replace=False):
            r[b >> 3] ^= 1 << (b & 7)
        return r
    for i in range(nq):
        db[10 + 13 * i] = flipped(q[i], 5, 3 * i)            # best of the sample: distance 5
        db[12 + 13 * i] = flipped(q[i], 9, 3 * i + 1)        # second best of the sample: distance 9
        kind = i % 5
        r = 6000 + 200 * i                                   # a row of the rest
        if kind == 0:
            db[r] = flipped(q[i], 9, 3 * i + 2)              # ties with the sample's second best: the lower index wins
        elif kind == 1:
            db[r] = flipped(q[i], 5, 3 * i + 2)              # ties with the best: becomes the second neighbour
        elif kind == 2:
            db[r] = flipped(q[i], 2, 3 * i + 2)              # beats both
        elif kind == 3:
            db[r] = q[i]                                     # exact copies, twice
            db[r + 1] = q[i]
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, 2)
    oracle_mod.set_threads(8)
    oi, od = oracle_mod.knn_hamming(q, db, 2)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert (idx[0::5, 1] == 12 + 13 * np.arange(0, nq, 5)).all() and (idx[1::5, 1] == 6000 + 200 * np.arange(1, nq, 5)).all()
    i1 = gpu_pkg.feature_extraction.knn_match(q, db, 1)[0]
    assert np.array_equal(i1[:, 0], oi[:, 0])


def test_get_knn_matches(gpu_pkg, oracle_mod):
    db = gpu_pkg.synth.make_descriptor_db(10000)
    q, src = gpu_pkg.synth.make_queries(db, 3000)
    got = gpu_pkg.feature_extraction.get_knn_matches(q, db, 2, 0.3)
    want = oracle_mod.get_knn_matches(q, db, 2, 0.3)
    assert np.array_equal(got, want)
    assert len(got) == (src >= 0).sum()
    # k > 2 gives the same result: only the two nearest are consumed (lib.rs:107-111)
    assert np.array_equal(gpu_pkg.feature_extraction.get_knn_matches(q, db, 5, 0.3), want)
    for fs in (0.0, 0.5, 0.8, 1.0, 1.5):
        assert np.array_equal(gpu_pkg.feature_extraction.get_knn_matches(q, db, 2, fs), oracle_mod.get_knn_matches(q, db, 2, fs))


@pytest.mark.parametrize("nq,nt", [(50, 80), (1000, 1000), (3000, 10000), (9000, 2500)])
def test_get_bruteforce_matches(gpu_pkg, oracle_mod, nq, nt):
    db = gpu_pkg.synth.make_descriptor_db(nt)
    q, _ = gpu_pkg.synth.make_queries(db, nq, planted=0.5)
    got = gpu_pkg.feature_extraction.get_bruteforce_matches(q, db)
    want = oracle_mod.get_bruteforce_matches(q, db)
    assert np.array_equal(got, want)


def test_other_descriptor_lengths(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(3)
    for nb in (1, 4, 32, 61, 64):
        q = rng.integers(0, 256, (100, nb), dtype=np.uint8)
        t = rng.integers(0, 256, (999, nb), dtype=np.uint8)
        idx, dist = gpu_pkg.feature_extraction.knn_match(q, t, 2)
        oi, od = oracle_mod.knn_hamming(q, t, 2)
        assert np.array_equal(dist, od) and np.array_equal(idx, oi), nb


def test_points_from_matches(gpu_pkg, oracle_mod):
    rng = np.random.default_rng(11)
    K, M = gpu_pkg._lib.KEYPOINT_DTYPE, gpu_pkg._lib.DMATCH_DTYPE
    k1, k2 = np.zeros(500, K), np.zeros(700, K)
    for k in (k1, k2):
        k["x"], k["y"] = rng.random(len(k)) * 4096, rng.random(len(k)) * 4096
    m = np.zeros(300, M)
    m["query_idx"], m["train_idx"] = rng.integers(0, 500, 300), rng.integers(0, 700, 300)
    for bug in (False, True):
        g1, g2 = gpu_pkg.feature_extraction.get_points_from_matches(k1, k2, m, bug)
        o1, o2 = oracle_mod.get_points_from_matches(k1, k2, m, bug)
        assert np.array_equal(g1, o1) and np.array_equal(g2, o2)
    m["train_idx"][7] = 700
    with pytest.raises(gpu_pkg.ApdsError) as e:
        gpu_pkg.feature_extraction.get_points_from_matches(k1, k2, m)
    assert e.value.code == -211


def test_raster_to_mat(gpu_pkg, oracle_mod):
    # reference KAT mod.rs:556-603 plus a random image against the oracle
    n = 4
    px = np.array([[1, (i % n) + 1, (i // n) + 1, 1] for i in range(n * n)], np.uint8)
    m = gpu_pkg.homographier.raster_to_mat(px, n, n).mat
    assert tuple(m[0, 0]) == (1, 1, 1, 1) and tuple(m[3, 3]) == (4, 4, 1, 1)
    assert tuple(m[0, 3]) == (1, 4, 1, 1) and tuple(m[3, 0]) == (4, 1, 1, 1)
    rng = np.random.default_rng(2)
    big = rng.integers(0, 256, (333 * 517, 4), dtype=np.uint8)
    assert np.array_equal(gpu_pkg.homographier.raster_to_mat(big, 517, 333).mat, oracle_mod.raster_to_mat(big, 517, 333))


@pytest.mark.parametrize("k", [3, 4, 5, 8, 16])
def test_larger_k(gpu_pkg, oracle_mod, k):
    """k > 2 (the reference's `k: i32` is free; its own code only consumes two): one query per lane, K slots in registers."""
    db = gpu_pkg.synth.make_descriptor_db(150000 if k == 4 else 3000, seed=99 + k)
    db[1000] = db[10]
    db[2000] = db[10]          # ties must stay in train-index order at every rank
    q, _ = gpu_pkg.synth.make_queries(db, 700, seed=5 + k)
    q[0] = db[10]
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, k)
    oracle_mod.set_threads(8)
    oi, od = oracle_mod.knn_hamming(q, db, k)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert tuple(idx[0, :3]) == (10, 1000, 2000)
    few = db[:k - 1]           # fewer train rows than k: the tail is (-1, INT_MAX)
    idx, dist = gpu_pkg.feature_extraction.knn_match(q[:50], few, k)
    oi, od = oracle_mod.knn_hamming(q[:50], few, k)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi) and (idx[:, -1] == -1).all()


def test_full_size_properties_1m_rows(gpu_pkg, oracle_mod):
    """BASELINE size (1 M-row DB) through size-independent properties: (a) a query that is a DB row finds itself at distance 0, and a
    duplicated row resolves to the lower index; (b) planted neighbours are recovered; (c) sharding consistency: top-2 over the whole
    DB == u64-key merge of the top-2 over its two halves (what the multi-GPU path computes); (d) a bounded sample of the queries
    equals the oracle exactly."""
    n = 1_000_000
    db = gpu_pkg.synth.make_descriptor_db(n)
    db[700_001] = db[123]                                   # duplicate across the halves: index 123 must win
    q, src = gpu_pkg.synth.make_queries(db, 8192)
    q[0], q[1], q[2] = db[123], db[999_999], db[500_000]    # exact copies: first row region, last row, first row of the 2nd half
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, 2)
    assert tuple(idx[0]) == (123, 700_001) and tuple(dist[0]) == (0, 0)
    assert idx[1, 0] == 999_999 and dist[1, 0] == 0 and idx[2, 0] == 500_000 and dist[2, 0] == 0
    planted = np.nonzero(src >= 0)[0]
    planted = planted[planted > 2]
    assert len(planted) > 2000 and np.array_equal(idx[planted, 0], src[planted])
    assert (np.diff(dist.astype(np.int64), axis=1) >= 0).all()
    # (c) halves + merge on u64 keys (distance << 32 | global index)
    h = n // 2
    ia, da = gpu_pkg.feature_extraction.knn_match(q, db[:h], 2)
    ib, db_ = gpu_pkg.feature_extraction.knn_match(q, db[h:], 2)
    keys = np.concatenate([(da.astype(np.uint64) << np.uint64(32)) | ia.astype(np.uint64),
                           (db_.astype(np.uint64) << np.uint64(32)) | (ib.astype(np.uint64) + np.uint64(h))], axis=1)
    keys.sort(axis=1)
    assert np.array_equal((keys[:, :2] & np.uint64(0xFFFFFFFF)).astype(np.int64), idx.astype(np.int64))
    assert np.array_equal((keys[:, :2] >> np.uint64(32)).astype(np.int64), dist.astype(np.int64))
    # (d) exact against the oracle on a sample the CPU finishes in seconds
    oracle_mod.set_threads(8)
    sel = np.arange(0, len(q), 64)
    oi, od = oracle_mod.knn_hamming(q[sel], db, 2)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od)


def test_config5_db_size_10m_rows(gpu_pkg, oracle_mod):
    """BASELINE config 5's DB size (10 M rows, 640 MB resident) on one GPU: row indices pass 2^22 (the width of the in-chunk row
    offset of the packed partial keys), so the scan splits into more chunks. Properties: exact copies found at distance 0 anywhere in
    the index range (including the last row), duplicated rows resolve to the lower index, planted neighbours recovered, and a bounded
    sample of the queries equals the oracle exactly."""
    n = 10_000_000
    db = gpu_pkg.synth.make_descriptor_db(n)
    db[9_000_001] = db[4_194_303]                           # duplicate across the 2^22 boundary region: the lower index must win
    q, src = gpu_pkg.synth.make_queries(db, 1024)
    q[0], q[1], q[2], q[3] = db[4_194_303], db[n - 1], db[4_194_304], db[0]
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, 2)
    assert tuple(idx[0]) == (4_194_303, 9_000_001) and tuple(dist[0]) == (0, 0)
    assert idx[1, 0] == n - 1 and dist[1, 0] == 0 and idx[2, 0] == 4_194_304 and dist[2, 0] == 0 and idx[3, 0] == 0 and dist[3, 0] == 0
    planted = np.nonzero(src >= 0)[0]
    planted = planted[planted > 3]
    assert len(planted) > 200 and np.array_equal(idx[planted, 0], src[planted])
    assert (np.diff(dist.astype(np.int64), axis=1) >= 0).all()
    oracle_mod.set_threads(8)
    sel = np.arange(0, len(q), 32)
    oi, od = oracle_mod.knn_hamming(q[sel], db, 2)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od)


@pytest.mark.parametrize("k,nt", [(17, 3000), (32, 150000), (50, 40000), (100, 5000)])
def test_k_above_16_runs_in_pages(gpu_pkg, oracle_mod, k, nt):
    """BFMatcher::knnMatch takes any k (lib.rs:94-103): above 16 the scan runs in pages of 16, each page the keys above the last key of the
    page before. 40 identical train rows make one query's ties straddle the page borders (index order must hold across them); with 150 000
    rows every page also has its own threshold pre-pass."""
    db = gpu_pkg.synth.make_descriptor_db(nt, seed=7 + k)
    twins = np.linspace(10, nt - 1, 40).astype(np.int64)
    db[twins] = db[10]
    q, _ = gpu_pkg.synth.make_queries(db, 300, seed=k)
    q[0] = db[10]
    idx, dist = gpu_pkg.feature_extraction.knn_match(q, db, k)
    oracle_mod.set_threads(8)
    oi, od = oracle_mod.knn_hamming(q, db, k)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert np.array_equal(idx[0, :min(k, 40)], twins[:min(k, 40)]) and (dist[0, :min(k, 40)] == 0).all()
    few = db[:k - 5]           # fewer train rows than k: the tail is (-1, INT_MAX), also across a page border
    idx, dist = gpu_pkg.feature_extraction.knn_match(q[:50], few, k)
    oi, od = oracle_mod.knn_hamming(q[:50], few, k)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi) and (idx[:, -5:] == -1).all()


@pytest.mark.parametrize("nq,nt,k", [(3000, 200000, 2), (20000, 60000, 2), (500, 5000, 2), (9000, 70000, 1)])
def test_split_scan_equals_the_one_call_scan(gpu_pkg, nq, nt, k):
    # apds_dev_topk_prepass / _scan / _merge on a state object, each step on ITS OWN stream with events between them and two frames in
    # flight (the second frame's pre-pass is issued before the first frame's merge, as the streamed pipeline does), against
    # apds_dev_hamming_topk on the same rows: identical keys
    import ctypes as C
    import torch
    L, check = gpu_pkg.lib(), gpu_pkg._lib.check
    dev = torch.device("cuda:0")
    db = gpu_pkg.synth.make_descriptor_db(nt, seed=0x44420001 + nt)
    qa, _ = gpu_pkg.synth.make_queries(db, nq, seed=0x51550001 + nq)
    qb = np.ascontiguousarray(qa[::-1])
    pad = lambda a: torch.from_numpy(np.concatenate([a, np.zeros((len(a), 3), np.uint8)], 1)).to(dev)   # noqa: E731
    rows, da, dbq = pad(db), pad(qa), pad(qb)
    want = torch.empty((nq, k), dtype=torch.int64, device=dev)
    st_pre, st_main, st_merge = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    check(L.apds_dev_hamming_topk(da.data_ptr(), nq, rows.data_ptr(), nt, 7, k, want.data_ptr(), C.c_void_p(st_main.cuda_stream)))
    st_main.synchronize()
    states = [C.c_void_p(), C.c_void_p()]
    outs = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(2)]
    ev_pre, ev_scan = [torch.cuda.Event() for _ in range(2)], [torch.cuda.Event() for _ in range(2)]
    for s in states:
        check(L.apds_dev_topk_state_create(C.byref(s)))
    try:
        for rep in range(2):      # second round: the state buffers are reused
            for f, q in enumerate((da, dbq)):
                check(L.apds_dev_topk_prepass(states[f], q.data_ptr(), nq, rows.data_ptr(), nt, 7, k, C.c_void_p(st_pre.cuda_stream)))
                ev_pre[f].record(st_pre)
                st_main.wait_event(ev_pre[f])
                check(L.apds_dev_topk_scan(states[f], q.data_ptr(), rows.data_ptr(), C.c_void_p(st_main.cuda_stream)))
                ev_scan[f].record(st_main)
            for f in range(2):
                st_merge.wait_event(ev_scan[f])
                check(L.apds_dev_topk_merge(states[f], 7, outs[f].data_ptr(), C.c_void_p(st_merge.cuda_stream)))
            torch.cuda.synchronize()
            assert torch.equal(outs[0], want)
            assert torch.equal(outs[1], torch.flip(want, dims=(0,)))
    finally:
        for s in states:
            check(L.apds_dev_topk_state_destroy(s))
