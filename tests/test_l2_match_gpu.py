"""GPU: float-descriptor L2 brute-force match (BASELINE config 3) — MFMA distance GEMM vs the oracle.
No reference call site (the reference matches Hamming only): parity unpinned, oracle = this repo's scalar restatement.
Tolerance (stated): distances rtol 2e-4 + atol 2e-5 (|q|^2+|t|^2-2q.t in f32 vs the direct sum of squared differences);
neighbour indices must agree wherever the oracle's gap to the next candidate exceeds that tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2e-4, 2e-5


def _check(gpu_pkg, oracle_mod, q, t, k):
    idx, dist = gpu_pkg.feature_extraction.l2_knn_match(q, t, k)
    oracle_mod.set_threads(8)
    oi, od = oracle_mod.knn_l2(q, t, min(k + 1, len(t)))      # one extra neighbour to know the gaps
    assert np.allclose(dist, od[:, :k], rtol=RTOL, atol=ATOL), np.abs(dist - od[:, :k]).max()
    for j in range(k):
        nxt = od[:, j + 1] if od.shape[1] > j + 1 else np.full(len(q), np.inf)
        prv = od[:, j - 1] if j > 0 else np.full(len(q), -np.inf)
        clear = (nxt - od[:, j] > 4 * (RTOL * od[:, j] + ATOL)) & (od[:, j] - prv > 4 * (RTOL * od[:, j] + ATOL))
        assert np.array_equal(idx[clear, j], oi[clear, j]), j
        assert clear.mean() > 0.5      # random unit vectors in 128-D have tightly packed distances: many near-ties
    return idx, dist


@pytest.mark.parametrize("nq,nt,dim", [(1, 2, 128), (50, 300, 128), (129, 1000, 64), (1000, 5000, 128), (300, 777, 100), (5000, 20001, 128)])
def test_l2_knn_equals_oracle(gpu_pkg, oracle_mod, nq, nt, dim):
    db, q, src = gpu_pkg.synth.make_l2_set(nt, nq, dim=dim, seed=0x4C320001 + nt)
    idx, dist = _check(gpu_pkg, oracle_mod, q, db, 2 if nt >= 2 else 1)
    planted = src >= 0
    if planted.any() and nt > 10:
        assert (idx[planted, 0] == src[planted]).mean() > 0.99


def test_l2_exact_duplicates_tie_to_lower_index(gpu_pkg):
    rng = np.random.default_rng(2)
    t = rng.normal(size=(700, 128)).astype(np.float32)
    t[400] = t[13]
    t[650] = t[13]
    q = t[[13, 5]].copy()
    idx, dist = gpu_pkg.feature_extraction.l2_knn_match(q, t, 2)
    assert tuple(idx[0]) == (13, 400) and dist[0, 0] <= 1e-3 and dist[0, 1] <= 1e-3
    assert idx[1, 0] == 5


def test_l2_k1_and_empty(gpu_pkg, oracle_mod):
    db, q, _ = gpu_pkg.synth.make_l2_set(2000, 100)
    i1, d1 = gpu_pkg.feature_extraction.l2_knn_match(q, db, 1)
    i2, d2 = gpu_pkg.feature_extraction.l2_knn_match(q, db, 2)
    assert np.array_equal(i1[:, 0], i2[:, 0]) and np.array_equal(d1[:, 0], d2[:, 0])
    ie, de = gpu_pkg.feature_extraction.l2_knn_match(q, db[:0], 2)
    assert (ie == -1).all() and np.isinf(de).all()
    with pytest.raises(gpu_pkg.ApdsError):
        gpu_pkg.feature_extraction.l2_knn_match(q, db, 3)


def test_l2_config3_full_size_properties(gpu_pkg, oracle_mod):
    """BASELINE config 3 at its full size: 1,048,576 query descriptors (256 tiles x 4096) x 1,000,000 DB rows x 128 f32, top-2, through
    the device C ABI. Checked through size-independent properties + a 1/4096 sample of the queries against the oracle:
      * every planted query (30 %: DB row + N(0, 0.05^2) noise, renormalised) finds its source row first;
      * exact duplicates of a DB row inside the DB: a query equal to that row gets (lower index, higher index), distance 0;
      * halves + merge: top-2 against rows [0, N/2) and [N/2, N) with index_base, merged by u64 min, equals the one-call result
        bit for bit (the shard path of the matcher);
      * 256 sampled queries: distances within the stated tolerance of the oracle's, indices equal where the oracle's gap is clear."""
    import ctypes as C
    import torch
    pl = __import__("cubesat_apds_amd.pipeline", fromlist=["x"])
    L, check = gpu_pkg.lib(), gpu_pkg._lib.check
    dev = torch.device("cuda:0")
    nq, nt, dim = 1_048_576, 1_000_000, 128
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        g = torch.Generator(device=dev)
        g.manual_seed(0x4C33)
        db = torch.nn.functional.normalize(torch.randn((nt, dim), device=dev, generator=g), dim=1)
        db[700_001] = db[123]                                   # duplicates across the two halves
        db[999_999] = db[123]
        q = torch.nn.functional.normalize(torch.randn((nq, dim), device=dev, generator=g), dim=1)
        npl = int(0.3 * nq)
        src = torch.randint(0, nt, (npl,), device=dev, generator=g)
        q[:npl] = torch.nn.functional.normalize(db[src] + 0.05 * torch.randn((npl, dim), device=dev, generator=g), dim=1)
        q[npl] = db[123]
        out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
        check(L.apds_dev_l2_topk(q.data_ptr(), nq, db.data_ptr(), nt, dim, 0, 2, out.data_ptr(), pl.torch_stream()))
        torch.cuda.synchronize()
        idx = out & 0xFFFFFFFF
        # planted recovered (a planted query whose source row is one of the three duplicates may report the lowest of them)
        hit = (idx[:npl, 0] == src) | ((src == 700_001) | (src == 999_999))
        assert float(hit.float().mean()) > 0.9999
        # duplicates: lower index first, squared distance ~0 (f32 |q|^2+|t|^2-2q.t of a unit vector with itself: < 1e-5)
        k0, k1 = int(out[npl, 0].item()), int(out[npl, 1].item())
        assert (k0 & 0xFFFFFFFF, k1 & 0xFFFFFFFF) == (123, 700_001)
        assert np.array([k0 >> 32, k1 >> 32], np.uint32).view(np.float32).max() < 1e-5
        # halves + merge == whole
        half = nt // 2
        parts = torch.empty((2, nq, 2), dtype=torch.int64, device=dev)
        check(L.apds_dev_l2_topk(q.data_ptr(), nq, db.data_ptr(), half, dim, 0, 2, parts[0].data_ptr(), pl.torch_stream()))
        check(L.apds_dev_l2_topk(q.data_ptr(), nq, db[half:].data_ptr(), nt - half, dim, half, 2, parts[1].data_ptr(), pl.torch_stream()))
        merged = torch.empty((nq, 2), dtype=torch.int64, device=dev)
        check(L.apds_dev_merge_topk(parts.data_ptr(), 2, nq, 2, merged.data_ptr(), pl.torch_stream()))
        torch.cuda.synchronize()
        assert torch.equal(merged, out)
        # a 1/4096 sample of the queries against the oracle
        sel = torch.arange(0, nq, 4096, device=dev)
        qs, keys = q[sel].cpu().numpy(), out[sel].cpu().numpy()
        dbh = db.cpu().numpy()
    oracle_mod.set_threads(16)
    oi, od = oracle_mod.knn_l2(qs, dbh, 3)
    gi = (keys & 0xFFFFFFFF).astype(np.int64)
    gd = np.sqrt((keys >> 32).astype(np.uint32).view(np.float32).reshape(keys.shape))
    assert np.allclose(gd, od[:, :2], rtol=RTOL, atol=ATOL), np.abs(gd - od[:, :2]).max()
    for j in range(2):
        prv = od[:, j - 1] if j > 0 else np.full(len(qs), -np.inf)
        clear = (od[:, j + 1] - od[:, j] > 4 * (RTOL * od[:, j] + ATOL)) & (od[:, j] - prv > 4 * (RTOL * od[:, j] + ATOL))
        assert np.array_equal(gi[clear, j], oi[clear, j]) and clear.mean() > 0.5


def _dev_topk(gpu_pkg, q, t, mode, index_base=0):
    """apds_dev_l2_topk_ex on device copies of q, t -> (keys [nq, 2] int64 numpy, mode_used, candidates per query)"""
    import ctypes as C
    import torch
    pl = __import__("cubesat_apds_amd.pipeline", fromlist=["x"])
    L, check = gpu_pkg.lib(), gpu_pkg._lib.check
    dev = torch.device("cuda:0")
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
        out = torch.empty((len(q), 2), dtype=torch.int64, device=dev)
        used, cpq = C.c_int(-1), C.c_double(0)
        check(L.apds_dev_l2_topk_ex(dq.data_ptr(), len(q), dt.data_ptr(), len(t), q.shape[1], index_base, 2, mode, out.data_ptr(), pl.torch_stream(), C.byref(used),
                                    C.byref(cpq)))
        torch.cuda.synchronize()
        return out.cpu().numpy(), used.value, cpq.value


@pytest.mark.parametrize("nq,nt,kind", [(300, 2000, "unit"), (1000, 20001, "unit"), (4096, 100000, "unit"), (513, 7777, "raw"), (257, 3000, "dups"), (100, 2, "unit")])
def test_l2_bf16_screen_returns_the_exact_modes_keys(gpu_pkg, nq, nt, kind):
    # APDS_L2_SCREEN (bf16 MFMA screen with a proved bound + f32 re-rank) must return the keys of APDS_L2_EXACT bit for bit:
    # unit-norm descriptors with planted matches, raw Gaussian rows of very different norms, and a DB full of exact duplicates
    rng = np.random.default_rng(nt)
    if kind == "unit":
        db, q, _ = gpu_pkg.synth.make_l2_set(nt, nq, dim=128, seed=0x4C320100 + nt)
    elif kind == "raw":
        db = (rng.normal(size=(nt, 128)) * rng.uniform(0.01, 30.0, size=(nt, 1))).astype(np.float32)
        q = (rng.normal(size=(nq, 128)) * rng.uniform(0.01, 30.0, size=(nq, 1))).astype(np.float32)
    else:
        base = rng.normal(size=(nt // 10, 128)).astype(np.float32)
        db = base[rng.integers(0, len(base), nt)]                       # every row appears ~10 times
        q = db[rng.integers(0, nt, nq)] + (rng.normal(size=(nq, 128)) * 1e-3).astype(np.float32)
    exact, used0, _ = _dev_topk(gpu_pkg, q, db, 0)
    screen, used1, cpq = _dev_topk(gpu_pkg, q, db, 1, index_base=0)
    assert used0 == 0 and used1 == 1, (used0, used1)
    assert np.array_equal(exact, screen), (np.nonzero((exact != screen).any(1))[0][:5], cpq)
    assert cpq >= 2.0 - 1e-9


def test_l2_screen_falls_back_when_it_does_not_apply(gpu_pkg):
    db, q, _ = gpu_pkg.synth.make_l2_set(500, 50, dim=64)
    exact, used0, _ = _dev_topk(gpu_pkg, q, db, 0)
    screen, used1, _ = _dev_topk(gpu_pkg, q, db, 1)
    assert used1 == 0 and np.array_equal(exact, screen)      # dim != 128: the exact kernel ran
