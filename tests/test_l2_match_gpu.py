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
