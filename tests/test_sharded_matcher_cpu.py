"""CPU, world_size 2 and 4 (gloo): the multi-GPU match choreography of cubesat-apds_amd/pipeline.py — DB row shards,
all-gather of queries, per-shard top-k with global indices, all-to-all of the keys (each rank receives only its own
queries' candidates), per-rank merge — gives every rank exactly the single-device result for its own queries, in the
one-call form, with the counts exchanged ahead on the host, and in the split form a pipeline uses (frame i+1's query
gather issued before frame i's key exchange). The local compute is injected (oracle + numpy) because the HIP
kernels need a GPU; what is under test is the sharding/collective logic that runs unchanged over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nt, nqs, result_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    import oracle
    pkg = graft.load_package()
    from cubesat_apds_amd import pipeline as pl
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class CpuBackend:   # stands in for HipBackend: same contract, numpy/oracle arithmetic
        def topk(self, q, train, index_base, k, out=None):
            idx, d = oracle.knn_hamming(q.numpy()[:, :61], train.numpy()[:, :61], k)
            keys = (d.astype(np.uint64) << np.uint64(32)) | (idx.astype(np.int64) + index_base).astype(np.uint64)
            keys[idx < 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            res = torch.from_numpy(keys.view(np.int64).copy())
            if out is not None:
                out.copy_(res)
                return out
            return res

        def merge(self, parts, k, out=None):
            assert parts.is_contiguous()
            p = parts.numpy().view(np.uint64)                      # [P, Q, k]
            allk = np.sort(np.concatenate(list(p), axis=1), axis=1)[:, :k]
            res = torch.from_numpy(allk.view(np.int64).copy())
            if out is not None:
                out.copy_(res)
                return out
            return res

    db = pkg.synth.make_descriptor_db(nt, seed=123)
    db[nt // 2 + 5] = db[7]                                         # a cross-shard tie: lower global index must win
    db64 = np.zeros((nt, 64), np.uint8)
    db64[:, :61] = db
    lo, hi = rank * nt // world, (rank + 1) * nt // world
    q, _ = pkg.synth.make_queries(db, nqs[rank], seed=1000 + rank)
    if rank == 0 and len(q):
        q[0] = db[7]
    q64 = np.zeros((len(q), 64), np.uint8)
    q64[:, :61] = q
    meta = dist.new_group(backend="gloo")                          # the host-side group bench.py uses for the per-frame counts
    m = pl.ShardedMatcher(torch.from_numpy(db64[lo:hi].copy()), lo, group=dist.group.WORLD, backend=CpuBackend(), pad_rows=16, meta_group=meta)
    keys = m.knn(torch.from_numpy(q64), 2).numpy().view(np.uint64).copy()
    counts = m.exchange_counts(len(q64))                            # second form: counts exchanged ahead on the host
    keys2 = m.knn(torch.from_numpy(q64), 2, counts=counts).numpy().view(np.uint64).copy()
    # third form, as the streamed pipeline issues it: two frames in flight (A = these queries, B = the same rows reversed),
    # B's query gather goes out before A's key exchange; each frame owns its buffers
    qb = torch.from_numpy(q64[::-1].copy())
    bufs = [m.make_buffers(max(counts)), m.make_buffers(max(counts))]
    m.gather_queries(torch.from_numpy(q64), counts, bufs[0])
    m.scan_gathered(bufs[0], 2)
    m.gather_queries(qb, counts, bufs[1])
    keys3 = m.exchange_merge(bufs[0], 2).numpy().view(np.uint64).copy()
    m.scan_gathered(bufs[1], 2)
    out_b = torch.empty((max(len(q64), 1), 2), dtype=torch.int64)
    keys3b = m.exchange_merge(bufs[1], 2, out=out_b).numpy().view(np.uint64).copy()
    want_idx, want_d = oracle.knn_hamming(q, db, 2)
    got_idx = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    got_d = (keys >> np.uint64(32)).astype(np.int64)
    ok = np.array_equal(got_idx, want_idx) and np.array_equal(got_d, want_d)
    ok = ok and counts == list(nqs) and np.array_equal(keys, keys2) and np.array_equal(keys, keys3) and np.array_equal(keys[::-1], keys3b)
    if rank == 0 and len(q):
        ok = ok and tuple(got_idx[0]) == (7, nt // 2 + 5) and tuple(got_d[0]) == (0, 0)
    open(os.path.join(result_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nt,nqs", [(600, (37, 90)), (1001, (0, 5)), (803, (20, 0, 33, 7))])
def test_sharded_match_gloo(tmp_path, nt, nqs):
    import torch.multiprocessing as mp
    port, world = _free_port(), len(nqs)
    mp.spawn(_worker, args=(world, port, nt, nqs, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok", r
