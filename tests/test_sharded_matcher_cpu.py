"""CPU, world_size 2 and 4 (gloo): the multi-GPU match choreography - DB row shards, all-gather of queries, per-shard top-k with
global indices, all-to-all of the keys (each rank receives only its own queries' candidates), per-rank merge - gives every rank
exactly the single-device result for its own queries, in the one-call form, with the counts exchanged ahead, and in the split form a
pipeline uses (frame i+1's query gather issued before frame i's key exchange).

What runs is the PRODUCT's choreography text, csrc/shard_core.h (the code behind apds_shard_* in libapds_hip.so), compiled here by g++
into a host-only test library (tests/cpp/shard_host.cpp: device memory = host memory, the local compute - top-k of a shard, u64-min
merge - injected from this file with oracle + numpy because the HIP kernels need a GPU), driven through the same host-callback
transport (cubesat-apds_amd/pipeline.py: host_transport_from_group) the one-GPU rehearsal uses. The same text over the loopback and
RCCL transports runs on the GPU box (tests/test_shard_native.py)."""
import ctypes as C
import os
import socket
import subprocess
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


def build_shard_host(out_dir):
    so = os.path.join(str(out_dir), "libshard_host.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "shard_host.cpp"), "-o", so])
    return so


TOPK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_void_p)
MERGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p)


def _view(addr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)).view(dtype).reshape(shape)


class HostShard:
    """The test library's C entry points (the CPU twin of apds_shard_*)."""

    def __init__(self, so, oracle, rank, world, transport, rows64, index_base):
        self.L = L = C.CDLL(so)
        L.shardhost_last_error.restype = C.c_char_p
        self.rows64 = np.ascontiguousarray(rows64)

        def topk(_u, q, nq, rows, n_rows, base, k, out):      # what hamming_topk_kernel computes, by the oracle
            keys = _view(out, (nq, k), np.uint64)
            keys[:] = np.uint64(0xFFFFFFFFFFFFFFFF)
            if nq and n_rows:
                idx, d = oracle.knn_hamming(_view(q, (nq, 64), np.uint8)[:, :61].copy(), _view(rows, (n_rows, 64), np.uint8)[:, :61].copy(), k)
                got = (d.astype(np.uint64) << np.uint64(32)) | (idx.astype(np.int64) + base).astype(np.uint64)
                got[idx < 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
                keys[:] = got

        def merge(_u, parts, nparts, nq, k, out):              # what merge_topk_kernel computes
            p = _view(parts, (nparts, nq, k), np.uint64)
            _view(out, (nq, k), np.uint64)[:] = np.sort(np.concatenate(list(p), axis=1), axis=1)[:, :k]

        self._cb = (TOPK_FN(topk), MERGE_FN(merge), transport)
        self.h = C.c_void_p()
        self.check(L.shardhost_create(C.byref(self.h), rank, world, C.byref(transport), self.rows64.ctypes.data_as(C.c_void_p), C.c_int64(len(self.rows64)),
                                      C.c_uint32(index_base), self._cb[0], self._cb[1], None))

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(f"shardhost error {rc}: {self.L.shardhost_last_error().decode()}")

    def counts(self, nq, world):
        c = (C.c_int * world)()
        self.check(self.L.shardhost_counts(self.h, nq, c))
        return list(c)

    def knn(self, q64, k, counts=None):
        out = np.zeros((len(q64), k), np.uint64)
        ca = (C.c_int * len(counts))(*counts) if counts is not None else None
        self.check(self.L.shardhost_knn(self.h, q64.ctypes.data_as(C.c_void_p), len(q64), ca, k, out.ctypes.data_as(C.c_void_p)))
        return out

    def knn_replicated(self, q64, nq, root, k):
        out = np.zeros((nq, k), np.uint64)
        self.check(self.L.shardhost_knn_replicated(self.h, q64.ctypes.data_as(C.c_void_p) if q64 is not None else None, nq, root, k, out.ctypes.data_as(C.c_void_p)))
        return out

    def slot(self, max_queries, kmax):
        s = C.c_void_p()
        self.check(self.L.shardhost_slot_create(self.h, max_queries, kmax, C.byref(s)))
        return s

    def gather(self, slot, q64, counts):
        return self.L.shardhost_gather(self.h, slot, q64.ctypes.data_as(C.c_void_p), len(q64), (C.c_int * len(counts))(*counts))

    def scan(self, slot, k):
        self.check(self.L.shardhost_scan(self.h, slot, k))

    def exchange_merge(self, slot, nq, k):
        out = np.zeros((nq, k), np.uint64)
        self.check(self.L.shardhost_exchange_merge(self.h, slot, k, out.ctypes.data_as(C.c_void_p)))
        return out


def _worker(rank, world, port, nt, nqs, result_dir, so):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    import oracle
    pkg = graft.load_package()
    from cubesat_apds_amd import pipeline as pl
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

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
    transport = pl.host_transport_from_group(dist, dist.group.WORLD)    # the product's callbacks over this gloo group
    m = HostShard(so, oracle, rank, world, transport, db64[lo:hi], lo)
    keys = m.knn(q64, 2)                                            # counts exchanged inside the call
    counts = m.counts(len(q64), world)                              # second form: counts exchanged ahead
    keys2 = m.knn(q64, 2, counts=counts)
    # third form, as the streamed pipeline issues it: two frames in flight (A = these queries, B = the same rows reversed),
    # B's query gather goes out before A's key exchange; each frame owns its slot
    qb = np.ascontiguousarray(q64[::-1])
    slots = [m.slot(max(counts), 2), m.slot(max(counts), 2)]
    ok = m.gather(slots[0], q64, counts) == 0
    m.scan(slots[0], 2)
    ok = ok and m.gather(slots[1], qb, counts) == 0
    keys3 = m.exchange_merge(slots[0], len(q64), 2)
    m.scan(slots[1], 2)
    keys3b = m.exchange_merge(slots[1], len(q64), 2)
    # a frame larger than the slot is refused on every rank alike (the size rule depends on the counts only), before any collective
    big = [c + 5000 for c in counts]
    ok = ok and m.gather(slots[0], np.zeros((big[rank], 64), np.uint8), big) == -215
    # the strong-scaling form (SURVEY 8e's literal shape): ONE frame's queries (a seed every rank knows), replicated on every rank or brought
    # by the last rank alone and broadcast; per-shard top-k, ALL-GATHER of the key lists, merge: every rank ends with the whole answer.
    # k = 3 also goes through a merge width that is not a power of two.
    fq, _ = pkg.synth.make_queries(db, 41, seed=77)
    fq[0] = db[7]
    fq64 = np.zeros((len(fq), 64), np.uint8)
    fq64[:, :61] = fq
    for kk in (2, 3):
        f_idx, f_d = oracle.knn_hamming(fq, db, kk)
        f_want = (f_d.astype(np.uint64) << np.uint64(32)) | f_idx.astype(np.uint64)
        ok = ok and np.array_equal(m.knn_replicated(fq64, len(fq64), -1, kk), f_want)
        ok = ok and np.array_equal(m.knn_replicated(fq64 if rank == world - 1 else None, len(fq64), world - 1, kk), f_want)
    want_idx, want_d = oracle.knn_hamming(q, db, 2)
    got_idx = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    got_d = (keys >> np.uint64(32)).astype(np.int64)
    ok = ok and np.array_equal(got_idx, want_idx) and np.array_equal(got_d, want_d)
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
    mp.spawn(_worker, args=(world, port, nt, nqs, str(tmp_path), build_shard_host(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok", r
