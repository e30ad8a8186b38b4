"""Device-resident composition of the hot path and its multi-GPU form (python fronts of the C ABI's apds_shard_* and apds_pipeline_*).

frame (u8, HBM) -> AKAZE -> descriptors -> Hamming top-2 against a resident descriptor DB -> ratio test ->
matched points -> RANSAC homography.  The reference only chains these steps inside unit tests
(/root/reference/feature_extraction/src/lib.rs:197-249) and never calls find_homography_mat on the result; this
module is the composed pipeline the north-star metric (frames/s) is measured on.

Multi-GPU (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm):
  * the descriptor DB is row-sharded: rank r holds rows [base_r, base_r + n_r) resident in its HBM;
  * frames are data-parallel: every rank extracts its own frame;
  * the only exchange is in the match step, and it lives behind the C ABI (apds_shard_*, csrc/shard_core.h): all-gather of the
    ranks' query descriptors, local top-k of ALL queries against the local shard, all-to-all of the per-shard top-k keys
    (packed u64 = distance << 32 | global row), then each rank merges the candidates of its own queries.
    u64 min-merge reproduces the single-GPU result exactly, including the lowest-index tie break.
torch is used for device buffers and streams only; every compute step is a libapds_hip kernel and the collectives are the library's
(RCCL; or the host-callback transport over a gloo group when several ranks share one GPU).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

EMPTY_KEY = -1  # 0xFFFF_FFFF_FFFF_FFFF as int64


def torch_stream():
    """torch's current HIP stream as the void* the C ABI takes, so kernels, torch ops and collectives are ordered on ONE
    stream. torch's default stream has handle 0, which the C ABI reads as "the calling thread's own stream" (a different,
    non-blocking stream): callers must therefore work inside `with torch.cuda.stream(torch.cuda.Stream())`."""
    h = torch.cuda.current_stream().cuda_stream
    if h == 0:
        raise RuntimeError("run the pipeline under an explicit torch.cuda.Stream (the default stream's handle is NULL)")
    return C.c_void_p(h)


def _gather_into(dist, group, dst, src):
    """all_gather_into_tensor that also works for the gloo rehearsal backend with device tensors (staged through the host). Used by
    bench.py's SETUP only (building the shared DB); the per-frame exchange is the C ABI's (apds_shard_*)."""
    if dist.get_backend(group) == "gloo" and src.is_cuda:
        d, s = torch.empty(dst.shape, dtype=dst.dtype), src.cpu()
        dist.all_gather_into_tensor(d.view(-1), s.view(-1), group=group)
        dst.copy_(d)
    else:
        dist.all_gather_into_tensor(dst.view(-1), src.view(-1), group=group)


def host_transport_from_group(dist, group):
    """apds_host_transport (include/apds.h) over a torch.distributed process group whose collectives take HOST tensors (gloo): the two
    callbacks wrap the library's staging buffers as tensors without copying. This is what carries the exchange step when several ranks
    share one GPU (the one-GPU rehearsal of bench.py, APDS_BENCH_BACKEND=gloo) and in the CPU tests. Returns the ctypes struct; it keeps
    the callback objects alive."""
    world = dist.get_world_size(group)

    def view(addr, n):
        if not n:
            return torch.empty(0, dtype=torch.uint8)
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(int(n),)))

    def all_gather(_user, send, recv, nbytes):
        try:
            dist.all_gather_into_tensor(view(recv, nbytes * world), view(send, nbytes), group=group)
            return 0
        except Exception:   # noqa: BLE001 - an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    def all_to_all(_user, send, soff, sbytes, recv, roff, rbytes):
        try:
            so, sb, ro, rb = ([int(a[p]) for p in range(world)] for a in (soff, sbytes, roff, rbytes))
            # all_to_all_single wants the blocks back to back in rank order, which is how the matcher lays them out
            assert all(so[p] == sum(sb[:p]) for p in range(world)) and all(ro[p] == sum(rb[:p]) for p in range(world)), "blocks are not contiguous"
            dist.all_to_all_single(view(recv, sum(rb)), view(send, sum(sb)), output_split_sizes=rb, input_split_sizes=sb, group=group)
            return 0
        except Exception:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    ht = _lib.HostTransport(None, _lib.HOST_ALL_GATHER(all_gather), _lib.HOST_ALL_TO_ALL(all_to_all))
    ht._keep = (all_gather, all_to_all)
    return ht


class _DevBytes:
    """A library-owned device buffer as something torch.as_tensor can wrap without a copy (__cuda_array_interface__; ROCm builds read it too)."""

    def __init__(self, addr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(addr), False), "version": 2, "strides": None}


def device_transport_from_group(dist, group, device):
    """apds_device_transport (include/apds.h) over a torch.distributed process group whose collectives take DEVICE tensors ("nccl" = RCCL):
    the library's exchange buffers are wrapped as tensors in place and the collectives are issued on the stream the library names, so the
    exchange travels over torch's own RCCL communicator. The second way onto xGMI: bench.py falls back to it (--transport torch) if the
    library's own communicator (APDS_TRANSPORT_RCCL: ncclCommInitRank beside torch's) cannot be set up on a node."""
    world = dist.get_world_size(group)

    def view(addr, n):
        if not n:
            return torch.empty(0, dtype=torch.uint8, device=device)
        return torch.as_tensor(_DevBytes(addr, n), device=device)

    def all_gather(_user, send, recv, nbytes, stream):
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=device)):
                dist.all_gather_into_tensor(view(recv, nbytes * world), view(send, nbytes), group=group)
            return 0
        except Exception:   # noqa: BLE001 - an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    def all_to_all(_user, send, soff, sbytes, recv, roff, rbytes, stream):
        try:
            so, sb, ro, rb = ([int(a[p]) for p in range(world)] for a in (soff, sbytes, roff, rbytes))
            assert all(so[p] == sum(sb[:p]) for p in range(world)) and all(ro[p] == sum(rb[:p]) for p in range(world)), "blocks are not contiguous"
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=device)):
                dist.all_to_all_single(view(recv, sum(rb)), view(send, sum(sb)), output_split_sizes=rb, input_split_sizes=sb, group=group)
            return 0
        except Exception:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    dt = _lib.DeviceTransport(None, _lib.DEV_ALL_GATHER(all_gather), _lib.DEV_ALL_TO_ALL(all_to_all))
    dt._keep = (all_gather, all_to_all)
    return dt


class GatheredQueries:
    """One frame's exchange slot of a shard handle (apds_shard_slot_create): the buffers live in the library; `merged` is where
    exchange_merge leaves the keys when the caller passes no output tensor."""

    def __init__(self, matcher, max_queries, kmax):
        self.matcher, self.kmax = matcher, kmax
        self.handle = C.c_void_p()
        check(lib().apds_shard_slot_create(matcher.handle, int(max_queries), int(kmax), C.byref(self.handle)))
        self.pad = -(-max(int(max_queries), 1) // 1024) * 1024
        self.merged = torch.empty((self.pad, kmax), dtype=torch.int64, device=matcher.rows.device)
        self.counts, self.total, self.nq = None, 0, 0

    def close(self):
        h, self.handle = self.handle, None
        if h and self.matcher.handle:
            lib().apds_shard_slot_destroy(self.matcher.handle, h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


class ShardedMatcher:
    """Hamming k-NN of per-rank query sets against a row-sharded resident DB: a thin front of the C ABI's apds_shard_* (csrc/shard.cpp,
    csrc/shard_core.h), which owns the whole choreography - all-gather of the ranks' query rows, local top-k of all of them against the
    local shard with global row indices, all-to-all of the keys, u64-min merge (= the single-GPU result, lowest-index tie-break
    included). What this class adds is the choice of transport from the torch process group it is given:
      * no group / one rank: no exchange step, the scan is called directly;
      * a "nccl" group: the library's RCCL transport (its own communicator; rank 0's id travels through the torch group once), or -
        transport="torch" - device callbacks that run the same two collectives through torch.distributed on that group;
      * a "gloo" group: the host-callback transport over that group (several ranks on one GPU: rehearsal).
    The three steps are separate calls (gather_queries / scan_gathered / exchange_merge) so that a pipeline can issue frame i+1's query
    gather on another stream BEFORE frame i's key exchange. All collective calls must come from ONE thread in the same order on every
    rank. No per-frame allocations."""

    def __init__(self, local_rows64, index_base, group=None, pad_rows=32768, meta_group=None, kmax=2, always_exchange=False, transport="rccl"):
        self.rows = local_rows64            # borrowed by the shard handle for its whole life
        self.index_base = int(index_base)
        self.group = group
        self.meta_group = meta_group      # optional host-side (gloo) group for the per-frame query counts
        self.pad_rows = pad_rows          # smallest slot capacity (rows per rank)
        self.kmax = kmax
        self.always_exchange = always_exchange   # run the collectives even with one rank (RCCL self-test on a one-GPU box)
        self._own = None                  # slot of the plain knn() form
        self.handle = None
        self.transport = "none"
        if group is not None:
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.dist, self.world, self.rank = None, 1, 0
        if self.world == 1 and not (always_exchange and self.dist is not None):
            return
        L, dist = lib(), self.dist
        backend = dist.get_backend(group)
        h = C.c_void_p()
        n_rows = int(local_rows64.shape[0])
        if backend == "nccl" and transport == "torch":
            # the exchange over torch's own RCCL communicator: device callbacks on this group
            self._dev_transport = device_transport_from_group(dist, group, local_rows64.device)
            check(L.apds_shard_create(C.byref(h), self.rank, self.world, _lib.TRANSPORT_DEVICE, None, C.byref(self._dev_transport), local_rows64.data_ptr(), n_rows,
                                      self.index_base))
            self.transport = "device-callbacks (torch.distributed nccl)"
        elif backend == "nccl":
            # rank 0 creates the RCCL id; 128 bytes through the torch group; every rank then joins the library's own communicator
            cid = _lib.CommId()
            if self.rank == 0:
                check(L.apds_comm_id_create(_lib.TRANSPORT_RCCL, C.byref(cid)))
            t = torch.frombuffer(bytearray(bytes(cid)), dtype=torch.uint8).to(local_rows64.device)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0, group=group)
            C.memmove(C.byref(cid), bytes(t.cpu().numpy().tobytes()), _lib.COMM_ID_BYTES)
            check(L.apds_shard_create(C.byref(h), self.rank, self.world, _lib.TRANSPORT_RCCL, C.byref(cid), None, local_rows64.data_ptr(), n_rows, self.index_base))
            self.transport = "rccl"
        else:
            self._host_transport = host_transport_from_group(dist, group)
            check(L.apds_shard_create(C.byref(h), self.rank, self.world, _lib.TRANSPORT_HOST, None, C.byref(self._host_transport), local_rows64.data_ptr(), n_rows,
                                      self.index_base))
            self.transport = "host-callbacks (%s)" % backend
        self.handle = h

    def close(self):
        if self._own is not None:
            self._own.close()
            self._own = None
        h, self.handle = self.handle, None
        if h:
            lib().apds_shard_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def info(self):
        """{rank, world, rows, index_base, transport, rccl_version} as the library reports them."""
        if not self.handle:
            return dict(rank=0, world=1, rows=int(self.rows.shape[0]), index_base=self.index_base, transport="none (one rank)", rccl_version=None)
        r, w, ver, n, base, name = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_uint32(), C.c_char_p()
        check(lib().apds_shard_info(self.handle, C.byref(r), C.byref(w), C.byref(n), C.byref(base), C.byref(name), C.byref(ver)))
        return dict(rank=r.value, world=w.value, rows=n.value, index_base=base.value, transport=name.value.decode(), rccl_version=ver.value)

    def exchange_counts(self, nq):
        """Every rank's query count for one frame, as host ints, through the host-side group: no device work and no stream
        synchronisation, so the thread that later issues the match keeps queueing kernels ahead of the GPU. Must be called
        once per frame, in frame order, by the same thread on every rank."""
        if self.handle is None:
            return [int(nq)]
        if self.meta_group is None:
            return None
        t = torch.tensor([int(nq)], dtype=torch.int64)
        o = torch.empty(self.world, dtype=torch.int64)
        self.dist.all_gather_into_tensor(o, t, group=self.meta_group)
        return [int(v) for v in o.tolist()]

    def make_buffers(self, max_queries=None, k=None):
        """An exchange slot for frames of at most `max_queries` queries per rank (one per frame in flight)."""
        return GatheredQueries(self, max(self.pad_rows, int(max_queries or 0)), k or self.kmax)

    @staticmethod
    def _counts_arg(counts):
        return (C.c_int * len(counts))(*[int(c) for c in counts])

    def gather_queries(self, q_rows64, counts, buf):
        """Step (1) on the CURRENT stream (collective): afterwards the slot holds every rank's queries."""
        nq = int(q_rows64.shape[0])
        buf.counts, buf.total, buf.nq = list(counts), int(sum(counts)), nq
        check(lib().apds_shard_gather(self.handle, buf.handle, q_rows64.data_ptr() if nq else None, nq, self._counts_arg(counts), torch_stream()))
        return buf

    def scan_gathered(self, buf, k=2):
        """Step (2) on the CURRENT stream: this shard's top-k of every gathered query (no collective; waits for the gather's event)."""
        check(lib().apds_shard_scan(self.handle, buf.handle, k, torch_stream()))

    def exchange_merge(self, buf, k=2, out=None):
        """Steps (3)-(4) on the CURRENT stream (the one scan_gathered ran on; collective): all-to-all of the keys, then the per-query merge."""
        nq = buf.nq
        dst = out[:nq] if out is not None else buf.merged[:nq, :k] if k == buf.kmax else torch.empty((nq, k), dtype=torch.int64, device=self.rows.device)
        check(lib().apds_shard_exchange_merge(self.handle, buf.handle, k, dst.data_ptr() if nq else None, torch_stream()))
        return dst

    def match_gathered(self, buf, k=2, out=None):
        """Steps (2)-(4) on the CURRENT stream for a frame whose queries gather_queries() has put into `buf`."""
        self.scan_gathered(buf, k)
        return self.exchange_merge(buf, k, out=out)

    def knn(self, q_rows64, k=2, out=None, counts=None):
        """q_rows64: this rank's queries [Q_r, 64] u8. Returns [Q_r, k] int64 keys over the WHOLE DB. `counts`: every rank's
        query count (exchange_counts); without it the library exchanges them first, which costs a host synchronisation."""
        nq = int(q_rows64.shape[0])
        dst = out[:nq] if out is not None else torch.empty((nq, k), dtype=torch.int64, device=q_rows64.device)
        if self.handle is None:
            check(lib().apds_dev_hamming_topk(q_rows64.data_ptr(), nq, self.rows.data_ptr(), int(self.rows.shape[0]), self.index_base, k, dst.data_ptr(), torch_stream()))
            return dst
        check(lib().apds_shard_knn(self.handle, q_rows64.data_ptr() if nq else None, nq, self._counts_arg(counts) if counts is not None else None, k,
                                   dst.data_ptr() if nq else None, torch_stream()))
        return dst


class FramePipeline:
    """One rank's resident state: frames, DB shard (+ the DB rows' keypoint coordinates), output buffers."""

    def __init__(self, db_rows64, db_xy, index_base=0, group=None, max_points=(1 << 18) - 1, device="cuda:0"):
        self.dev = torch.device(device)
        self.stream = torch.cuda.Stream(self.dev)
        self.matcher = ShardedMatcher(db_rows64, index_base, group)
        self.db_xy_all = db_xy                      # [N_total, 2] f32 on device (replicated: 8 B/row)
        self.cap = max_points
        self.kps = torch.empty((self.cap, 7), dtype=torch.float32, device=self.dev)       # 28-byte cv::KeyPoint rows
        self.desc = torch.empty((self.cap, 64), dtype=torch.uint8, device=self.dev)
        self.matches = torch.empty((self.cap, 4), dtype=torch.int32, device=self.dev)     # 16-byte cv::DMatch rows
        self.p1 = torch.empty((self.cap, 2), dtype=torch.float32, device=self.dev)
        self.p2 = torch.empty((self.cap, 2), dtype=torch.float32, device=self.dev)
        self.mask = torch.empty(self.cap, dtype=torch.uint8, device=self.dev)

    def step(self, frame, filter_strength=0.8, reproj_thr=3.0, max_iters=2000, confidence=0.995):
        """frame: [H, W, C] u8 device tensor. Returns dict(n_keypoints, n_matches, H (3x3 numpy) or None, n_inliers)."""
        with torch.cuda.stream(self.stream):
            return self._step(frame, filter_strength, reproj_thr, max_iters, confidence)

    def _step(self, frame, filter_strength, reproj_thr, max_iters, confidence):
        L = lib()
        h, w = frame.shape[0], frame.shape[1]
        ch = 1 if frame.dim() == 2 else frame.shape[2]
        n = C.c_int(0)
        check(L.apds_dev_akaze_extract(frame.data_ptr(), h, w, ch, frame.stride(0), self.cap, self.kps.data_ptr(), self.desc.data_ptr(), self.cap,
                                       C.byref(n), torch_stream()))
        K = n.value
        out = dict(n_keypoints=K, n_matches=0, H=None, n_inliers=0)
        keys = self.matcher.knn(self.desc[:K], 2)
        if K == 0:
            return out
        nm = C.c_int(0)
        check(L.apds_dev_ratio_filter(keys.data_ptr(), K, 2, float(filter_strength), self.matches.data_ptr(), C.byref(nm), torch_stream()))
        M = nm.value
        out["n_matches"] = M
        if M < 4:
            return out
        # query_idx -> this frame's keypoints, train_idx -> DB row coordinates (28-byte rows with x,y first are not needed:
        # the DB side keeps only xy, so the gather is done on packed float2 rows)
        check(L.apds_dev_points_from_matches(self.kps.data_ptr(), K, self._db_kp_view().data_ptr(), self.db_xy_all.shape[0], self.matches.data_ptr(), M, 0,
                                             self.p1.data_ptr(), self.p2.data_ptr(), torch_stream()))
        H = np.zeros(9, np.float64)
        rc = L.apds_dev_find_homography(self.p1.data_ptr(), self.p2.data_ptr(), M, 8, float(reproj_thr), int(max_iters), float(confidence),
                                        _lib.ptr(H), self.mask.data_ptr(), torch_stream())
        if rc == 0:
            out["H"] = H.reshape(3, 3)
            out["n_inliers"] = int(self.mask[:M].sum().item())
        elif rc != _lib.ERR_EMPTY:
            check(rc)
        return out

    def _db_kp_view(self):
        if not hasattr(self, "_db_kp"):
            kp = torch.zeros((self.db_xy_all.shape[0], 7), dtype=torch.float32, device=self.dev)
            kp[:, 0:2] = self.db_xy_all
            self._db_kp = kp
        return self._db_kp


class StreamedFramePipeline:
    """The same path as FramePipeline, software-pipelined over a stream of frames - a FRONT of the library's own pipeline
    (apds_pipeline_create / _submit / _poll / _stats / _destroy, csrc/pipeline.cpp): the stage threads (two extraction workers, the match
    worker with its three streams, the homography worker), their HIP streams, events, slots and the starvation watch live inside
    libapds_hip.so, so a Rust or C++ host gets the same frames/s from four calls. What this class adds: the torch tensors behind the
    pointers (train rows, their keypoints, frames), the transport choice for a sharded DB (ShardedMatcher), and results as dicts."""

    def __init__(self, db_rows64, db_xy, index_base=0, group=None, max_points=(1 << 18) - 1, device="cuda:0", slots=6, reserve_cus=0,
                 n_cus=256, meta_group=None, extract_workers=None, transport="rccl"):
        import os
        self.dev = torch.device(device)
        self.matcher = ShardedMatcher(db_rows64, index_base, group, meta_group=meta_group, transport=transport)
        self.n_db = db_xy.shape[0]
        kp = torch.zeros((self.n_db, 7), dtype=torch.float32, device=self.dev)     # 28-byte cv::KeyPoint rows of ALL train rows (x, y used)
        kp[:, 0:2] = db_xy
        self.db_kp = kp
        self.cap = max_points
        self.n_slots = slots
        self.extract_workers = int(extract_workers if extract_workers is not None else os.environ.get("APDS_EXTRACT_WORKERS", "0") or 0)
        self.debug_extract_delay = float(os.environ.get("APDS_DEBUG_EXTRACT_DELAY_MS", "0")) * 1e-3    # test hook (seconds)
        self.cap_bytes = 0              # occupancy cap this pipeline's scans run with (set by the starvation watch; kept across re-creations)
        self.cap_events = []
        self.gap_log = []               # idle time of the match stream before each frame's main scan (ms; the first 16 of a run)
        self.gap_mean = None
        self._masked_stream_handle = None
        self._pipe, self._key = None, None
        if reserve_cus > 0:
            # keep `reserve_cus` CUs (spread evenly over the CU index space) out of the MATCH stream only
            words = (n_cus + 31) // 32
            mask = np.full(words, 0xFFFFFFFF, np.uint32)
            stride = n_cus // reserve_cus
            for cu in (r * stride for r in range(reserve_cus)):
                mask[cu // 32] &= ~np.uint32(1 << (cu % 32))
            h = C.c_void_p()
            check(lib().apds_stream_create(0, _lib.ptr(mask), words, C.byref(h)))
            self._masked_stream_handle = h
        torch.cuda.synchronize()

    # ---- the native handle ------------------------------------------------------------------------------------------------------------
    def _ensure(self, shape, filter_strength, reproj_thr, max_iters, confidence, timing):
        rows, cols = int(shape[0]), int(shape[1])
        ch = 1 if len(shape) == 2 else int(shape[2])
        key = (rows, cols, ch, float(filter_strength), float(reproj_thr), int(max_iters), float(confidence), bool(timing), float(self.debug_extract_delay),
               int(self.cap_bytes))
        if self._pipe is not None and key == self._key:
            return
        self._destroy_native()
        p = _lib.PipelineParams(rows=rows, cols=cols, channels=ch, max_points=int(self.cap), n_slots=int(self.n_slots), extract_workers=int(self.extract_workers),
                                filter_strength=float(filter_strength), homography_method=8, reproj_threshold=float(reproj_thr), max_iters=int(max_iters),
                                confidence=float(confidence), timing=1 if timing else 0, match_lds_cap=int(self.cap_bytes),
                                match_stream=self._masked_stream_handle, debug_extract_delay_ms=float(self.debug_extract_delay) * 1e3)
        h = C.c_void_p()
        m = self.matcher
        check(lib().apds_set_device(self.dev.index or 0))
        check(lib().apds_pipeline_create(C.byref(h), m.rows.data_ptr(), int(m.rows.shape[0]), m.index_base, m.handle, self.db_kp.data_ptr(), self.n_db, C.byref(p)))
        self._pipe, self._key = h, key

    def _destroy_native(self):
        h, self._pipe = self._pipe, None
        if h is not None:
            check(lib().apds_pipeline_destroy(h))

    def close(self):
        """Destroy the native pipeline and the CU-masked match stream (if any)."""
        self._destroy_native()
        h, self._masked_stream_handle = self._masked_stream_handle, None
        if h is not None:
            lib().apds_stream_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    # ---- submit / poll (what bench.py's timed loop calls, through the bound C functions directly) -----------------------------------------
    def prepare(self, frame_shape, filter_strength=0.8, reproj_thr=3.0, max_iters=2000, confidence=0.995, timing=False):
        """Create (or keep) the native pipeline for frames of this shape and these parameters; returns its handle (void*)."""
        self._ensure(frame_shape, filter_strength, reproj_thr, max_iters, confidence, timing)
        return self._pipe

    @staticmethod
    def frame_args(f):
        """(pointer, row stride in bytes, on_device) of a [H, W(, C)] u8 tensor: a device tensor, or a host tensor (pinned: overlapped upload)."""
        return C.c_void_p(f.data_ptr()), int(f.stride(0)), 1 if f.is_cuda else 0

    @staticmethod
    def result_dict(r):
        return dict(n_keypoints=r.n_keypoints, n_matches=r.n_matches, n_inliers=r.n_inliers, status=r.status,
                    H=np.array(r.H, np.float64).reshape(3, 3) if r.homography_found else None)

    def stats(self, reset=False):
        st = _lib.PipelineCounters()
        check(lib().apds_pipeline_stats(self._pipe, C.byref(st), 1 if reset else 0))
        return st

    def run(self, frames, count, filter_strength=0.8, reproj_thr=3.0, max_iters=2000, confidence=0.995, timing=False):
        """Push `count` frames (cycled from `frames`) through the pipeline. Returns (results in frame order, timers)."""
        L = lib()
        self._ensure(frames[0].shape, filter_strength, reproj_thr, max_iters, confidence, timing)
        if timing:
            self.stats(reset=True)
        args = [self.frame_args(f) for f in frames]
        res = _lib.FrameResult()
        results = []
        for i in range(count):
            ptr, stride, on_dev = args[i % len(args)]
            check(L.apds_pipeline_submit(self._pipe, ptr, stride, on_dev, None))
            while True:       # take what is finished, so the result queue stays short on long runs
                rc = L.apds_pipeline_poll(self._pipe, C.byref(res), 0)
                if rc != 0:
                    if rc < 0:
                        check(rc)
                    break
                results.append(self.result_dict(res))
        while len(results) < count:
            check(L.apds_pipeline_poll(self._pipe, C.byref(res), 1))
            results.append(self.result_dict(res))
        for i, r in enumerate(results):
            if r["status"] != 0:
                raise _lib.ApdsError(r["status"], f"frame {i} failed in the pipeline")
        st = self.stats(reset=timing)
        timers = {}
        if timing:
            timers = {"hamming_topk": (st.hamming_topk_ms, st.hamming_topk_launches), "hamming_topk_sample": (st.hamming_topk_sample_ms, st.hamming_topk_sample_launches),
                      "akaze_extract": (st.akaze_extract_ms, st.akaze_extract_calls), "ransac_score": (st.ransac_score_ms, st.ransac_score_launches)}
        self.gap_log = [float(st.match_gaps_first_ms[i]) for i in range(min(16, st.match_gaps))]
        self.gap_mean = float(st.match_gap_mean_ms) if st.match_gaps else None
        if st.match_lds_cap_set_at_frame >= 0 and not self.cap_events:
            self.cap_events.append(dict(frame=int(st.match_lds_cap_set_at_frame), gaps_ms=[round(float(g), 2) for g in st.match_lds_cap_gaps_ms], previous=0))
            self.cap_bytes = int(st.match_lds_cap_bytes)
            self._key = self._key[:-1] + (int(self.cap_bytes),)     # the live native pipeline IS capped: no re-creation for that
        return results, timers
