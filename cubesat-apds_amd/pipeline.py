"""Device-resident composition of the hot path and its multi-GPU form.

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
import time

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
      * a "nccl" group: the library's RCCL transport (its own communicator; rank 0's id travels through the torch group once);
      * a "gloo" group: the host-callback transport over that group (several ranks on one GPU: rehearsal).
    The three steps are separate calls (gather_queries / scan_gathered / exchange_merge) so that a pipeline can issue frame i+1's query
    gather on another stream BEFORE frame i's key exchange. All collective calls must come from ONE thread in the same order on every
    rank. No per-frame allocations."""

    def __init__(self, local_rows64, index_base, group=None, pad_rows=32768, meta_group=None, kmax=2, always_exchange=False):
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
        if backend == "nccl":
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
    """The same path as FramePipeline, software-pipelined over a stream of frames: three host threads, each with its own
    HIP stream and device workspace (the C ABI is re-entrant per thread): extract | match | ratio filter + point gather +
    homography. Frame i+1 is extracted (HBM-bound stencils, raised wave priority) while frame i is matched
    (integer-VALU-bound), and frame i-1's filter/RANSAC host round trips hide behind both. The match stage never
    synchronises with the host (single GPU), so match kernels of consecutive frames queue back to back. Stages hand
    over through HIP events; results come back in frame order."""

    def __init__(self, db_rows64, db_xy, index_base=0, group=None, max_points=(1 << 18) - 1, device="cuda:0", slots=6, reserve_cus=0,
                 n_cus=256, meta_group=None, extract_workers=None):
        import os
        import queue
        self.queue = queue
        # Extraction is ~100 short launches and ~8 count read-backs per frame: 3 ms of GPU work whose WALL time, once the
        # match kernel owns every CU, is set by dispatch and host round-trip latency (measured 8-33 ms depending on the
        # box). Two workers extract alternate frames on two streams, so a frame may take up to two match periods of wall
        # time before extraction paces the pipeline. Each worker owns its share of the slots (a shared pool would let one
        # worker run ahead with every slot while the ordering thread waits for the other's frame).
        self.extract_workers = max(1, int(extract_workers if extract_workers is not None else os.environ.get("APDS_EXTRACT_WORKERS", "2")))
        slots = max(slots, 2 * self.extract_workers)
        self.dev = torch.device(device)
        self.matcher = ShardedMatcher(db_rows64, index_base, group, meta_group=meta_group)
        self._masked_stream_handle = None
        self.cap_bytes = 0                # occupancy cap this pipeline runs its scans with (set by the starvation watch)
        self.n_db = db_xy.shape[0]
        kp = torch.zeros((self.n_db, 7), dtype=torch.float32, device=self.dev)
        kp[:, 0:2] = db_xy
        self.db_kp = kp
        self.cap = max_points
        self.slots = []
        for _ in range(slots):
            s = dict(kps=torch.empty((self.cap, 7), dtype=torch.float32, device=self.dev),
                     desc=torch.empty((self.cap, 64), dtype=torch.uint8, device=self.dev),
                     matches=torch.empty((self.cap, 4), dtype=torch.int32, device=self.dev),
                     p1=torch.empty((self.cap, 2), dtype=torch.float32, device=self.dev),
                     p2=torch.empty((self.cap, 2), dtype=torch.float32, device=self.dev),
                     mask=torch.empty(self.cap, dtype=torch.uint8, device=self.dev),
                     keys=torch.empty((self.cap, 2), dtype=torch.int64, device=self.dev), keys_view=None,
                     ev_extract=torch.cuda.Event(), ev_match=torch.cuda.Event(), K=0, M=0, index=0,
                     ev_mstart=torch.cuda.Event(enable_timing=True), ev_mend=torch.cuda.Event(enable_timing=True),
                     ev_pre=torch.cuda.Event(), ev_scan=torch.cuda.Event(), topk_state=None,
                     owner=len(self.slots) % self.extract_workers,
                     gq=self.matcher.make_buffers(self.cap) if self.matcher.world > 1 else None)
            self.slots.append(s)
        # the match kernel alone fills every CU for ~30 ms; the short extraction / homography kernels get the high-priority
        # queues so that their blocks are dispatched as soon as match workgroups retire
        hp = int(os.environ.get("APDS_PIPE_PRIO", "-1"))      # priority of the short-kernel streams (extraction, homography, query gather)
        self.streams = [torch.cuda.Stream(self.dev, priority=hp), torch.cuda.Stream(self.dev, priority=0), torch.cuda.Stream(self.dev, priority=hp)]
        # Optional (APDS_MATCH_WORKERS=2, single GPU only): two match workers alternate frames on two streams, so the short,
        # poorly filled phases of one frame's match (threshold pre-pass, merges, grid tail) run under the other frame's
        # main kernel: +1 % frames/s, but per-launch kernel times then overlap and no longer read as kernel efficiency,
        # so the default is one worker. With a sharded DB the collectives must be issued in frame order by one thread.
        self.extract_streams = [self.streams[0]] + [torch.cuda.Stream(self.dev, priority=hp) for _ in range(self.extract_workers - 1)]
        # 1: cap the match kernel's occupancy when the match stream is seen starving (see match_worker); APDS_ADAPTIVE_CAP=0 turns it off
        self.adaptive_cap = os.environ.get("APDS_ADAPTIVE_CAP", "1") != "0"
        self.cap_events = []
        self.gap_log = []                 # idle time of the match stream before each frame's match (ms), for diagnosis
        self.debug_extract_delay = float(os.environ.get("APDS_DEBUG_EXTRACT_DELAY_MS", "0")) * 1e-3
        self.match_workers = 2 if (group is None and os.environ.get("APDS_MATCH_WORKERS", "1") == "2") else 1
        self.gather_stream = torch.cuda.Stream(self.dev, priority=hp) if self.matcher.world > 1 else None
        # One GPU: the scan of a frame is issued as three launches groups on three streams (apds_dev_topk_prepass / _scan / _merge on a
        # per-slot state object): frame i + 1's threshold pre-pass and frame i - 1's record merge run beside frame i's main scan, so the
        # main scans follow each other on their stream without the ~0.8 ms of pre-pass, merge and dependent-launch gaps between them.
        # Per-launch event timing stays on the main kernel only (its stream carries nothing else). APDS_MATCH_SPLIT=0: one call per frame.
        self.split_match = self.matcher.world == 1 and os.environ.get("APDS_MATCH_SPLIT", "1") != "0" and self.match_workers == 1
        if self.split_match:
            self.pre_stream = torch.cuda.Stream(self.dev, priority=0)
            self.merge_stream = torch.cuda.Stream(self.dev, priority=hp)
            for s in self.slots:
                h = C.c_void_p()
                check(lib().apds_dev_topk_state_create(C.byref(h)))
                s["topk_state"] = h
        if reserve_cus > 0:
            # keep `reserve_cus` CUs (spread evenly over the CU index space) out of the MATCH stream only
            words = (n_cus + 31) // 32
            mask = np.full(words, 0xFFFFFFFF, np.uint32)
            import os
            layout = os.environ.get("APDS_CU_MASK_LAYOUT", "spread")
            if layout == "tail":
                cus = range(n_cus - reserve_cus, n_cus)
            elif layout == "head":
                cus = range(reserve_cus)
            elif layout == "wordtop":      # the top bits of every 32-bit word
                per = max(1, reserve_cus // words)
                cus = [wd * 32 + 31 - b for wd in range(words) for b in range(per)]
            else:
                stride = n_cus // reserve_cus
                cus = [r * stride for r in range(reserve_cus)]
            for cu in cus:
                mask[cu // 32] &= ~np.uint32(1 << (cu % 32))
            h = C.c_void_p()
            check(lib().apds_stream_create(0, _lib.ptr(mask), words, C.byref(h)))
            self._masked_stream_handle = h
            self.streams[1] = torch.cuda.ExternalStream(h.value, device=self.dev)
        # built AFTER the CU-mask block: the match workers launch on these, so the masked stream must already be in place
        self.match_streams = [self.streams[1]] + [torch.cuda.Stream(self.dev, priority=0) for _ in range(self.match_workers - 1)]
        torch.cuda.synchronize()

    def close(self):
        """Destroy the CU-masked match stream (if any). The pipeline must be idle."""
        h, self._masked_stream_handle = self._masked_stream_handle, None
        if h is not None:
            torch.cuda.synchronize()
            lib().apds_stream_destroy(h)
        for s in getattr(self, "slots", []):
            st, s["topk_state"] = s.get("topk_state"), None
            if st:
                lib().apds_dev_topk_state_destroy(st)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, frames, count, filter_strength=0.8, reproj_thr=3.0, max_iters=2000, confidence=0.995, timing=False):
        """Push `count` frames (cycled from `frames`) through the three stages. Returns (results in frame order, timers)."""
        import threading
        L = lib()
        q1, q2 = self.queue.Queue(), self.queue.Queue()
        E = self.extract_workers
        q_free = [self.queue.Queue() for _ in range(E)]      # per extraction worker: its own slots
        q_ext = [self.queue.Queue() for _ in range(E)]       # per extraction worker: its finished frames, in its order
        q_any = self.queue.Queue()                           # tokens: which worker finished a frame (arrival order)
        for s in self.slots:
            q_free[s["owner"]].put(s)
        results = [None] * count
        timers, errors = {}, []
        dev_index = self.dev.index or 0
        # The occupancy cap of the scan is process-wide in the library; a pipeline that decided to cap (starvation watch) applies
        # its cap for the duration of its own runs only and puts the previous value back when the run ends.
        cap_restore = [None]
        if self.cap_bytes:
            old = C.c_int(0)
            check(L.apds_dev_match_lds_cap(self.cap_bytes, C.byref(old)))
            cap_restore[0] = old.value

        def guarded(fn):
            def wrap():
                try:
                    torch.cuda.set_device(dev_index)
                    check(L.apds_set_device(dev_index))
                    check(L.apds_dev_timing_enable(1 if timing else 0))
                    fn()
                except BaseException as e:   # surface worker failures instead of dead-locking the queues
                    errors.append(e)
                    q1.put(None)
                    q2.put(None)
                    for q in q_ext + q_free + [q_any]:
                        q.put(None)
                finally:
                    try:
                        L.apds_dev_timing_enable(0)
                        torch.cuda.synchronize()
                        # this worker's stream + device workspace (threads are per run(); long-lived stage threads that keep their
                        # workspaces across runs measured 0.5 ms/frame SLOWER, reproducibly, so each run starts from fresh ones)
                        L.apds_thread_release()
                    except Exception:
                        pass
            return wrap

        def collect(names):
            if timing:
                for n in names:
                    ms, k = _lib.kernel_ms(n)
                    timers[n] = (ms, k)

        timer_lock = threading.Lock()

        def make_extract_worker(e):
            def extract_worker():
                stream = self.extract_streams[e]
                with torch.cuda.stream(stream):
                    for i in range(e, count, E):
                        s = q_free[e].get()
                        if s is None:
                            return
                        f = frames[i % len(frames)]
                        if not f.is_cuda:
                            # a host frame (pinned memory): upload it on THIS worker's stream in front of the extraction, into the slot's own
                            # device buffer; the other extraction worker and the match run meanwhile, so the PCIe copy is overlapped
                            if s.get("frame") is None or s["frame"].shape != f.shape:
                                s["frame"] = torch.empty(f.shape, dtype=f.dtype, device=self.dev)
                            s["frame"].copy_(f, non_blocking=True)
                            f = s["frame"]
                        ch = 1 if f.dim() == 2 else f.shape[2]
                        n = C.c_int(0)
                        check(L.apds_dev_akaze_extract(f.data_ptr(), f.shape[0], f.shape[1], ch, f.stride(0), self.cap, s["kps"].data_ptr(),
                                                       s["desc"].data_ptr(), self.cap, C.byref(n), torch_stream()))
                        s["K"], s["index"] = n.value, i
                        s["ev_extract"].record(stream)
                        if self.debug_extract_delay > 0:       # test hook: emulate a box on which extraction cannot keep up
                            time.sleep(self.debug_extract_delay)
                        q_ext[e].put(s)
                        q_any.put(e)
                    if timing:
                        ms, k = _lib.kernel_ms("akaze_extract")
                        with timer_lock:
                            old = timers.get("akaze_extract", (0.0, 0))
                            timers["akaze_extract"] = (old[0] + ms, old[1] + k)
            return extract_worker

        def order_worker():
            # Sharded DB: frames back into order; the host-side count exchange (one gloo collective per frame, which every rank
            # must issue in the same order) happens here, so the match thread stays free of host synchronisation.
            # One GPU: no collective, so nothing requires frame order on the match stream (results are stored by frame index):
            # frames go to the match first come, first served. One of the two extraction workers tends to finish just after
            # a match ends (its last kernels only run freely once the match kernel is gone); in frame order the match stream then
            # waited ~1.2 ms for every second frame, now it takes the other worker's next frame, which is already there.
            if self.matcher.world > 1:
                for i in range(count):
                    s = q_ext[i % E].get()
                    if s is None:
                        return
                    s["counts"] = self.matcher.exchange_counts(s["K"])
                    q1.put(s)
                q1.put(None)
                return
            for _ in range(count):
                e = q_any.get()            # a worker's token: its next finished frame is in q_ext[e]
                if e is None:
                    return
                s = q_ext[e].get()
                if s is None:
                    return
                s["counts"] = [int(s["K"])]
                q1.put(s)
            q1.put(None)

        done_lock = threading.Lock()
        alive = [self.match_workers]

        def sharded_match_worker():
            # Sharded DB: this thread issues ALL collectives of the run, in the same order on every rank: gather(i+1) [its own
            # stream, waits only for frame i+1's extraction] BEFORE exchange(i) [match stream, after scan(i)], so the query
            # all-gather of the next frame travels under the current frame's scan. Frame i's scan is launched before the thread
            # blocks on frame i+1, so the GPU never waits for the host here.
            stream, m, prev = self.match_streams[0], self.matcher, None
            with torch.cuda.stream(stream):
                while True:
                    s = q1.get()
                    if s is not None and s["counts"] is not None:
                        with torch.cuda.stream(self.gather_stream):
                            self.gather_stream.wait_event(s["ev_extract"])
                            m.gather_queries(s["desc"][:s["K"]], s["counts"], s["gq"])
                    if prev is not None:
                        prev["keys_view"] = m.exchange_merge(prev["gq"], 2, out=prev["keys"])
                        prev["ev_match"].record(stream)
                        q2.put(prev)
                        prev = None
                    if s is None:
                        break
                    if s["counts"] is None:      # no host-side count exchange available: one-call form (synchronises the stream)
                        stream.wait_event(s["ev_extract"])
                        s["keys_view"] = m.knn(s["desc"][:s["K"]], 2, out=s["keys"])
                        s["ev_match"].record(stream)
                        q2.put(s)
                        continue
                    m.scan_gathered(s["gq"], 2)
                    prev = s
                if timing:
                    for n in ("hamming_topk", "hamming_topk_sample"):
                        timers[n] = _lib.kernel_ms(n)
                q2.put(None)

        def make_match_worker(stream):
            def match_worker():
                # Starvation watch (single match worker): the match stream should never wait for a frame. On some boxes the short
                # extraction kernels are dispatched so late under the match kernel, which owns every wave slot, that extraction
                # paces the pipeline (36 ms per frame instead of 30, seen on about one box in eight). The gap on the match stream
                # between one frame's last match kernel and the next frame's first is measured with events; if at least three of
                # six consecutive gaps exceed 4 ms (the healthy pattern is 0.01 / 1.2 ms alternating) the match kernel's occupancy is capped at two workgroups per CU (apds_dev_match_lds_cap), which leaves
                # wave slots free for the other stages at ~1.5 % of match throughput.
                watch = self.adaptive_cap and self.match_workers == 1 and self.cap_bytes == 0
                pending, gaps, prev = [], [], None
                with torch.cuda.stream(stream):
                    while True:
                        s = q1.get()
                        if s is None:
                            q1.put(None)            # let the other match worker see the end marker too
                            break
                        if self.split_match and s["K"] > 0:
                            m, K = self.matcher, s["K"]
                            with torch.cuda.stream(self.pre_stream):        # threshold pre-pass: beside the previous frame's main scan
                                self.pre_stream.wait_event(s["ev_extract"])
                                check(L.apds_dev_topk_prepass(s["topk_state"], s["desc"].data_ptr(), K, m.rows.data_ptr(), int(m.rows.shape[0]), m.index_base, 2,
                                                              torch_stream()))
                                s["ev_pre"].record(self.pre_stream)
                            stream.wait_event(s["ev_pre"])
                            if watch:
                                s["ev_mstart"].record(stream)
                            check(L.apds_dev_topk_scan(s["topk_state"], s["desc"].data_ptr(), m.rows.data_ptr(), torch_stream()))
                            s["ev_scan"].record(stream)
                            if watch:
                                s["ev_mend"].record(stream)
                            with torch.cuda.stream(self.merge_stream):      # record merge: beside the next frame's main scan
                                self.merge_stream.wait_event(s["ev_scan"])
                                check(L.apds_dev_topk_merge(s["topk_state"], m.index_base, s["keys"].data_ptr(), torch_stream()))
                                s["ev_match"].record(self.merge_stream)
                            s["keys_view"] = s["keys"][:K]
                        else:
                            stream.wait_event(s["ev_extract"])
                            if watch:
                                s["ev_mstart"].record(stream)
                            s["keys_view"] = self.matcher.knn(s["desc"][:s["K"]], 2, out=s["keys"], counts=s["counts"])
                            s["ev_match"].record(stream)
                            if watch:
                                s["ev_mend"].record(stream)
                        if watch:
                            if prev is not None and s["index"] >= 4:           # the first frames are the pipeline filling up
                                pending.append((prev["ev_mend"], s["ev_mstart"]))
                            prev = s
                            while pending and pending[0][1].query():       # both recorded before it, both complete
                                a, b = pending.pop(0)
                                try:
                                    g = a.elapsed_time(b)
                                except RuntimeError:                        # an event was re-recorded in the meantime: skip the sample
                                    continue
                                if 0.0 <= g < 1000.0:
                                    gaps.append(g)
                                    self.gap_log.append(g)
                            # with two extraction workers late frames arrive in pairs (gaps alternate long / short), and one long
                            # stall (an allocation, a page fault storm) is not starvation: at least three of the last six gaps
                            if len(gaps) >= 6 and sum(1 for g in gaps[-6:] if g > 4.0) >= 3:
                                old = C.c_int(0)
                                check(L.apds_dev_match_lds_cap(55000, C.byref(old)))
                                self.cap_bytes = 55000                 # this pipeline's later runs start capped; run() restores the
                                cap_restore[0] = old.value if cap_restore[0] is None else cap_restore[0]   # process-wide value when it ends
                                self.cap_events.append(dict(frame=s["index"], gaps_ms=[round(g, 2) for g in gaps[-6:]], previous=old.value))
                                watch = False
                        q2.put(s)
                    with done_lock:
                        alive[0] -= 1
                        last = alive[0] == 0
                    if timing:
                        for n in ("hamming_topk", "hamming_topk_sample"):
                            ms, k = _lib.kernel_ms(n)
                            with done_lock:
                                old = timers.get(n, (0.0, 0))
                                timers[n] = (old[0] + ms, old[1] + k)
                    if last:
                        q2.put(None)
            return match_worker

        def homography_worker():
            with torch.cuda.stream(self.streams[2]):
                while True:
                    s = q2.get()
                    if s is None:
                        break
                    self.streams[2].wait_event(s["ev_match"])
                    K, M = s["K"], 0
                    if K > 0:
                        nm = C.c_int(0)
                        check(L.apds_dev_ratio_filter(s["keys_view"].data_ptr(), K, 2, float(filter_strength), s["matches"].data_ptr(), C.byref(nm),
                                                      torch_stream()))
                        M = nm.value
                    out = dict(n_keypoints=K, n_matches=M, H=None, n_inliers=0)
                    if M >= 4:
                        check(L.apds_dev_points_from_matches(s["kps"].data_ptr(), K, self.db_kp.data_ptr(), self.n_db, s["matches"].data_ptr(), M, 0,
                                                             s["p1"].data_ptr(), s["p2"].data_ptr(), torch_stream()))
                        H = np.zeros(9, np.float64)
                        rc = L.apds_dev_find_homography(s["p1"].data_ptr(), s["p2"].data_ptr(), M, 8, float(reproj_thr), int(max_iters),
                                                        float(confidence), _lib.ptr(H), s["mask"].data_ptr(), torch_stream())
                        if rc == 0:
                            out["H"] = H.reshape(3, 3)
                            out["n_inliers"] = int(s["mask"][:M].sum().item())
                        elif rc != _lib.ERR_EMPTY:
                            check(rc)
                    results[s["index"]] = out
                    q_free[s["owner"]].put(s)
                collect(["ransac_score"])

        match_stage = [sharded_match_worker] if self.matcher.world > 1 else [make_match_worker(st) for st in self.match_streams]
        workers = [make_extract_worker(e) for e in range(E)] + [order_worker] + match_stage + [homography_worker]
        threads = [threading.Thread(target=guarded(f), daemon=True) for f in workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if cap_restore[0] is not None:
            check(L.apds_dev_match_lds_cap(cap_restore[0], None))
        if errors:
            raise errors[0]
        return results, timers
