"""Device-resident composition of the hot path and its multi-GPU form.

frame (u8, HBM) -> AKAZE -> descriptors -> Hamming top-2 against a resident descriptor DB -> ratio test ->
matched points -> RANSAC homography.  The reference only chains these steps inside unit tests
(/root/reference/feature_extraction/src/lib.rs:197-249) and never calls find_homography_mat on the result; this
module is the composed pipeline the north-star metric (frames/s) is measured on.

Multi-GPU (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm):
  * the descriptor DB is row-sharded: rank r holds rows [base_r, base_r + n_r) resident in its HBM;
  * frames are data-parallel: every rank extracts its own frame;
  * the only exchange is in the match step: all-gather of the ranks' query descriptors (padded to a fixed row
    count), local top-k of ALL queries against the local shard, all-gather of the per-shard top-k keys
    (packed u64 = distance << 32 | global row), then each rank merges the candidates of its own queries.
    u64 min-merge reproduces the single-GPU result exactly, including the lowest-index tie break.
torch is used for device buffers, streams and the collectives only; every compute step is a libapds_hip kernel.
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


class HipBackend:
    """Local compute through the C ABI (device pointers of torch tensors, launched on torch's current stream)."""

    def topk(self, q_rows64, train_rows64, index_base, k, out=None):
        nq, nt = q_rows64.shape[0], train_rows64.shape[0]
        if out is None:
            out = torch.empty((nq, k), dtype=torch.int64, device=q_rows64.device)
        check(lib().apds_dev_hamming_topk(q_rows64.data_ptr(), nq, train_rows64.data_ptr(), nt, int(index_base), k, out.data_ptr(), torch_stream()))
        return out

    def merge(self, parts, k, out=None):
        """parts: [P, Q, k] int64 (contiguous) -> [Q, k]"""
        p, q = parts.shape[0], parts.shape[1]
        if out is None:
            out = torch.empty((q, k), dtype=torch.int64, device=parts.device)
        check(lib().apds_dev_merge_topk(parts.data_ptr(), p, q, k, out.data_ptr(), torch_stream()))
        return out


def _gather_into(dist, group, dst, src):
    """all_gather_into_tensor that also works for the gloo rehearsal backend with device tensors (staged through the host)."""
    if dist.get_backend(group) == "gloo" and src.is_cuda:
        d, s = torch.empty(dst.shape, dtype=dst.dtype), src.cpu()
        dist.all_gather_into_tensor(d.view(-1), s.view(-1), group=group)
        dst.copy_(d)
    else:
        dist.all_gather_into_tensor(dst.view(-1), src.view(-1), group=group)


def _all_to_all_into(dist, group, dst, src, out_splits, in_splits):
    """all_to_all_single on flat tensors (gloo rehearsal with device tensors: staged through the host)."""
    if dist.get_backend(group) == "gloo" and src.is_cuda:
        d, s = torch.empty(dst.shape, dtype=dst.dtype), src.cpu()
        dist.all_to_all_single(d, s, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        dst.copy_(d)
    else:
        dist.all_to_all_single(dst, src, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


class GatheredQueries:
    """One frame's query exchange buffers (allocated once, reused every frame that goes through the same slot):
    `mine` = this rank's rows padded to the common row count, `gathered` = every rank's padded rows, `all_q` = the ranks' rows
    back to back without padding (what the local shard is scanned with), `local` = this shard's top-k of all of them,
    `recv` = every shard's top-k of THIS rank's queries, `merged` = their u64-min merge."""

    def __init__(self, world, pad, kmax, device):
        self.world, self.pad, self.kmax = world, pad, kmax          # pad = capacity in rows per rank
        self.mine = torch.zeros((pad, 64), dtype=torch.uint8, device=device)
        self.gathered = torch.empty(world * pad * 64, dtype=torch.uint8, device=device)
        self.all_q = torch.empty((world * pad, 64), dtype=torch.uint8, device=device)
        self.local = torch.empty(world * pad * kmax, dtype=torch.int64, device=device)
        self.recv = torch.empty(world * pad * kmax, dtype=torch.int64, device=device)
        self.merged = torch.empty(pad * kmax, dtype=torch.int64, device=device)
        self.counts, self.total, self.nq = None, 0, 0
        self.event = torch.cuda.Event() if torch.device(device).type == "cuda" else None


class ShardedMatcher:
    """Hamming k-NN of per-rank query sets against a row-sharded resident DB.

    Per frame: (1) all-gather of the ranks' query rows (fixed pad, so the message size never changes); (2) local top-k of ALL
    queries against the local shard, keys carry global row indices; (3) all-to-all of the keys: rank r sends rank s the
    [counts[s], k] block of s's queries and receives [counts[r], k] from every shard (1/world of what an all-gather of the
    keys moves); (4) u64-min merge of the `world` candidate lists per query = the single-GPU result, lowest-index tie-break
    included. The three steps are separate calls (gather_queries / scan_gathered / exchange_merge) so that a pipeline can issue
    frame i+1's query gather on another stream BEFORE frame i's key exchange: it then runs under frame i's scan. All collectives
    use ONE communicator and must be issued by ONE thread in the same order on every rank (torch serialises them on the process
    group's internal stream; two communicators driven from two threads could be launched in different orders on different ranks
    and dead-lock). No per-frame allocations."""

    def __init__(self, local_rows64, index_base, group=None, backend=None, pad_rows=32768, meta_group=None, kmax=2, always_exchange=False):
        self.rows = local_rows64
        self.index_base = int(index_base)
        self.group = group
        self.meta_group = meta_group      # optional host-side (gloo) group for the per-frame query counts
        self.backend = backend or HipBackend()
        self.pad_rows = pad_rows          # smallest buffer capacity (rows per rank)
        self.msg_round = 1024             # the query all-gather's row count is the frame's largest count rounded up to this
        self.kmax = kmax
        self.always_exchange = always_exchange   # run the collectives even with one rank (RCCL self-test on a one-GPU box)
        self._own = None                  # buffers of the plain knn() form
        if group is not None:
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.dist, self.world, self.rank = None, 1, 0

    def exchange_counts(self, nq):
        """Every rank's query count for one frame, as host ints, through the host-side group: no device work and no stream
        synchronisation, so the thread that later issues the match keeps queueing kernels ahead of the GPU. Must be called
        once per frame, in frame order, by the same thread on every rank."""
        if self.world == 1 and not self.always_exchange:
            return [int(nq)]
        if self.meta_group is None:
            return None
        t = torch.tensor([int(nq)], dtype=torch.int64)
        o = torch.empty(self.world, dtype=torch.int64)
        self.dist.all_gather_into_tensor(o, t, group=self.meta_group)
        return [int(v) for v in o.tolist()]

    def make_buffers(self, max_queries=None, k=None):
        """Exchange buffers for frames of at most `max_queries` queries per rank (one set per frame in flight)."""
        pad = max(self.pad_rows, int(max_queries or 0))
        pad = -(-pad // self.msg_round) * self.msg_round          # whole messages: gather_queries sends multiples of msg_round rows
        return GatheredQueries(self.world, pad, k or self.kmax, self.rows.device)

    def gather_queries(self, q_rows64, counts, buf):
        """Step (1) on the CURRENT stream, into `buf`: after it `buf.all_q[:buf.total]` holds every rank's queries."""
        nq = q_rows64.shape[0]
        assert counts[self.rank] == nq and max(counts) <= buf.pad and buf.world == self.world
        buf.counts, buf.total, buf.nq = list(counts), int(sum(counts)), nq
        buf.mine[:nq].copy_(q_rows64)
        # message size of this frame: the largest count, rounded up (the buffers are sized for the capacity, the wire is not)
        # a function of `counts` alone (identical on every rank, whatever each rank's buffer capacity is): a message size taken from the
        # rank-local buffer would make the collective's sizes differ between ranks, which RCCL answers with a hang, not an error
        rows = max(self.msg_round, -(-max(counts) // self.msg_round) * self.msg_round)
        assert rows <= buf.pad, f"exchange buffer holds {buf.pad} rows per rank, this frame needs {rows}: make_buffers(max_queries) rounds up to {self.msg_round}"
        gathered = buf.gathered[:self.world * rows * 64].view(self.world, rows, 64)
        _gather_into(self.dist, self.group, gathered, buf.mine[:rows])
        off = 0
        for r, c in enumerate(counts):          # drop the padding: `world` slice copies into the preallocated block
            if c:
                buf.all_q[off:off + c].copy_(gathered[r, :c])
            off += c
        if buf.event is not None:
            buf.event.record()
        return buf

    def scan_gathered(self, buf, k=2):
        """Step (2) on the CURRENT stream: this shard's top-k of every gathered query (no collective)."""
        assert k <= buf.kmax
        if buf.event is not None:
            torch.cuda.current_stream().wait_event(buf.event)
        if buf.total:
            self.backend.topk(buf.all_q[:buf.total], self.rows, self.index_base, k, out=buf.local[:buf.total * k].view(buf.total, k))

    def exchange_merge(self, buf, k=2, out=None):
        """Steps (3)-(4) on the CURRENT stream (the one scan_gathered ran on): all-to-all of the keys, then the per-query merge."""
        world, nq, total = self.world, buf.nq, buf.total
        recv = buf.recv[:world * nq * k]
        _all_to_all_into(self.dist, self.group, recv, buf.local[:total * k], [nq * k] * world, [c * k for c in buf.counts])
        dst = out[:nq] if out is not None else buf.merged[:nq * k].view(nq, k)
        if nq:
            self.backend.merge(recv.view(world, nq, k), k, out=dst)
        return dst

    def match_gathered(self, buf, k=2, out=None):
        """Steps (2)-(4) on the CURRENT stream for a frame whose queries gather_queries() has put into `buf`."""
        self.scan_gathered(buf, k)
        return self.exchange_merge(buf, k, out=out)

    def knn(self, q_rows64, k=2, out=None, counts=None):
        """q_rows64: this rank's queries [Q_r, 64] u8. Returns [Q_r, k] int64 keys over the WHOLE DB. `counts`: every rank's
        query count (exchange_counts); without it the counts are gathered on the device, which costs a host synchronisation.
        Without `out` the result is a view of an internal buffer, valid until the next call."""
        be = self.backend
        if self.world == 1 and not (self.always_exchange and self.dist is not None):
            if out is not None:
                return be.topk(q_rows64, self.rows, self.index_base, k, out=out[:q_rows64.shape[0]])
            return be.topk(q_rows64, self.rows, self.index_base, k)
        dist, dev = self.dist, q_rows64.device
        nq = q_rows64.shape[0]
        if counts is None:
            cnt = torch.zeros(self.world, dtype=torch.int64, device=dev)
            _gather_into(dist, self.group, cnt, torch.tensor([nq], dtype=torch.int64, device=dev))
            counts = [int(c) for c in cnt.tolist()]
        if self._own is None or self._own.pad < max(counts) or self._own.kmax < k:
            self._own = self.make_buffers(max(counts), max(k, self.kmax))
        self.gather_queries(q_rows64, counts, self._own)
        return self.match_gathered(self._own, k, out=out)


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
                        stream.wait_event(s["ev_extract"])
                        if watch:
                            s["ev_mstart"].record(stream)
                        s["keys_view"] = self.matcher.knn(s["desc"][:s["K"]], 2, out=s["keys"], counts=s["counts"])
                        s["ev_match"].record(stream)
                        if watch:
                            s["ev_mend"].record(stream)
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
