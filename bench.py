#!/usr/bin/env python3
"""bench.py — frames/s of the hot path on MI355X: AKAZE detect+describe -> Hamming top-2 vs a resident descriptor
DB -> ratio test -> RANSAC homography, on 4096x4096 synthetic tiles already resident in HBM.

    python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = every rank pushes ONE frame through the whole path. The DB (default 1M rows x 64 B, BASELINE.json's
"4k x 4k tile vs 1M-desc DB") is row-sharded over the ranks; the match step all-gathers queries and per-shard
top-2 keys over RCCL (cubesat-apds_amd/pipeline.py). Per-GPU work per step is constant in N (1 frame detected,
Q_total x N_db/N pairs matched), so scaling is "weak" and value = N frames / step time.

Prints ONE JSON line (rank 0). roofline is for the dominant kernel (hamming_topk); cpu_baseline times the oracle
("port") on the host cores for a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

METRIC = "frames/sec (detect+match+homography) 4k×4k tile vs 1M-desc DB; Mmatches/sec"
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP4_DENSE_PEAK_TFLOPS = 10000.0              # MI355X_MICROARCH.md: FP6/FP4 MFMA ~10 PF dense (the 20 PF spec figure is 2:1 sparse)
VALU_FP32_LANE_RATE_SPEC = 256 * 128 * 2.4e9   # MI355X_MICROARCH.md: SIMD-32 x 4, a wave64 VALU op per 2 cycles = 128 lanes/clk/CU


def detect_algorithmic_bytes(w, h):
    """8*WH + 36*PL + 12*PS + HS for the levels that exist at this size."""
    sizes, lw, lh = [], w, h
    for o in range(4):
        sizes.append((lw, lh))
        lw, lh = lw >> 1, lh >> 1
        if lw < 80 or lh < 40:
            break
    steps = [[0, 3, 3, 4], [4, 5, 6, 7], [8, 10, 12, 14], [17, 20, 24, 29]]
    total = 8.0 * w * h
    for o, (a, b) in enumerate(sizes):
        px = float(a * b)
        total += 36.0 * 4 * px + 12.0 * px * sum(steps[o])
        if o + 1 < len(sizes):
            total += 5.0 * px
    return total


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`
    as a child process (one rank per GPU, rendezvous on 127.0.0.1, a free port), pass rank 0's JSON line through and return the child's exit
    code. The parent never imports torch or touches HIP. If the ranks fail on the library's own RCCL communicator (the default transport) and
    the caller did not pin a transport, a FRESH set of ranks is started with --transport torch (the same exchange through torch.distributed's
    collectives): the first multi-GPU run a node ever gives this code should not be lost to a communicator set-up problem."""
    import socket
    import subprocess

    def run_once(extra):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:] + extra
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        json_lines = []
        for line in proc.stdout:
            if line.lstrip().startswith("{"):          # the contract: ONE JSON line on stdout; anything else a rank printed goes to stderr
                json_lines.append(line)
            else:
                sys.stderr.write(line)
        rc = proc.wait()
        if rc == 0 and len(json_lines) != 1:
            print(f"bench.py: the ranks exited 0 but printed {len(json_lines)} JSON lines", file=sys.stderr)
            rc = 1
        return rc, json_lines

    pinned = any(a == "--transport" or a.startswith("--transport=") for a in sys.argv[1:])
    rc, lines = run_once([])
    if rc != 0 and not pinned:
        print(f"bench.py: the ranks exited {rc} on the default transport (the library's own RCCL communicator); starting a fresh set with --transport torch",
              file=sys.stderr, flush=True)
        rc, lines = run_once(["--transport", "torch", "--transport-fallback-from", f"rccl (exit code {rc})"])
    if rc == 0:
        sys.stdout.write(lines[0])
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--db-rows", type=int, default=1_000_000, help="total descriptor DB rows (sharded over the ranks)")
    ap.add_argument("--frames", type=int, default=2, help="distinct frames per rank, cycled")
    ap.add_argument("--db", choices=["mixed", "real"], default="mixed",
                    help="mixed (default, SURVEY 8d): descriptors of a shifted copy of every frame + i.i.d. random rows. real: EVERY row is an AKAZE "
                         "descriptor of some image (the frames' shifted copies + blended / flipped variants of them), rows shuffled (1 GPU)")
    ap.add_argument("--filter-strength", type=float, default=0.3, help="Lowe ratio (reference test: 0.3, lib.rs:222)")
    ap.add_argument("--workload", choices=["frame", "l2"], default="frame",
                    help="frame: the north-star pipeline (default). l2: BASELINE config 3, float-descriptor L2 match as an MFMA GEMM (1 GPU)")
    ap.add_argument("--l2-queries", type=int, default=1_048_576)
    ap.add_argument("--l2-mode", choices=["exact", "screen"], default="exact",
                    help="exact: every distance in f32 MFMA. screen: bf16 MFMA screen (proved bound) + f32 re-rank of the candidates; same keys")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reserve-cus", type=int, default=0, help="CUs masked out of the match stream so the other stages overlap it")
    ap.add_argument("--serial", action="store_true", help="one frame at a time (no cross-frame overlap of the three stages)")
    ap.add_argument("--host-frames", action="store_true", default=True,
                    help="(default) additionally time the streamed pipeline with every step's frame coming from pinned HOST memory (hipMemcpyAsync on "
                         "the extraction stream): reported as value_host_frames next to the resident `value` (the reference's bench times Mat "
                         "construction + extraction, benchmarks/benches/feature_extraction.rs:35-45)")
    ap.add_argument("--no-host-frames", dest="host_frames", action="store_false", help="skip the host-frames leg")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="N > 1: what carries the match step's exchange. rccl (default): the library's own RCCL communicator (ncclCommInitRank inside "
                         "libapds_hip.so, id broadcast through the torch group). torch: the same all-gather / all-to-all through torch.distributed's "
                         "device collectives on its nccl group (device-callback transport). Started bare, `--gpus N` tries rccl first and starts a "
                         "fresh set of ranks on torch if that set exits non-zero")
    ap.add_argument("--transport-fallback-from", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started bare (`python bench.py --gpus N`): this process becomes the launcher. Nothing has touched the GPU yet (no torch import,
        # no HIP call), the ranks are CHILD processes (never exec from a process that initialised the GPU), rank 0's JSON line is relayed.
        return spawn_ranks(args.gpus)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it bare, or with torch.distributed.run --nproc-per-node {args.gpus})")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("APDS_BENCH_BACKEND", "nccl")      # "gloo" = single-GPU rehearsal of the multi-rank path
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    group = meta_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."); stdout carries the one JSON line of the
        # contract, so file descriptor 1 points at stderr while the process groups come up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            group = dist.group.WORLD
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    pkg = graft.load_package()
    from cubesat_apds_amd import pipeline as pl
    L = pkg.lib()
    check = pkg._lib.check
    check(L.apds_set_device(dev_index))
    if args.workload == "l2":
        return bench_l2(args, pkg, pl, torch, dev, world)
    synth = pkg.synth
    T, NDB = args.tile, args.db_rows

    # ---- setup (untimed): frames, and a DB holding the descriptors of a shifted copy of every frame + random rows
    shift = (37, 52)   # (dy, dx): DB images are np.roll'ed frames, so the true homography is a translation
    frames_np = [synth.make_tile(T, T, frame_index=rank * args.frames + i) for i in range(args.frames)]
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]
    cap = pkg.feature_extraction.MAX_POINTS
    kps = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
    planted_rows, planted_xy = [], []
    setup_stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(setup_stream)      # everything below (kernels, torch ops, collectives) is ordered on this stream
    for f in frames_np:
        rolled = torch.from_numpy(np.roll(f, shift, axis=(0, 1)).copy()).to(dev)
        n = C.c_int(0)
        check(L.apds_dev_akaze_extract(rolled.data_ptr(), T, T, rolled.shape[2], rolled.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n),
                                       pl.torch_stream()))
        planted_rows.append(desc[:n.value].clone())
        planted_xy.append(kps[:n.value, 0:2].clone())
    mine_rows, mine_xy = torch.cat(planted_rows), torch.cat(planted_xy)
    if world > 1:
        cnt_t = torch.zeros(world, dtype=torch.int64, device=dev)
        pl._gather_into(dist, group, cnt_t, torch.tensor([mine_rows.shape[0]], dtype=torch.int64, device=dev))
        cnt = [int(c) for c in cnt_t.tolist()]
        padn = max(cnt)
        br = torch.zeros((padn, 64), dtype=torch.uint8, device=dev)
        bx = torch.zeros((padn, 2), dtype=torch.float32, device=dev)
        br[:mine_rows.shape[0]], bx[:mine_xy.shape[0]] = mine_rows, mine_xy
        gr = torch.empty((world, padn, 64), dtype=torch.uint8, device=dev)
        gx = torch.empty((world, padn, 2), dtype=torch.float32, device=dev)
        pl._gather_into(dist, group, gr, br)
        pl._gather_into(dist, group, gx, bx)
        all_rows = torch.cat([gr[r, :cnt[r]] for r in range(world)])
        all_xy = torch.cat([gx[r, :cnt[r]] for r in range(world)])
    else:
        all_rows, all_xy = mine_rows, mine_xy
    P = min(all_rows.shape[0], NDB)
    real_variants = 0
    # global DB = [P planted rows | NDB - P random rows]; this rank keeps rows [lo, hi)
    lo, hi = rank * NDB // world, (rank + 1) * NDB // world
    parts = []
    if lo < P:
        parts.append(all_rows[lo:min(hi, P)])
    if hi > P and args.db == "real":
        # the rest of the DB = real AKAZE descriptors of OTHER images: blends of the frames with rolled / flipped copies of each other
        # (new local structure, hence new descriptors; generated on the GPU, the CPU tile generator takes 15 s per 4096^2 tile)
        assert world == 1, "--db real is a single-GPU configuration"
        need, v = hi - max(lo, P), 0
        base = [f.to(torch.float32) for f in frames]
        while need > 0:
            a, b = base[v % len(base)], base[(v + 1) % len(base)]
            b = torch.roll(b, shifts=(137 * v + 50, 211 * v + 70), dims=(0, 1))
            if v & 1:
                b = torch.flip(b, dims=(0,))
            if v & 2:
                b = torch.flip(b, dims=(1,))
            if v & 4:
                a = torch.flip(a, dims=(0, 1))
            w = 0.35 + 0.05 * (v % 7)
            img = (a * w + b * (1.0 - w)).round().clamp(0, 255).to(torch.uint8).contiguous()
            img[..., 3:] = 255
            n = C.c_int(0)
            check(L.apds_dev_akaze_extract(img.data_ptr(), T, T, img.shape[2], img.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n),
                                           pl.torch_stream()))
            take = min(need, n.value)
            parts.append(desc[:take].clone())
            need -= take
            v += 1
            assert v < 4096 and n.value > 0, "variant images produce no keypoints"
        real_variants = v
    elif hi > P:
        r0 = max(lo, P)
        rnd = synth.make_descriptor_db(hi - r0, seed=synth.DB_SEED + r0)    # rows are a pure function of (seed, index block)
        pad = np.zeros((hi - r0, 64), np.uint8)
        pad[:, :61] = rnd
        parts.append(torch.from_numpy(pad).to(dev))
    db_local = torch.cat(parts).contiguous()
    db_xy = torch.zeros((NDB, 2), dtype=torch.float32, device=dev)
    db_xy[:P] = all_xy[:P]
    if args.db == "real":     # shuffle: the threshold pre-pass's first 16384 rows are then a random sample of the whole DB, not the planted rows
        gperm = torch.Generator(device=dev)
        gperm.manual_seed(0x5245414C)
        perm = torch.randperm(NDB, device=dev, generator=gperm)
        db_local, db_xy = db_local[perm].contiguous(), db_xy[perm].contiguous()
    torch.cuda.synchronize()
    # stand-alone extraction time (no other stage on the GPU) for the detect roofline: BEFORE the pipeline and its streams exist. (The
    # runtime maps streams onto a few hardware queues; once the pipeline's eight streams are around, the extraction's side stream shares
    # a queue with its main stream and the same call takes 2.2 ms instead of 1.9: measured both ways, profiles/r02.)
    solo = pl.FramePipeline(db_local[:4096].contiguous(), db_xy, index_base=0, group=None, device=str(dev))
    with torch.cuda.stream(solo.stream):
        nk = C.c_int(0)

        def extract_once(rep):
            f = frames[rep % len(frames)]
            check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), solo.cap, solo.kps.data_ptr(), solo.desc.data_ptr(), solo.cap,
                                           C.byref(nk), pl.torch_stream()))
        for rep in range(3):
            extract_once(rep)
        torch.cuda.synchronize()
        akaze_solo_n = 10

        def timed_solo():
            ts = time.perf_counter()
            for rep in range(akaze_solo_n):
                extract_once(rep)
            torch.cuda.synchronize()
            return (time.perf_counter() - ts) * 1e3
        akaze_solo_ms = timed_solo()
    del solo
    # The same stage as THROUGHPUT: two host threads, each with its own stream and output buffers, extract alternate frames, so one
    # frame's latency-bound phases (the small octaves' launch chain, the suppression rounds) run beside the other frame's bandwidth-bound
    # ones - what the streamed pipeline's two extraction workers do. Informational: `frac` stays the one-call-at-a-time figure.
    akaze_pair_ms = None
    if not args.serial:
        import threading

        def pair_worker(tid, reps, errs):
            try:
                torch.cuda.set_device(dev_index)
                check(L.apds_set_device(dev_index))
                st = torch.cuda.Stream(dev)
                k2 = torch.empty((cap, 7), dtype=torch.float32, device=dev)
                d2 = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
                n2 = C.c_int(0)
                with torch.cuda.stream(st):
                    for rep in range(reps):
                        f = frames[(rep + tid) % len(frames)]
                        check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), cap, k2.data_ptr(), d2.data_ptr(), cap, C.byref(n2),
                                                       pl.torch_stream()))
                    st.synchronize()
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
            finally:
                L.apds_thread_release()

        def run_pair(reps):
            errs = []
            ts = [threading.Thread(target=pair_worker, args=(t, reps, errs)) for t in range(2)]
            t0p = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if errs:
                raise errs[0]
            return (time.perf_counter() - t0p) * 1e3
        run_pair(3)
        akaze_pair_ms = run_pair(10) / 20.0
        torch.cuda.synchronize()
    # ... and as ONE batched call over four resident frames (apds_dev_akaze_extract_batch: every launch four images wide, so the dependent
    # launch chain that bounds a single frame is paid once per four). Informational as well; the batch's workspace is given back afterwards.
    akaze_batch4_ms = None
    if not args.serial and world == 1 and T * T * 4 * 24 * 4 < 40e9:
        with torch.cuda.stream(setup_stream):
            bimgs = torch.stack([frames[i % len(frames)] for i in range(4)]).contiguous()
            bcap = 1 << 16
            bk = torch.empty((4, bcap, 7), dtype=torch.float32, device=dev)
            bd = torch.empty((4, bcap, 64), dtype=torch.uint8, device=dev)
            bcounts = (C.c_int * 4)()
            try:
                for rep in range(4):
                    torch.cuda.synchronize()
                    tb = time.perf_counter()
                    check(L.apds_dev_akaze_extract_batch(bimgs.data_ptr(), 4, bimgs.stride(0), T, T, bimgs.shape[3], bimgs.stride(1), bcap, bk.data_ptr(), bd.data_ptr(),
                                                         bcap, bcounts, pl.torch_stream()))
                    torch.cuda.synchronize()
                    akaze_batch4_ms = (time.perf_counter() - tb) * 1e3 / 4
            except pkg._lib.ApdsError as e:   # (e.g. more keypoints per frame than the probe's capacity: the line is informational)
                print(f"[bench] batched extraction probe skipped: {e}", file=sys.stderr, flush=True)
                akaze_batch4_ms = None
            del bimgs, bk, bd
        check(L.apds_thread_release())
        check(L.apds_release_cached_memory())
        check(L.apds_set_device(dev_index))
    if world > 1 and os.environ.get("APDS_BENCH_FAIL_TRANSPORT") == args.transport:     # test hook: this transport "cannot be set up" (exercises the fallback)
        raise SystemExit(f"bench.py: APDS_BENCH_FAIL_TRANSPORT={args.transport}: simulated communicator failure")
    if args.serial:
        pipe = pl.FramePipeline(db_local, db_xy, index_base=lo, group=group, device=str(dev))
    else:
        pipe = pl.StreamedFramePipeline(db_local, db_xy, index_base=lo, group=group, device=str(dev), reserve_cus=args.reserve_cus, meta_group=meta_group,
                                        transport=args.transport)

    def run_step(i):
        return pipe.step(frames[i % len(frames)], filter_strength=args.filter_strength)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup, then EXACTLY K timed steps (K frames per rank, fully processed) between barriers
    if args.serial:
        for i in range(args.warmup):
            run_step(i)
        check(L.apds_dev_timing_enable(1))
        for name in ("hamming_topk", "hamming_topk_sample", "akaze_extract", "ransac_score"):
            pkg._lib.kernel_ms(name)     # drop warmup events
        fence()
        t0 = time.perf_counter()
        stats = []
        for i in range(args.steps):
            stats.append(run_step(args.warmup + i))
        fence()
        elapsed = time.perf_counter() - t0
        check(L.apds_dev_timing_enable(0))
        timers = {n: pkg._lib.kernel_ms(n) for n in ("hamming_topk", "hamming_topk_sample", "akaze_extract", "ransac_score")}
    else:
        # The library's own pipeline (apds_pipeline_*, csrc/pipeline.cpp): the timed loop below makes C calls only - submit K frames, poll K
        # results - no torch op, no python thread; the stage threads, streams, events and slots live inside libapds_hip.so.
        FR = pkg._lib.FrameResult

        def stream_frames(handle, fargs, count):
            res, out = FR(), []
            for i in range(count):
                ptr, stride, on_dev = fargs[i % len(fargs)]
                check(L.apds_pipeline_submit(handle, ptr, stride, on_dev, None))
            for i in range(count):
                check(L.apds_pipeline_poll(handle, C.byref(res), 1))
                if res.status != 0:
                    raise RuntimeError(f"frame {res.frame} failed in the pipeline: status {res.status}")
                out.append(dict(n_keypoints=res.n_keypoints, n_matches=res.n_matches, n_inliers=res.n_inliers, H=True if res.homography_found else None))
            return out

        def timed_stream(frame_list, warm):
            handle = pipe.prepare(frame_list[0].shape, filter_strength=args.filter_strength, timing=True)
            fargs = [pipe.frame_args(f) for f in frame_list]
            if warm:
                stream_frames(handle, fargs, warm)
            if os.environ.get("APDS_BENCH_FRESH_PIPE") == "1":      # A/B: fresh stage threads for the timed region (what the python pipeline of rounds 1-3 had)
                pipe._destroy_native()
                handle = pipe.prepare(frame_list[0].shape, filter_strength=args.filter_strength, timing=True)
            pipe.stats(reset=True)
            fence()
            t_start = time.perf_counter()
            got = stream_frames(handle, fargs, args.steps)
            fence()
            return time.perf_counter() - t_start, got, pipe.stats(reset=True)

        elapsed, stats, st = timed_stream(frames, args.warmup)
        timers = {"hamming_topk": (st.hamming_topk_ms, st.hamming_topk_launches), "hamming_topk_sample": (st.hamming_topk_sample_ms, st.hamming_topk_sample_launches),
                  "akaze_extract": (st.akaze_extract_ms, st.akaze_extract_calls), "ransac_score": (st.ransac_score_ms, st.ransac_score_launches)}
        split_scan = bool(st.split_scan)
        pipe.gap_log = [float(st.match_gaps_first_ms[i]) for i in range(min(16, st.match_gaps))]
        pipe.gap_mean = float(st.match_gap_mean_ms) if st.match_gaps else None
        if st.match_lds_cap_set_at_frame >= 0:
            pipe.cap_events = [dict(frame=int(st.match_lds_cap_set_at_frame), gaps_ms=[round(float(g), 2) for g in st.match_lds_cap_gaps_ms])]
    split_scan = locals().get("split_scan", False)
    elapsed_host = None
    if args.host_frames and not args.serial:
        # the same K steps with every frame coming from pinned host memory: an extraction worker uploads it (hipMemcpyAsync on its own
        # stream, into a per-slot device buffer) in front of the extraction, so frame i+1's PCIe copy travels under frame i's match
        host_frames = [torch.from_numpy(f).pin_memory() for f in frames_np]
        elapsed_host, stats_host, _ = timed_stream(host_frames, max(args.warmup, 2))
        assert [s["n_keypoints"] for s in stats_host] == [s["n_keypoints"] for s in stats], "host-frame run differs from the resident run"

    def max_over_ranks(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    elapsed = max_over_ranks(elapsed)
    if elapsed_host is not None:
        elapsed_host = max_over_ranks(elapsed_host)
    # The main scan alone on an idle GPU (three launches after the timed region): in the streamed run the neighbouring frames' pre-pass and
    # merge, and the extraction, share the GPU with it, so its live launch time reads longer than the kernel needs by itself.
    solo_launch_ms = None
    other_launch_ms, other_keys_equal = None, None
    if rank == 0 and not args.serial and stats and stats[0]["n_keypoints"] > 0:
        with torch.cuda.stream(setup_stream):
            n_solo = C.c_int(0)
            check(L.apds_dev_akaze_extract(frames[0].data_ptr(), T, T, frames[0].shape[2], frames[0].stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap,
                                           C.byref(n_solo), pl.torch_stream()))
            nq0 = n_solo.value
            keys_tmp = torch.empty((nq0, 2), dtype=torch.int64, device=dev)
            check(L.apds_dev_timing_enable(1))
            pkg._lib.kernel_ms("hamming_topk")
            for _ in range(3):
                check(L.apds_dev_hamming_topk(desc.data_ptr(), nq0, db_local.data_ptr(), db_local.shape[0], lo, 2, keys_tmp.data_ptr(), pl.torch_stream()))
            torch.cuda.synchronize()
            ms3, n3 = pkg._lib.kernel_ms("hamming_topk")
            solo_launch_ms = ms3 / max(n3, 1)
            # the OTHER backend alone on the same buffers (north_star names the integer formulation for the Hamming match: its figures stay
            # in every line): apds_dev_hamming_topk_backend, 1 = vector ALU, 2 = matrix cores
            backend_now = C.c_int(0)
            check(L.apds_dev_match_backend(C.byref(backend_now)))
            other = 1 if backend_now.value else 2
            keys_other = torch.empty((nq0, 2), dtype=torch.int64, device=dev)
            pkg._lib.kernel_ms("hamming_topk")
            for _ in range(2):
                check(L.apds_dev_hamming_topk_backend(desc.data_ptr(), nq0, db_local.data_ptr(), db_local.shape[0], lo, 2, keys_other.data_ptr(), other, pl.torch_stream()))
            torch.cuda.synchronize()
            ms_o, n_o = pkg._lib.kernel_ms("hamming_topk")
            other_launch_ms = ms_o / max(n_o, 1)
            other_keys_equal = bool(torch.equal(keys_tmp, keys_other))
            check(L.apds_dev_timing_enable(0))
    topk_ms, topk_n = timers.get("hamming_topk", (0.0, 0))
    sample_ms, sample_n = timers.get("hamming_topk_sample", (0.0, 0))
    akaze_ms, akaze_n = timers.get("akaze_extract", (0.0, 0))
    score_ms, score_n = timers.get("ransac_score", (0.0, 0))

    if rank == 0:
        K = float(np.mean([s["n_keypoints"] for s in stats]))
        Q_step = K * world                                   # queries matched per step by every rank (its shard)
        rows_local = hi - lo
        # The match of one step = a threshold pre-pass over the first `sample_rows` rows (kernel instance <1,2>, timed as
        # "hamming_topk_sample") + ONE main launch over the remaining rows (instance <4,2>, "hamming_topk"): the roofline
        # object is for the main launch. SURVEY §8d algorithmic work: 64 B per train row + 64 B per query + 8 B per key.
        sample_rows = min(16384, (rows_local // 16) & ~1023) if rows_local >= 32768 else 0      # match_hamming.hip: topk_device_k
        rows_main = rows_local - sample_rows
        match_bytes = 64.0 * rows_main + 64.0 * Q_step + 8.0 * Q_step * 2
        match_ops = 32.0 * Q_step * rows_main
        launches_per_step = topk_n / max(args.steps, 1)
        topk_ms_step = topk_ms / max(args.steps, 1)
        mfma_backend = C.c_int(0)
        check(L.apds_dev_match_backend(C.byref(mfma_backend)))
        peak = C.c_double(0)
        if not mfma_backend.value:   # the xor + popcount ceiling of the vector-ALU matcher (a register-only microbenchmark: 4 x 32 ms)
            check(L.apds_dev_valu_popcount_peak(C.byref(peak)))
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the PMC profile was taken on the default workload on one GPU: it says nothing about other DB / tile sizes
                backend_now = C.c_int(0)
                check(L.apds_dev_match_backend(C.byref(backend_now)))
                same_kernel = (tj.get("kernel", "hamming_topk_kernel<4, 2>") == "hamming_mfma_kernel") == bool(backend_now.value)
                if (tj.get("db_rows_per_gpu", 1_000_000), tj.get("tile", 4096)) == (rows_local, args.tile) and args.db == "mixed" and same_kernel:
                    traffic = tj.get("hamming_topk_hbm_bytes_per_launch")
                    # not measured in this run: rocprofv3 --pmc passes (tools/profile_bench.sh) wrote the file; say which commit's
                    traffic_source = f"profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh; collected at {tj.get('commit', 'an earlier commit')})"
            except Exception:
                traffic = None
        bytes_per_launch = match_bytes / max(launches_per_step, 1e-9)
        avg_launch_ms = topk_ms / max(topk_n, 1)
        achieved_gbps = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if topk_n else 0.0
        achieved_tops = match_ops / max(launches_per_step, 1e-9) / (avg_launch_ms * 1e-3) / 1e12 if topk_n else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        if mfma_backend.value:
            # The matrix-core matcher (csrc/hamming_mfma.hip): a threshold launch over the leading rows (a sixteenth, at most 16 384; timed as
            # "hamming_topk_sample" together with the expansion of the frame's queries into FP4 operands - the DB's expanded copy is made once at
            # apds_pipeline_create) and ONE main launch per step over the rest: the roofline object is for the main launch.
            # Algorithmic work per pair: 512 one-bit products + 512 adds on the padded 64-byte rows (SURVEY 8d counts the same 16 dwords) =
            # 1024 flop; peak = the guide's dense FP4 figure (MI355X_MICROARCH.md: ~10 PF, f8f6f4 with e2m1 operands = 4x the BF16 rate).
            sample_cap = int(os.environ.get("APDS_MATCH_MFMA_SAMPLE", "16384") or 0)
            mfma_sample = (min(sample_cap, (rows_local // 16) & ~127) & ~127) if (rows_local >= 65536 and sample_cap > 0) else 0   # hamming_mfma.hip: hm_sample_rows (the threshold launch,
            rows_main = rows_local - mfma_sample                                                  # timed as "hamming_topk_sample" with the query expansion)
            match_ops = 32.0 * Q_step * rows_main
            match_bytes = 64.0 * rows_main + 64.0 * Q_step + 8.0 * Q_step * 2
            bytes_per_launch = match_bytes / max(launches_per_step, 1e-9)
            achieved_gbps = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if topk_n else 0.0
            flop_per_launch = 1024.0 * Q_step * rows_main / max(launches_per_step, 1e-9)
            achieved_tflops = flop_per_launch / (avg_launch_ms * 1e-3) / 1e12 if topk_n else 0.0
            solo_tflops = (1024.0 * stats[0]["n_keypoints"] * rows_main / (solo_launch_ms * 1e-3) / 1e12) if solo_launch_ms else None
            lane_equiv = match_ops / max(launches_per_step, 1e-9) / (avg_launch_ms * 1e-3) / 1e12 if topk_n else 0.0
            roofline = {"kernel": "hamming_mfma_kernel (v_mfma_scale_f32_16x16x128_f8f6f4, e2m1 operands: train bit -> 1.0, query bit -> -2.0, accumulator preset to popcount(train))",
                        "bound": "mfma", "achieved": achieved_tflops, "peak": FP4_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tflops / FP4_DENSE_PEAK_TFLOPS,
                        "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": launches_per_step, "avg_launch_ms": avg_launch_ms,
                        "solo": ({"avg_launch_ms": solo_launch_ms, "queries": int(stats[0]["n_keypoints"]), "achieved": solo_tflops, "frac": solo_tflops / FP4_DENSE_PEAK_TFLOPS,
                                  "note": "the same kernel alone on the GPU after the timed region (frame 0's query count); `achieved` / `frac` above are the LIVE "
                                          "launches of the timed region, which share the GPU with the extraction of the following frames"} if solo_launch_ms else None),
                        "algorithmic_flop_per_launch": flop_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
                        "tpairs_per_s": Q_step * rows_main / (topk_ms_step * 1e-3) / 1e12 if topk_ms_step else 0.0,
                        "integer_path": ({"kernel": "hamming_topk_kernel<4,2> (xor + popcount on the vector ALU: APDS_MATCH_MFMA=0, and every k > 2)",
                                          "avg_launch_ms_solo": other_launch_ms, "bound": "int32-valu",
                                          "achieved": 32.0 * stats[0]["n_keypoints"] * (rows_local - (min(16384, (rows_local // 16) & ~1023) if rows_local >= 32768 else 0)) / (other_launch_ms * 1e-3) / 1e12,
                                          "peak": VALU_FP32_LANE_RATE_SPEC / 1e12, "unit": "T lane-op/s",
                                          "frac": 32.0 * stats[0]["n_keypoints"] * (rows_local - (min(16384, (rows_local // 16) & ~1023) if rows_local >= 32768 else 0)) / (other_launch_ms * 1e-3) / VALU_FP32_LANE_RATE_SPEC,
                                          "keys_equal_matrix_core_path": other_keys_equal,
                                          "note": "north_star's formulation of the Hamming match, alone on the GPU on frame 0's queries in this run (its main launch; "
                                                  "0.97 of the xor + half-rate-popcount ceiling); the matrix-core path returns the same keys"}
                                         if other_launch_ms else None),
                        "vector_alu_equivalent": {"achieved": lane_equiv, "unit": "T lane-op/s", "vs_valu_peak": lane_equiv / (VALU_FP32_LANE_RATE_SPEC / 1e12),
                                                  "note": "SURVEY 8d's figure for this path (32 xor + popcount lane-operations per pair) divided by the launch time: what the "
                                                          "vector-ALU formulation (APDS_MATCH_MFMA=0, hamming_topk_kernel: 0.63 of the 78.6 T peak, 0.97 of its xor + half-rate-bcnt ceiling) "
                                                          "would have to issue to keep up"},
                        "hbm": {"achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved_gbps / HBM_PEAK_GBPS},
                        "note": "1024 algorithmic flop per pair (512 bit products + 512 adds, exact in binary32: every partial sum is an integer below 2^11); the kernel "
                                "reads the train rows as 256-byte FP4 rows from the pipeline's expanded copy of the DB (4x the 64-byte rows `hbm` and `algorithmic_bytes_per_launch` count)"}
        else:
            # The binding bound of the dominant kernel is integer-VALU issue, so that is what `roofline` carries:
            # achieved = ALGORITHMIC lane-ops (32 per pair: 16 dword xor + 16 popcount-accumulate) / launch time;
            # peak = the guide's VALU rate (MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 op every 2 cycles = 128 lanes/clk/CU x 256 CU x
            # 2.4 GHz = 78.6 T lane-op/s) and frac = achieved / that. mix_ceiling = what THIS instruction mix can reach: the xor + bcnt pair
            # rate measured on this GPU by the register-only microbenchmark in its best issue order (apds_dev_valu_popcount_peak; v_xor_b32
            # issues at the full rate, v_bcnt_u32_b32 at half of it: 2 + 4 cycles per dword pair, 52.4 T in theory; calibration
            # profiles/r02/valu_calib_*.log). north_star's "% of HBM" is the sub-object `hbm`.
            roofline = {"kernel": f"hamming_topk_kernel<{4 if Q_step >= 16384 else (2 if Q_step >= 8192 else 1)},2>", "bound": "int32-valu",
                         "achieved": achieved_tops, "peak": VALU_FP32_LANE_RATE_SPEC / 1e12, "unit": "T lane-op/s", "frac": achieved_tops / (VALU_FP32_LANE_RATE_SPEC / 1e12),
                         "mix_ceiling": peak.value / 1e12, "frac_of_mix_ceiling": achieved_tops / (peak.value / 1e12) if peak.value else None,
                         "mix_ceiling_source": "apds_dev_valu_popcount_peak in this run (xor at the full VALU rate + half-rate bcnt = 6 issue cycles per dword pair); calibration of every instruction kind: profiles/r02/valu_calib_patterns.log, valu_calib_modes.log",
                         "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": launches_per_step, "avg_launch_ms": avg_launch_ms,
                         "solo": ({"avg_launch_ms": solo_launch_ms, "queries": int(stats[0]["n_keypoints"]),
                                   "frac": (32.0 * stats[0]["n_keypoints"] * rows_main / (solo_launch_ms * 1e-3) / VALU_FP32_LANE_RATE_SPEC) if solo_launch_ms else None,
                                   "frac_of_mix_ceiling": (32.0 * stats[0]["n_keypoints"] * rows_main / (solo_launch_ms * 1e-3) / peak.value) if peak.value and solo_launch_ms else None,
                                   "note": "the same kernel alone on the GPU after the timed region (frame 0's query count); `achieved` / `frac` above are the "
                                           "LIVE launches of the timed region, which share the GPU with the next frame's threshold pre-pass, the previous "
                                           "frame's merge and the extraction"} if solo_launch_ms else None),
                         "algorithmic_lane_ops_per_launch": match_ops / max(launches_per_step, 1e-9), "algorithmic_bytes_per_launch": bytes_per_launch,
                         "tpairs_per_s": Q_step * rows_main / (topk_ms_step * 1e-3) / 1e12 if topk_ms_step else 0.0,
                         "hbm": {"achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved_gbps / HBM_PEAK_GBPS},
                         "note": "32 algorithmic lane-ops per pair; peak = 128 lanes/clk/CU x 256 CU x 2.4 GHz (the guide's full-rate VALU figure, which no popcount loop reaches: "
                                 "v_bcnt_u32_b32 is a half-rate instruction); the kernel screens on 15 of the 16 dwords (30.6 issued ops per pair), so frac_of_mix_ceiling can pass 1.0 slightly"}
        out = {
            "metric": METRIC, "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp4 e2m1 operands / f32 accumulate = exact integer Hamming (match), f32 (AKAZE), f64 (homography solve)" if mfma_backend.value
                      else "u8/u32 popcount (match), f32 (AKAZE), f64 (homography solve)"), "data": "synthetic",
            "config": {"workload": f"frame{T}x{T}_bgra_detect+describe -> hamming_top2 vs db{NDB} (sharded/{world}) -> ratio{args.filter_strength} -> ransac_homography",
                       "tile": T, "db_rows": NDB, "db_rows_per_gpu": rows_local,
                       "db_composition": ("all rows are AKAZE descriptors of images (shifted frames + %d blended variants), shuffled" % real_variants) if args.db == "real"
                                         else f"{P} AKAZE descriptors of the frames' shifted copies + {NDB - P} i.i.d. random rows", "frames_per_step": world, "parallelism": f"frame-dp{world}+db-shard{world}",
                       "match_backend": "matrix cores (hamming_mfma_kernel)" if mfma_backend.value else "vector ALU (hamming_topk_kernel)",
                       "stage_overlap": "none (serial)" if args.serial else "extract (2 workers, alternate frames) | match (" + ("pre-pass, main scan and merge of consecutive frames on three streams" if split_scan else "one stream") + ") | homography, software-pipelined over frames by the library's own host threads (apds_pipeline_*: the timed loop is K submits + K polls)",
                       "match_occupancy_cap": ({"lds_bytes": 55000, "set_at": pipe.cap_events[0]} if getattr(pipe, "cap_events", None) else
                                               {"lds_bytes": int(os.environ.get("APDS_MATCH_LDS_CAP", "0") or 0)}),
                       "match_stream_gap_ms": (round(pipe.gap_mean, 3) if getattr(pipe, "gap_mean", None) is not None else None),
                       "match_stream_gaps_ms_first16": ([round(float(g), 2) for g in pipe.gap_log[:16]] if getattr(pipe, "gap_log", None) else None),
                       "keypoints_per_frame": K, "matches_per_frame": float(np.mean([s["n_matches"] for s in stats])),
                       "inliers_per_frame": float(np.mean([s["n_inliers"] for s in stats])), "homography_found": all(s["H"] is not None for s in stats)},
            "collectives": dict(collectives_info(torch, dist, world, backend, meta_group), transport=pipe.matcher.info()["transport"] if hasattr(pipe, "matcher") else None,
                                transport_requested=args.transport, fell_back_from=args.transport_fallback_from),
            "value_host_frames": (world * args.steps / elapsed_host) if elapsed_host else None,
            "ms_per_step_host_frames": (elapsed_host / args.steps * 1e3) if elapsed_host else None,
            "mmatches_per_s": world * K * args.steps / elapsed / 1e6,
            "gpairs_per_s": world * Q_step * rows_local * args.steps / elapsed / 1e9,
            "roofline": roofline,
            "stages_ms_per_step": {"akaze_extract": akaze_ms / max(args.steps, 1), "hamming_topk": topk_ms_step,
                                   "hamming_topk_sample": sample_ms / max(args.steps, 1), "ransac_score": score_ms / max(args.steps, 1)},
            "detect_roofline": {"bound": "hbm", "algorithmic_bytes_per_frame": detect_algorithmic_bytes(T, T),
                                "achieved": detect_algorithmic_bytes(T, T) / (akaze_solo_ms / max(akaze_solo_n, 1) * 1e-3) / 1e9 if akaze_solo_n else 0.0,
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": detect_algorithmic_bytes(T, T) / (akaze_solo_ms / max(akaze_solo_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS if akaze_solo_n else 0.0,
                                "ms_standalone": akaze_solo_ms / max(akaze_solo_n, 1),
                                "two_in_flight": ({"ms_per_frame": akaze_pair_ms, "frac": detect_algorithmic_bytes(T, T) / (akaze_pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                   "note": "throughput form: two host threads / streams extracting alternate frames (20 frames, wall clock); the "
                                                           "Hessian kernels are not forked to a side stream in this mode"} if akaze_pair_ms else None),
                                "batch_of_4": ({"ms_per_frame": akaze_batch4_ms, "frac": detect_algorithmic_bytes(T, T) / (akaze_batch4_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                "note": "throughput form: one apds_dev_akaze_extract_batch call over four resident frames (wall clock / 4)"}
                                               if akaze_batch4_ms else None),
                                "library_contexts_alive": int(L.apds_live_contexts()),
                                "note": "whole extraction (incl. orientation, descriptors, the count read-back), 10 back-to-back calls on resident frames timed by the wall clock with nothing else on the GPU, before the pipeline's streams are created, against the detect stages' algorithmic bytes; stages_ms_per_step.akaze_extract is its wall span while overlapped with the match (two frames are extracted concurrently, so the span may exceed the step time)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            # the GPU's keypoints and descriptors of frame 0, for the baseline's equality check against the oracle's
            n0 = C.c_int(0)
            with torch.cuda.stream(setup_stream):
                check(L.apds_dev_akaze_extract(frames[0].data_ptr(), T, T, frames[0].shape[2], frames[0].stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap,
                                               C.byref(n0), pl.torch_stream()))
                torch.cuda.synchronize()
            gpu_kps = kps[:n0.value].cpu().numpy()
            gpu_desc = desc[:n0.value, :61].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(pkg, frames_np[0], db_local, int(K), args.filter_strength, db_xy, gpu_kps, gpu_desc)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def collectives_info(torch, dist, world, backend, meta_group):
    """What carried the exchange step of this run: ranks, backend, and the RCCL build torch links (so a SCALE record shows that RCCL
    itself saw N ranks, not the gloo rehearsal)."""
    try:
        v = torch.cuda.nccl.version()
        rccl = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:   # noqa: BLE001
        rccl = None
    return {"world": world, "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else "none (one rank, no exchange step)",
            "rccl_version": rccl, "ranks_in_group": dist.get_world_size() if world > 1 else 1,
            "count_exchange": "apds_shard_counts: a 4-byte all-gather through the same transport, issued by the pipeline's match thread on its own stream" if world > 1 else None,
            "exchange": "all-gather of query rows + all-to-all of per-shard top-2 keys + u64-min merge" if world > 1 else None}


def bench_l2(args, pkg, pl, torch, dev, world):
    """BASELINE config 3: 256 x 1024^2 tiles' worth of float descriptors (Q = 1,048,576 x 128 f32, 30 % planted) vs a
    1M x 128 f32 DB, top-2 by L2 distance, one MI355X. The reference has no float/L2 matcher (lib.rs:101,121)."""
    assert world == 1, "the L2 workload is a single-GPU configuration"
    L, check = pkg.lib(), pkg._lib.check
    nq, nt, dim = args.l2_queries, args.db_rows, 128
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    g = torch.Generator(device=dev)
    g.manual_seed(0x4C32)
    db = torch.nn.functional.normalize(torch.randn((nt, dim), device=dev, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn((nq, dim), device=dev, generator=g), dim=1)
    npl = int(0.3 * nq)
    src = torch.randint(0, nt, (npl,), device=dev, generator=g)
    q[:npl] = torch.nn.functional.normalize(db[src] + 0.05 * torch.randn((npl, dim), device=dev, generator=g), dim=1)
    out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    mode = 1 if args.l2_mode == "screen" else 0
    used, cpq = C.c_int(-1), C.c_double(0)

    def step():
        check(L.apds_dev_l2_topk_ex(q.data_ptr(), nq, db.data_ptr(), nt, dim, 0, 2, mode, out.data_ptr(), pl.torch_stream(), C.byref(used), C.byref(cpq)))

    for _ in range(args.warmup):
        step()
    check(L.apds_dev_timing_enable(1))
    for name in ("l2_topk", "l2_screen", "l2_rerank"):
        pkg._lib.kernel_ms(name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms, n = pkg._lib.kernel_ms("l2_topk")
    ms_s, n_s = pkg._lib.kernel_ms("l2_screen")
    ms_r, n_r = pkg._lib.kernel_ms("l2_rerank")
    check(L.apds_dev_timing_enable(0))
    found = int(((out[:npl, 0] & 0xFFFFFFFF) == src).sum().item())
    flops = 2.0 * nq * nt * dim
    if used.value == 1:
        # per step: a top-2 pass over a 1/12 sample of the rows + the candidate pass over all rows. The roofline figure charges the
        # step's whole screen time (both launches) with ONE pass of algorithmic flops (2*Q*N*D): the sample pass is overhead.
        achieved = flops / (ms_s / max(args.steps, 1) * 1e-3) / 1e12
        roof = {"kernel": "l2_screen_kernel (v_mfma_f32_16x16x32_bf16)", "bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s",
                "frac": achieved / 2500.0, "traffic": None, "screen_ms_per_step": ms_s / max(args.steps, 1), "launches_per_step": n_s / max(args.steps, 1),
                "algorithmic_flops_per_launch": flops, "rerank_ms_per_step": ms_r / max(args.steps, 1), "candidates_per_query": cpq.value,
                "note": "screen = running top-2 over a 1/12 row sample (threshold), then candidate collection over all rows against the proved threshold; "
                        "the f32 re-rank of the candidates returns the exact mode's keys bit for bit"}
        dtype = "bf16 screen (v_mfma_f32_16x16x32_bf16, f32 accumulate) + f32 re-rank"
    else:
        achieved = flops / (ms / max(n, 1) * 1e-3) / 1e12
        roof = {"kernel": "l2_topk_kernel<2,true,128>", "bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3,
                "traffic": None, "avg_launch_ms": ms / max(n, 1), "algorithmic_flops_per_launch": flops}
        dtype = "f32 (v_mfma_f32_16x16x4_f32, f32 accumulate)"
    print(json.dumps({
        "metric": "Mmatches/sec (L2 brute-force top-2, float descriptors 128-d, vs 1M-row DB)", "value": nq * args.steps / elapsed / 1e6, "unit": "Mmatches/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"l2_top2 q{nq}x{dim} vs db{nt}x{dim} (BASELINE config 3)", "mode": "screen" if used.value == 1 else "exact",
                   "planted_recovered": found / max(npl, 1)},
        "roofline": roof}), flush=True)


def cpu_baseline(pkg, frame, db_local, K, fs, db_xy, gpu_kps, gpu_desc):
    """The oracle ("port": this repo's scalar/OpenMP C++ restatement) on the host cores, same frame and DB, built
    on this host with -O3 -march=native. Reported at ALL host cores (`value`, `cores`) and at ONE thread (`one_thread`). The match
    runs the full query set when a probe says it fits in 60 s, otherwise a stated sample (`sampled_queries`, `scale`)."""
    import oracle
    flags = oracle.use_native()
    threads, nproc = oracle.host_threads()
    db = db_local[:, :61].cpu().numpy()
    xy = db_xy.cpu().numpy()

    def leg(threads, budget_match_s, full_detect):
        oracle.set_threads(threads)
        if full_detect:
            t0 = time.perf_counter()
            ex = oracle.akaze(frame)
            t_detect, detect_scale = time.perf_counter() - t0, 1.0
        else:   # one thread: a quarter of the frame (its central half-size window), scaled by the pixel ratio
            h, w = frame.shape[0] // 2, frame.shape[1] // 2
            crop = np.ascontiguousarray(frame[h // 2:h // 2 + h, w // 2:w // 2 + w])
            t0 = time.perf_counter()
            oracle.akaze(crop)
            t_detect, detect_scale = (time.perf_counter() - t0) * 4.0, 4.0
            ex = None
        return t_detect, detect_scale, ex

    # ---- all cores
    t_detect, _, ex = leg(threads, 60.0, True)
    nq = len(ex.descriptors)
    probe = min(nq, max(64, int(2.0e9 / max(db.shape[0], 1))))
    t0 = time.perf_counter()
    oracle.get_knn_matches(ex.descriptors[:probe], db, 2, fs)
    t_probe = time.perf_counter() - t0
    est_full = t_probe * nq / max(probe, 1)
    nq_sample = nq if est_full <= 60.0 else max(probe, min(nq, int(nq * 20.0 / est_full)))
    t0 = time.perf_counter()
    m = oracle.get_knn_matches(ex.descriptors[:nq_sample], db, 2, fs)
    t_match_sample = time.perf_counter() - t0
    scale = nq / max(nq_sample, 1)
    t_match = t_match_sample * scale
    t_h = 0.0
    if len(m) >= 4:
        p1 = np.stack([ex.keypoints["x"][m["query_idx"]], ex.keypoints["y"][m["query_idx"]]], 1)
        p2 = xy[m["train_idx"]]
        t0 = time.perf_counter()
        oracle.find_homography(p1, p2, 8, 3.0)
        t_h = time.perf_counter() - t0
    total = t_detect + t_match + t_h
    # ---- one thread, bounded: quarter-frame detect (x4), ~1e9 descriptor pairs of the match (scaled), the same RANSAC time
    t1_detect, d_scale, _ = leg(1, 10.0, False)
    nq1 = max(16, min(nq, int(1.0e9 / max(db.shape[0], 1))))
    oracle.set_threads(1)
    t0 = time.perf_counter()
    oracle.get_knn_matches(ex.descriptors[:nq1], db, 2, fs)
    t1_match_sample = time.perf_counter() - t0
    t1_match = t1_match_sample * nq / nq1
    total1 = t1_detect + t1_match + t_h
    oracle.set_threads(threads)
    return {"value": 1.0 / total, "unit": "frames/s", "cores": threads, "kind": "port", "nproc": nproc, "build_flags": flags,
            "threads_note": f"nproc = {nproc}; {threads} OpenMP threads used (the fastest of 8/16/32/64/.../nproc on a small detect: the box's CPU share is below its affinity mask)",
            "sample": f"oracle ({flags}) on {threads} OpenMP threads: full {frame.shape[1]}x{frame.shape[0]} detect+describe ({t_detect:.2f} s, K={nq}), "
                      f"match of {nq_sample} of the {nq} queries vs all {db.shape[0]} rows ({t_match_sample:.2f} s, x{scale:.2f}), RANSAC on those matches ({t_h * 1e3:.1f} ms)",
            "sampled_queries": nq_sample, "total_queries": nq, "scale": scale,
            "seconds_per_frame": total, "seconds": {"detect": t_detect, "match": t_match, "homography": t_h},
            "one_thread": {"value": 1.0 / total1, "unit": "frames/s", "cores": 1, "seconds_per_frame": total1,
                           "seconds": {"detect": t1_detect, "match": t1_match, "homography": t_h},
                           "sample": f"1 thread: detect+describe of the central {frame.shape[1] // 2}x{frame.shape[0] // 2} window x{d_scale:.0f}, "
                                     f"match of {nq1} of the {nq} queries ({t1_match_sample:.2f} s, x{nq / nq1:.1f})",
                           "sampled_queries": nq1, "scale": nq / nq1},
            # every keypoint field (x, y, size, angle, response as f32 bits; octave, class_id) and every descriptor byte of frame 0
            "keypoints_equal_gpu": bool(len(gpu_kps) == nq and np.array_equal(gpu_kps.view(np.uint32).reshape(-1, 7), ex.keypoints.view(np.uint32).reshape(-1, 7))
                                        and np.array_equal(gpu_desc, ex.descriptors))}


if __name__ == "__main__":
    sys.exit(main() or 0)
