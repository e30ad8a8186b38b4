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
VALU_INT_PEAK_SPEC = 256 * 64 * 2.4e9   # 32-bit integer VALU lane-ops/s (64 lanes/clk/CU)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--db-rows", type=int, default=1_000_000, help="total descriptor DB rows (sharded over the ranks)")
    ap.add_argument("--frames", type=int, default=2, help="distinct frames per rank, cycled")
    ap.add_argument("--filter-strength", type=float, default=0.3, help="Lowe ratio (reference test: 0.3, lib.rs:222)")
    ap.add_argument("--workload", choices=["frame", "l2"], default="frame",
                    help="frame: the north-star pipeline (default). l2: BASELINE config 3, float-descriptor L2 match as an MFMA GEMM (1 GPU)")
    ap.add_argument("--l2-queries", type=int, default=1_048_576)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reserve-cus", type=int, default=0, help="CUs masked out of the match stream so the other stages overlap it")
    ap.add_argument("--serial", action="store_true", help="one frame at a time (no cross-frame overlap of the three stages)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("APDS_BENCH_BACKEND", "nccl")      # "gloo" = single-GPU rehearsal of the multi-rank path
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    group = meta_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD
        # host-side exchange of the per-frame query counts (no device sync). Single node: loopback is always usable, the
        # container hostname may not resolve. If gloo cannot be set up the pipeline falls back to a device-side count gather.
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        try:
            meta_group = dist.new_group(backend="gloo")
        except Exception as e:   # noqa: BLE001
            print(f"[bench] gloo meta group unavailable ({e}); using device-side counts", file=sys.stderr, flush=True)
            meta_group = None

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
    # global DB = [P planted rows | NDB - P random rows]; this rank keeps rows [lo, hi)
    lo, hi = rank * NDB // world, (rank + 1) * NDB // world
    parts = []
    if lo < P:
        parts.append(all_rows[lo:min(hi, P)])
    if hi > P:
        r0 = max(lo, P)
        rnd = synth.make_descriptor_db(hi - r0, seed=synth.DB_SEED + r0)    # rows are a pure function of (seed, index block)
        pad = np.zeros((hi - r0, 64), np.uint8)
        pad[:, :61] = rnd
        parts.append(torch.from_numpy(pad).to(dev))
    db_local = torch.cat(parts).contiguous()
    db_xy = torch.zeros((NDB, 2), dtype=torch.float32, device=dev)
    db_xy[:P] = all_xy[:P]
    torch.cuda.synchronize()
    if args.serial:
        pipe = pl.FramePipeline(db_local, db_xy, index_base=lo, group=group, device=str(dev))
    else:
        pipe = pl.StreamedFramePipeline(db_local, db_xy, index_base=lo, group=group, device=str(dev), reserve_cus=args.reserve_cus, meta_group=meta_group)

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
        if args.warmup:
            pipe.run(frames, args.warmup, filter_strength=args.filter_strength)
        fence()
        t0 = time.perf_counter()
        stats, timers = pipe.run(frames, args.steps, filter_strength=args.filter_strength, timing=True)
        fence()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # stand-alone extraction time (no other stage on the GPU) for the detect roofline; outside the timed region
    solo = pl.FramePipeline(db_local[:4096].contiguous(), db_xy, index_base=0, group=None, device=str(dev))
    with torch.cuda.stream(solo.stream):
        nk = C.c_int(0)
        for rep in range(4):
            if rep == 1:
                check(L.apds_dev_timing_enable(1))
                pkg._lib.kernel_ms("akaze_extract")
            f = frames[rep % len(frames)]
            check(L.apds_dev_akaze_extract(f.data_ptr(), T, T, f.shape[2], f.stride(0), solo.cap, solo.kps.data_ptr(), solo.desc.data_ptr(), solo.cap,
                                           C.byref(nk), pl.torch_stream()))
        torch.cuda.synchronize()
        akaze_solo_ms, akaze_solo_n = pkg._lib.kernel_ms("akaze_extract")
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
        peak = C.c_double(0)
        check(L.apds_dev_valu_popcount_peak(C.byref(peak)))
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the PMC profile was taken on the default workload on one GPU: it says nothing about other DB / tile sizes
                if (tj.get("db_rows_per_gpu", 1_000_000), tj.get("tile", 4096)) == (rows_local, args.tile):
                    traffic = tj.get("hamming_topk_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        bytes_per_launch = match_bytes / max(launches_per_step, 1e-9)
        avg_launch_ms = topk_ms / max(topk_n, 1)
        achieved_gbps = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if topk_n else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": METRIC, "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 popcount (match), f32 (AKAZE), f64 (homography solve)", "data": "synthetic",
            "config": {"workload": f"frame{T}x{T}_bgra_detect+describe -> hamming_top2 vs db{NDB} (sharded/{world}) -> ratio{args.filter_strength} -> ransac_homography",
                       "tile": T, "db_rows": NDB, "db_rows_per_gpu": rows_local, "frames_per_step": world, "parallelism": f"frame-dp{world}+db-shard{world}",
                       "stage_overlap": "none (serial)" if args.serial else "extract (2 workers, alternate frames) | match | homography on their own streams, software-pipelined over frames",
                       "match_occupancy_cap": ({"lds_bytes": 55000, "set_at": pipe.cap_events[0]} if getattr(pipe, "cap_events", None) else
                                               {"lds_bytes": int(os.environ.get("APDS_MATCH_LDS_CAP", "0") or 0)}),
                       "match_stream_gap_ms": (round(float(np.mean(pipe.gap_log)), 3) if getattr(pipe, "gap_log", None) else None),
                       "match_stream_gaps_ms_first16": ([round(float(g), 2) for g in pipe.gap_log[:16]] if getattr(pipe, "gap_log", None) else None),
                       "keypoints_per_frame": K, "matches_per_frame": float(np.mean([s["n_matches"] for s in stats])),
                       "inliers_per_frame": float(np.mean([s["n_inliers"] for s in stats])), "homography_found": all(s["H"] is not None for s in stats)},
            "mmatches_per_s": world * K * args.steps / elapsed / 1e6,
            "gpairs_per_s": world * Q_step * rows_local * args.steps / elapsed / 1e9,
            "roofline": {"kernel": "hamming_topk_kernel<4,2>", "bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": traffic, "launches_per_step": launches_per_step,
                         "avg_launch_ms": avg_launch_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "north_star asks for the Hamming match as a fraction of HBM; the kernel is integer-VALU bound (32 lane-ops per pair), see valu"},
            "valu": {"bound": "int32 VALU xor+popcount", "achieved_lane_ops_per_s": match_ops / (topk_ms_step * 1e-3) if topk_ms_step else 0.0,
                     "peak_measured_lane_ops_per_s": peak.value, "peak_spec_lane_ops_per_s": VALU_INT_PEAK_SPEC,
                     "frac_of_measured": match_ops / (topk_ms_step * 1e-3) / peak.value if topk_ms_step else 0.0,
                     "frac_of_spec": match_ops / (topk_ms_step * 1e-3) / VALU_INT_PEAK_SPEC if topk_ms_step else 0.0,
                     "tpairs_per_s": Q_step * rows_main / (topk_ms_step * 1e-3) / 1e12 if topk_ms_step else 0.0,
                     "note": "algorithmic 32 lane-ops per pair (16 dword xor + 16 popcount); the fast path screens on 15 dwords = 30 ops, which is why the fraction can exceed the measured pair rate x 32"},
            "stages_ms_per_step": {"akaze_extract": akaze_ms / max(args.steps, 1), "hamming_topk": topk_ms_step,
                                   "hamming_topk_sample": sample_ms / max(args.steps, 1), "ransac_score": score_ms / max(args.steps, 1)},
            "detect_roofline": {"bound": "hbm", "algorithmic_bytes_per_frame": detect_algorithmic_bytes(T, T),
                                "achieved": detect_algorithmic_bytes(T, T) / (akaze_solo_ms / max(akaze_solo_n, 1) * 1e-3) / 1e9 if akaze_solo_n else 0.0,
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": detect_algorithmic_bytes(T, T) / (akaze_solo_ms / max(akaze_solo_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS if akaze_solo_n else 0.0,
                                "ms_standalone": akaze_solo_ms / max(akaze_solo_n, 1),
                                "note": "whole extraction (incl. orientation, descriptors, count read-backs) run alone after the timed region, against the detect stages' algorithmic bytes; stages_ms_per_step.akaze_extract is its wall span while overlapped with the match (two frames are extracted concurrently, so the span may exceed the step time)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, frames_np[0], db_local, int(K), args.filter_strength, db_xy, stats[0])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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

    def step():
        check(L.apds_dev_l2_topk(q.data_ptr(), nq, db.data_ptr(), nt, dim, 0, 2, out.data_ptr(), pl.torch_stream()))

    for _ in range(args.warmup):
        step()
    check(L.apds_dev_timing_enable(1))
    pkg._lib.kernel_ms("l2_topk")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms, n = pkg._lib.kernel_ms("l2_topk")
    check(L.apds_dev_timing_enable(0))
    found = int(((out[:npl, 0] & 0xFFFFFFFF) == src).sum().item())
    flops = 2.0 * nq * nt * dim
    achieved = flops / (ms / max(n, 1) * 1e-3) / 1e12
    print(json.dumps({
        "metric": "Mmatches/sec (L2 brute-force top-2, float descriptors 128-d, vs 1M-row DB)", "value": nq * args.steps / elapsed / 1e6, "unit": "Mmatches/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (v_mfma_f32_32x32x2_f32, f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"l2_top2 q{nq}x{dim} vs db{nt}x{dim} (BASELINE config 3)", "planted_recovered": found / max(npl, 1)},
        "roofline": {"kernel": "l2_topk_kernel<2,true>", "bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3,
                     "traffic": None, "avg_launch_ms": ms / max(n, 1), "algorithmic_flops_per_launch": flops}}), flush=True)


def cpu_baseline(pkg, frame, db_local, K, fs, db_xy, ref_stats):
    """The oracle ("port") on the host cores, same frame and DB: full detect+describe, a bounded sample of the
    query rows for the match (scaled to K), ratio test, RANSAC on that sample's matches."""
    import oracle
    threads = max(1, min(os.cpu_count() or 1, 32))
    oracle.set_threads(threads)
    t0 = time.perf_counter()
    ex = oracle.akaze(frame)
    t_detect = time.perf_counter() - t0
    db = db_local[:, :61].cpu().numpy()
    # ~1e10 descriptor pairs: about 9 s on 32 host threads, so that the whole CPU sample is 10 - 30 s of work on the box's cores
    nq_sample = max(1, min(len(ex.descriptors), max(256, int(1.0e10 / max(db.shape[0], 1)))))
    t0 = time.perf_counter()
    m = oracle.get_knn_matches(ex.descriptors[:nq_sample], db, 2, fs)
    t_match_sample = time.perf_counter() - t0
    t_match = t_match_sample * len(ex.descriptors) / nq_sample
    t_h = 0.0
    if len(m) >= 4:
        xy = db_xy.cpu().numpy()
        p1 = np.stack([ex.keypoints["x"][m["query_idx"]], ex.keypoints["y"][m["query_idx"]]], 1)
        p2 = xy[m["train_idx"]]
        t0 = time.perf_counter()
        oracle.find_homography(p1, p2, 8, 3.0)
        t_h = (time.perf_counter() - t0)
    total = t_detect + t_match + t_h
    return {"value": 1.0 / total, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle on {threads} OpenMP threads: full {frame.shape[1]}x{frame.shape[0]} detect+describe ({t_detect:.2f}s, K={len(ex.keypoints)}), "
                      f"match of {nq_sample} of the {len(ex.descriptors)} queries vs all {db.shape[0]} rows ({t_match_sample:.2f}s, scaled x{len(ex.descriptors) / nq_sample:.1f}), "
                      f"RANSAC on the sample's matches ({t_h * 1e3:.1f} ms)",
            "seconds_per_frame_estimated": total, "keypoints_equal_gpu": int(len(ex.keypoints)) == int(ref_stats["n_keypoints"])}


if __name__ == "__main__":
    main()
