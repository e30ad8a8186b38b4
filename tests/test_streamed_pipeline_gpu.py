"""GPU: the library's streamed pipeline (apds_pipeline_*, csrc/pipeline.cpp; bench.py's default path: extract | match | homography on
their own host threads and streams inside libapds_hip.so, two extraction workers) through its python front must give, frame by frame,
what the one-frame-at-a-time FramePipeline gives; and its starvation watch must cap the occupancy of ITS scans when extraction is made
artificially late. (tests/cpp/pipeline_test.cpp drives the same four calls from a g++-built host.)"""
import ctypes as C

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(pkg, T=768, ndb=60000, nframes=3):
    import torch
    from importlib import import_module
    pl = import_module(pkg.__name__ + ".pipeline")
    L, check, synth = pkg.lib(), pkg._lib.check, pkg.synth
    dev = torch.device("cuda:0")
    frames_np = [synth.make_tile(T, T, frame_index=40 + i) for i in range(nframes)]
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]
    cap = pkg.feature_extraction.MAX_POINTS
    kps = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
    rows, xy = [], []
    with torch.cuda.stream(torch.cuda.Stream(dev)):       # extraction and the torch copies of its output in one stream
        for f in frames_np:                               # DB: descriptors of a shifted copy of every frame + random rows
            rolled = torch.from_numpy(np.roll(f, (19, 23), axis=(0, 1)).copy()).to(dev)
            n = C.c_int(0)
            check(L.apds_dev_akaze_extract(rolled.data_ptr(), T, T, 4, rolled.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n),
                                           pl.torch_stream()))
            rows.append(desc[:n.value].clone())
            xy.append(kps[:n.value, 0:2].clone())
            torch.cuda.synchronize()
    rows, xy = torch.cat(rows), torch.cat(xy)
    P = rows.shape[0]
    pad = np.zeros((ndb - P, 64), np.uint8)
    pad[:, :61] = synth.make_descriptor_db(ndb - P)
    db = torch.cat([rows, torch.from_numpy(pad).to(dev)]).contiguous()
    db_xy = torch.zeros((ndb, 2), dtype=torch.float32, device=dev)
    db_xy[:P] = xy
    torch.cuda.synchronize()
    return pl, frames, db, db_xy


def test_streamed_results_equal_the_serial_pipeline(gpu_pkg):
    pl, frames, db, db_xy = _setup(gpu_pkg)
    serial = pl.FramePipeline(db, db_xy)
    want = [serial.step(frames[i % len(frames)], filter_strength=0.3) for i in range(7)]
    streamed = pl.StreamedFramePipeline(db, db_xy)
    got, _ = streamed.run(frames, 7, filter_strength=0.3)
    assert len(got) == 7
    for a, b in zip(want, got):
        assert a["n_keypoints"] == b["n_keypoints"] > 500 and a["n_matches"] == b["n_matches"] > 100 and a["n_inliers"] == b["n_inliers"] > 50
        assert a["H"] is not None and np.array_equal(a["H"], b["H"])
        assert abs(a["H"][0, 2] - 23) < 0.5 and abs(a["H"][1, 2] - 19) < 0.5      # the planted translation


def test_streamed_pipeline_with_frames_that_find_nothing(gpu_pkg):
    """A stream is not all good frames: a blank frame (no keypoint, hence no query, no match, no homography), a frame of noise (keypoints
    but nothing in the DB resembles them: too few matches survive the ratio test) between good ones - the pipeline must report each as
    the serial pipeline does and carry on with the next frame (split scan, slots and events are per frame: none may be left half-used)."""
    import torch
    pl, frames, db, db_xy = _setup(gpu_pkg, T=512, ndb=40000, nframes=2)
    dev = frames[0].device
    rng = np.random.default_rng(3)
    blank = torch.zeros_like(frames[0])
    blank[..., 3] = 255
    noise = torch.from_numpy(rng.integers(0, 256, tuple(frames[0].shape), dtype=np.uint8)).to(dev)
    seq = [frames[0], blank, frames[1], noise, blank, blank, frames[0], noise, frames[1]]
    serial = pl.FramePipeline(db, db_xy)
    want = [serial.step(f, filter_strength=0.3) for f in seq]
    streamed = pl.StreamedFramePipeline(db, db_xy)
    got, _ = streamed.run(seq, len(seq), filter_strength=0.3)
    assert len(got) == len(seq)
    for i, (a, b) in enumerate(zip(want, got)):
        assert a["n_keypoints"] == b["n_keypoints"] and a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"], (i, a, b)
        assert (a["H"] is None) == (b["H"] is None) and (a["H"] is None or np.array_equal(a["H"], b["H"])), i
    assert want[1]["n_keypoints"] == 0 and want[1]["H"] is None
    assert want[3]["n_keypoints"] > 100 and want[3]["H"] is None
    assert want[0]["H"] is not None and want[6]["H"] is not None and want[8]["H"] is not None


def test_starvation_watch_caps_the_match_kernel(gpu_pkg):
    """The watch belongs to the vector-ALU matcher (its occupancy cap is an LDS request of hamming_topk_kernel); the matrix-core matcher
    (the default, APDS_MATCH_MFMA=1) has no such knob and the pipeline does not watch it. The switch is read once per process, so under the
    default this test runs itself again in a child process with APDS_MATCH_MFMA=0."""
    if os.environ.get("APDS_MATCH_MFMA", "1") != "0":
        env = dict(os.environ, APDS_MATCH_MFMA="0")
        r = subprocess.run([sys.executable, "-m", "pytest", __file__ + "::test_starvation_watch_caps_the_match_kernel", "-q", "-m", "gpu", "-x", "-p",
                            "no:cacheprovider"], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])
        return
    pl, frames, db, db_xy = _setup(gpu_pkg)
    L, check = gpu_pkg.lib(), gpu_pkg._lib.check
    streamed = pl.StreamedFramePipeline(db, db_xy)
    streamed.debug_extract_delay = 0.012                  # every frame reaches the match 12 ms late: its stream sits idle
    old, lds = C.c_int(-1), C.c_int(-1)
    try:
        got, _ = streamed.run(frames, 16, filter_strength=0.3)
        assert all(r is not None and r["H"] is not None for r in got)
        assert streamed.cap_events and streamed.cap_events[0]["previous"] == 0 and sum(1 for g in streamed.cap_events[0]["gaps_ms"] if g > 4.0) >= 3
        # observable effect: the scans that followed were launched with the cap as their dynamic-LDS request (these 768^2 tiles
        # take the one-query-per-lane kernel variants, not only the T = 4 one)
        check(L.apds_dev_match_last_launch_lds(C.byref(lds)))
        assert lds.value == 55000 and streamed.cap_bytes == 55000
        # the cap is scoped to the pipeline's own match thread: the process-wide value was never touched
        check(L.apds_dev_match_lds_cap(0, C.byref(old)))
        assert old.value == 0
        # a second run of the same pipeline - re-created, because the delay parameter changed - starts capped
        streamed.debug_extract_delay = 0.0
        streamed.run(frames, 3, filter_strength=0.3)
        check(L.apds_dev_match_last_launch_lds(C.byref(lds)))
        assert lds.value == 55000 and streamed.stats().match_lds_cap_bytes == 55000
        # and another matcher in the same process is not capped
        pl.FramePipeline(db, db_xy).step(frames[0], filter_strength=0.3)
        check(L.apds_dev_match_last_launch_lds(C.byref(lds)))
        assert lds.value == 0
    finally:
        streamed.close()
        check(L.apds_dev_match_lds_cap(0, None))


def test_host_frames_and_resident_frames_give_the_same_results(gpu_pkg):
    """apds_pipeline_submit(on_device = 0): the extraction worker uploads the frame on its own stream in front of the extraction."""
    pl, frames, db, db_xy = _setup(gpu_pkg, T=512, ndb=40000, nframes=2)
    streamed = pl.StreamedFramePipeline(db, db_xy)
    want, _ = streamed.run(frames, 5, filter_strength=0.3)
    host = [f.cpu().pin_memory() for f in frames]
    got, timers = streamed.run(host, 5, filter_strength=0.3, timing=True)
    for a, b in zip(want, got):
        assert a["n_keypoints"] == b["n_keypoints"] > 100 and a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"]
        assert a["H"] is not None and np.array_equal(a["H"], b["H"])
    # timing on: every frame's main scan and extraction were timed with HIP events on their launch streams
    assert timers["hamming_topk"][1] == 5 and timers["hamming_topk"][0] > 0 and timers["akaze_extract"][1] == 5
    streamed.close()


def test_reserved_cus_reach_the_match_stream(gpu_pkg):
    """the CU-masked stream must be the one the match workers launch on."""
    pl, frames, db, db_xy = _setup(gpu_pkg)
    streamed = pl.StreamedFramePipeline(db, db_xy, reserve_cus=16)
    try:
        assert streamed._masked_stream_handle is not None       # (handed to the native pipeline as apds_pipeline_params.match_stream)
        got, _ = streamed.run(frames, 4, filter_strength=0.3)
        assert all(r is not None and r["H"] is not None for r in got)
    finally:
        streamed.close()
    assert streamed._masked_stream_handle is None


def test_the_stage_threads_workspaces_do_not_grow_with_the_frame_count(gpu_pkg):
    """Every stage thread's scratch is its thread workspace, reset at the start of every scan / solve: device memory in use after 60 frames
    and after 360 more must be the same (round 4: the matrix-core scan of the one-GPU pipeline once skipped that reset and grew by a
    frame's scratch per frame)."""
    import torch
    pl, frames, db, db_xy = _setup(gpu_pkg, T=512, ndb=70000, nframes=2)
    streamed = pl.StreamedFramePipeline(db, db_xy)
    try:
        streamed.run(frames, 60, filter_strength=0.3)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            got, _ = streamed.run(frames, 120, filter_strength=0.3)
            assert all(r is not None for r in got)
        torch.cuda.synchronize()
        assert abs(torch.cuda.mem_get_info()[0] - free0) < 16 << 20
    finally:
        streamed.close()
