"""GPU: the register-strip kernels (FED steps, smoothing + conductivity, the fused base stage) serve the large levels only; the LDS
tile kernels serve the rest. This runs the whole AKAZE parity file and the random-shape AKAZE fuzz again in a child process with the strips forced on for EVERY
level size (APDS_*_STRIP=2; the switches are read once per process), so their border variants meet the small, odd-sized, strided
and 1/3/4-channel images of those tests and must reproduce the oracle bit for bit there too."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_akaze_parity_with_strips_forced_on_every_level(gpu_pkg):
    env = dict(os.environ, APDS_NLD_STRIP="2", APDS_SF_STRIP="2", APDS_BASE_STRIP="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_akaze_gpu.py"),
                        os.path.join("tests", "test_fuzz_gpu.py") + "::test_akaze_random_shapes", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout


def _rerun(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_akaze_gpu.py"),
                        os.path.join("tests", "test_fuzz_gpu.py") + "::test_akaze_random_shapes", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_akaze_parity_with_level_strips_on_every_level(gpu_pkg):
    """level_strip_kernel (smoothing + conductivity + FED steps on register strips, borders included) normally serves levels of
    1 .. 8 Mpx: here it serves every level of every test image, so its border waves meet the small and odd-sized ones."""
    _rerun({"APDS_LEVEL_STRIP": "2", "APDS_LEVEL_FUSE": "0", "APDS_LEVEL_STREAM": "0"})


def test_akaze_parity_with_lds_fused_levels_on_every_level(gpu_pkg):
    """level_fused_kernel (one launch per level, register patches exchanging through LDS) normally serves levels up to 1 Mpx."""
    _rerun({"APDS_LEVEL_FUSE": "2", "APDS_LEVEL_STRIP": "0"})


def test_akaze_parity_on_the_round_2_path(gpu_pkg):
    """separate smoothing / FED launches per level, keypoints placed by two passes over the masks, the LDS-tile Hessian kernel on every
    level with the masks cleared by the zeroing kernel, default events: none of round 3's streaming kernels"""
    _rerun({"APDS_LEVEL_FUSE": "0", "APDS_LEVEL_STRIP": "0", "APDS_KP_RANKED": "0", "APDS_EVENT_SCOPE": "1", "APDS_DOH_STRIP": "0", "APDS_LEVEL_STREAM": "0",
            "APDS_KP_XCD": "0", "APDS_HALF_FUSE": "0"})


def test_akaze_parity_with_full_fed_sweeps_in_the_fused_levels(gpu_pkg):
    """level_fused_kernel on every level without round 4's shrinking zone (every FED step sweeps the whole halo'd region, as in round 3): the
    default (shrinking) runs in the test above and in test_akaze_gpu.py itself."""
    _rerun({"APDS_LEVEL_FUSE": "2", "APDS_LEVEL_STRIP": "0", "APDS_FED_SHRINK": "0"})


def test_akaze_parity_with_the_streaming_kernels_on_every_level(gpu_pkg):
    """Round 3's streaming kernels - doh_strip_kernel (akaze_doh_strips.hip: determinant of the Hessian + extrema walking down 64-column
    strips, the masks written for every pixel instead of cleared) and level_stream_kernel (akaze_level_stream.hip: Gaussian, Scharr,
    conductivity and the first FED steps of a level, the same way) - normally serve levels of 8 Mpx and more. Here they serve every level
    of at least 64 pixels of every test image: top / bottom bands with reflected or clamped rows, first / last strips with reflected or
    clamped columns, partial last bands; once with 16-row bands and once with tall ones."""
    _rerun({"APDS_DOH_STRIP": "2", "APDS_DOH_STRIP_ROWS": "16", "APDS_LEVEL_STREAM": "2", "APDS_LEVEL_STREAM_ROWS": "16", "APDS_LEVEL_STRIP": "2", "APDS_LEVEL_FUSE": "0"})
    _rerun({"APDS_DOH_STRIP": "2", "APDS_DOH_STRIP_ROWS": "112", "APDS_LEVEL_STREAM": "2", "APDS_LEVEL_STREAM_ROWS": "100", "APDS_LEVEL_STRIP": "2", "APDS_LEVEL_FUSE": "0"})


def test_akaze_parity_with_the_value_fork(gpu_pkg):
    """APDS_FLAG_FORK=1: the Hessian stream waits for a sequence number that the next kernel of the level chain stores as its first act
    (hipStreamWaitValue32) instead of for an event recorded between the chain's kernels; levels whose fork follows their last kernel defer
    their Hessian launch to the next level. Every kernel family's launcher carries the signal here (streaming, strips, LDS-fused; the small
    separate kernels take the one-thread fallback)."""
    _rerun({"APDS_FLAG_FORK": "1"})
    _rerun({"APDS_FLAG_FORK": "1", "APDS_LEVEL_FUSE": "0", "APDS_LEVEL_STRIP": "0", "APDS_LEVEL_STREAM": "0"})


def test_match_parity_on_the_vector_alu_matcher(gpu_pkg):
    """APDS_MATCH_MFMA=0: hamming_topk_kernel (xor + popcount on the vector ALU) serves k <= 2 as well - the default sends those to the FP4
    matrix pipe (hamming_mfma.hip). Both must reproduce the oracle's keys: the match tests, the matcher fuzz, the sharded matcher and the
    keypoint table's match run again in a child process on the vector kernel."""
    env = dict(os.environ, APDS_MATCH_MFMA="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_match_gpu.py"), os.path.join("tests", "test_fuzz_gpu.py") + "::test_match_random_shapes",
                        os.path.join("tests", "test_shard_native.py"), os.path.join("tests", "test_keypoint_table_gpu.py"), "-q", "-m", "gpu", "-x", "-p",
                        "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
