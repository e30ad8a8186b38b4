"""The row-sharded matcher behind the C ABI (apds_shard_*, csrc/shard.cpp) driven by a host that is not Python:
tests/cpp/shard_loopback_test.cpp, built by g++ against libapds_hip.so. CPU: it compiles and links (every apds_shard_* /
apds_dev_alloc / ... symbol resolves). GPU: it runs - threads as ranks on the one GPU over the loopback transport at world 2, 3 and 4
(keys == the single-device keys bit for bit, cross-shard tie to the lower global row, one-call / counts-ahead / split forms), the RCCL
transport at world 1, and a shard cut from the resident keypoint table (apds_db_shard). The same choreography text over gloo at world
2 / 4 runs on the CPU in tests/test_sharded_matcher_cpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "cubesat-apds_amd")


def _build(tmp_path):
    exe = str(tmp_path / "shard_loopback_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shard_loopback_test.cpp"), "-o", exe, "-L", LIBDIR, "-lapds_hip",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return exe


def test_native_shard_host_compiles_and_links(pkg, tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_threads_as_ranks_equal_the_single_device_keys(gpu_pkg, tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("... ok") == 7 and "0 failed" in out.stdout
    assert "rccl version code" in out.stdout and "rccl version code 0" not in out.stdout


@pytest.mark.gpu
def test_random_worlds_over_the_loopback_transport(gpu_pkg, tmp_path):
    # `fuzz <cases> <seed>`: 1 .. 6 ranks, one row and up (shards without rows), query counts around the message rounding, k = 1 / 2,
    # every form (one call, counts ahead, split with two frames in flight) against the single-device keys. 3000 worlds of this seed found
    # the one bug of the kind so far (a rank destroying its events while a slower peer still waited on them, case 898).
    # (150 worlds here as a smoke run; the race itself is pinned deterministically below. profiles/r03/shard_loopback_fuzz.txt: 6000 worlds.)
    out = subprocess.run([_build(tmp_path), "fuzz", "150", "11"], capture_output=True, text=True, timeout=900)
    print(out.stdout[-3000:], out.stderr[-3000:])
    assert out.returncode == 0 and out.stdout.count("... ok") == 150 and "0 failed" in out.stdout


@pytest.mark.gpu
def test_a_lagging_rank_outlives_its_peers_shard(gpu_pkg, tmp_path):
    # the event-lifetime race of round 3 made deterministic: rank 0 is held for 300 ms in front of the closing event waits of every
    # collective (test hook APDS_TEST_LOOPBACK_LAG) while rank 1 destroys its shard; then: a second attachment of a live rank is refused
    # without harming the first
    env = dict(os.environ, APDS_TEST_LOOPBACK_LAG="0:300")
    out = subprocess.run([_build(tmp_path), "lag"], capture_output=True, text=True, timeout=300, env=env)
    print(out.stdout[-3000:], out.stderr[-3000:])
    assert out.returncode == 0 and "lag + double attach ... ok" in out.stdout and "0 failed" in out.stdout


@pytest.mark.gpu
def test_rccl_transport_under_torch_at_world_one(gpu_pkg):
    # python front (pipeline.ShardedMatcher) on a torch "nccl" group of one rank: the library's own RCCL communicator (id broadcast through
    # the torch group), one-call and split forms == the direct scan
    env = dict(os.environ, MASTER_PORT="29561")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_selftest.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    print(out.stdout, out.stderr[-3000:])
    assert out.returncode == 0 and "nccl selftest OK" in out.stdout
