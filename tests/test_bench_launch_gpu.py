"""GPU: bench.py started exactly as the driver starts it.

`python3 bench.py --gpus N` with no launcher around it must spawn its own ranks (child processes of a parent that never touched the
GPU), print ONE JSON line and exit 0. On the one-GPU box the ranks share the card and the collectives go over gloo
(APDS_BENCH_BACKEND=gloo: same choreography, threads, streams and buffers as over RCCL); on an 8-GPU node the same command line with the
default backend is BASELINE config 5. The N = 1 line is checked for the fields the contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout,
                       cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


def test_bare_two_rank_launch_prints_one_json_line(gpu_pkg):
    rc, lines, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--tile", "1024", "--db-rows", "100000", "--no-cpu-baseline"],
                          {"APDS_BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["homography_found"] is True
    assert out["config"]["db_rows_per_gpu"] == 50000 and out["config"]["frames_per_step"] == 2
    assert out["collectives"]["world"] == 2 and out["collectives"]["ranks_in_group"] == 2 and out["collectives"]["backend"] == "gloo"
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


def test_a_failing_first_transport_is_retried_on_the_second(gpu_pkg):
    """The bare parent starts a fresh set of ranks with --transport torch when the first set (the library's own RCCL communicator) exits
    non-zero; the JSON line says which transport ran and what it fell back from. (Over gloo both map onto host callbacks: what is under
    test is the parent's retry, which never touches the GPU.)"""
    rc, lines, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--tile", "1024", "--db-rows", "100000", "--no-cpu-baseline", "--no-host-frames"],
                          {"APDS_BENCH_BACKEND": "gloo", "APDS_BENCH_FAIL_TRANSPORT": "rccl"})
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["collectives"]["transport_requested"] == "torch" and "rccl" in out["collectives"]["fell_back_from"]
    assert out["n_gpus"] == 2 and out["config"]["homography_found"] is True
    assert "starting a fresh set with --transport torch" in err


def test_single_gpu_line_carries_the_contract_fields(gpu_pkg):
    rc, lines, err = _run(["--steps", "3", "--warmup", "1", "--tile", "1024", "--db-rows", "100000", "--no-cpu-baseline", "--host-frames"])
    assert rc == 0, err[-3000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["collectives"]["world"] == 1
    assert out["value_host_frames"] and out["value_host_frames"] > 0          # frames uploaded from pinned host memory every step
    assert out["config"]["homography_found"] is True
