"""The C++ host mirror of the reference crates (include/apds.hpp) and the reference's own unit tests restated against it
(tests/cpp/reference_tests.cpp). CPU: the program compiles and links against libapds_hip.so. GPU: it runs and every test passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "cubesat-apds_amd")


def _build(tmp_path):
    exe = str(tmp_path / "reference_tests")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-o", exe, "-L", LIBDIR, "-lapds_hip",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_links(pkg, tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_reference_tests_in_cpp(gpu_pkg, tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" ... ok") == 10 and "0 failed" in out.stdout
