"""The streamed frame pipeline behind the C ABI (apds_pipeline_*, csrc/pipeline.cpp) driven by a host that is not Python:
tests/cpp/pipeline_test.cpp, built by g++ against libapds_hip.so (no torch, no HIP headers). CPU: it compiles and links, and the three
structs of the pipeline ABI have the layout the python front (ctypes) assumes. GPU: it runs - 24 + 9 frames streamed (device and host
frames alternating, blank frames among them), every result equal to what the one-call entry points give for that frame (keypoint,
match and inlier counts, H bit for bit), the counters and HIP-event timers filled, a destroy with frames in flight."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "cubesat-apds_amd")


def _build(tmp_path):
    exe = str(tmp_path / "pipeline_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "pipeline_test.cpp"), "-o", exe, "-L", LIBDIR, "-lapds_hip",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return exe


def test_native_pipeline_host_compiles_and_links(pkg, tmp_path):
    assert os.path.exists(_build(tmp_path))


def test_pipeline_struct_layouts_match_the_python_front(pkg, tmp_path):
    src = tmp_path / "layout.c"
    fields = {"apds_pipeline_params": [f[0] for f in pkg._lib.PipelineParams._fields_], "apds_frame_result": [f[0] for f in pkg._lib.FrameResult._fields_],
              "apds_pipeline_counters": [f[0] for f in pkg._lib.PipelineCounters._fields_]}
    lines = ["#include <stdio.h>", "#include <stddef.h>", "#include <apds.h>", "int main(void) {"]
    for st, names in fields.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for n in names:
            lines.append(f'printf("{st}.{n} %zu\\n", offsetof({st}, {n}));')
    lines += ["return 0; }"]
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(line.split() for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for st, cls in (("apds_pipeline_params", pkg._lib.PipelineParams), ("apds_frame_result", pkg._lib.FrameResult), ("apds_pipeline_counters", pkg._lib.PipelineCounters)):
        assert int(got[st]) == C.sizeof(cls), st
        for n, _t in cls._fields_:
            assert int(got[f"{st}.{n}"]) == getattr(cls, n).offset, (st, n)


@pytest.mark.gpu
def test_a_cpp_host_streams_frames_through_the_pipeline(gpu_pkg, tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr[-4000:])
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("... ok") == 2 and "0 failed" in out.stdout
