"""GPU probe: float-descriptor L2 top-2 as an MFMA distance GEMM (BASELINE config 3) — TFLOP/s vs the f32 MFMA peak."""
import ctypes as C
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
L = pkg.lib()
check = pkg._lib.check
F32_MFMA_PEAK = 157.3e12


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    from cubesat_apds_amd import pipeline as pl
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for nq, nt in ((16384, 100000), (131072, 1000000), (1048576, 1000000)):
        db = torch.nn.functional.normalize(torch.randn((nt, 128), device=dev, generator=g), dim=1)
        q = torch.nn.functional.normalize(torch.randn((nq, 128), device=dev, generator=g), dim=1)
        out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        check(L.apds_dev_timing_enable(1))
        for rep in range(2):
            t0 = time.perf_counter()
            check(L.apds_dev_l2_topk(q.data_ptr(), nq, db.data_ptr(), nt, 128, 0, 2, out.data_ptr(), pl.torch_stream()))
            ms, n = pkg._lib.kernel_ms("l2_topk")
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
        flops = 2.0 * nq * nt * 128
        print(json.dumps({"nq": nq, "nt": nt, "dim": 128, "kernel_ms": ms, "wall_ms": wall, "TFLOPs": flops / ms / 1e9,
                          "frac_of_f32_mfma_peak": flops / (ms * 1e-3) / F32_MFMA_PEAK, "Mqueries_per_s": nq / ms / 1e3}), flush=True)
        del db, q, out


if __name__ == "__main__":
    main()
