"""GPU probe: Hamming top-2 throughput vs the measured xor+popcount VALU ceiling (not part of the test suite)."""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
L = pkg.lib()
check = pkg._lib.check


def main():
    peak = C.c_double(0)
    check(L.apds_dev_valu_popcount_peak(C.byref(peak)))
    print(json.dumps({"valu_xor_bcnt_peak_Tlaneops": peak.value / 1e12, "pairs_per_s_ceiling_T": peak.value / 32 / 1e12}))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for nq, nt in ((20000, 1000000), (20000, 10000000), (5000, 100000), (100000, 1000000), (262143, 262143)):
        db = torch.randint(0, 256, (nt, 64), dtype=torch.uint8, device=dev, generator=g)
        q = torch.randint(0, 256, (nq, 64), dtype=torch.uint8, device=dev, generator=g)
        for t in (db, q):          # 486-bit M-LDB rows: bits 486.. of the 64-byte line are zero
            t[:, 60] &= 0x3F
            t[:, 61:] = 0
        out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        check(L.apds_dev_timing_enable(1))
        for rep in range(3):
            t0 = time.perf_counter()
            check(L.apds_dev_hamming_topk(q.data_ptr(), nq, db.data_ptr(), nt, 0, 2, out.data_ptr(), None))
            ms, n = pkg._lib.kernel_ms("hamming_topk")   # syncs on the events
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
        pairs = float(nq) * nt
        print(json.dumps({"nq": nq, "nt": nt, "kernel_ms": ms, "launches": n, "wall_ms": wall, "Tpairs_per_s_kernel": pairs / ms / 1e9,
                          "frac_of_valu_peak": pairs * 32 / (ms * 1e-3) / peak.value,
                          "hbm_GBps_algorithmic": (64.0 * nt + 64.0 * nq + 16.0 * nq) / ms / 1e6}))
        del db, q, out


if __name__ == "__main__":
    main()
