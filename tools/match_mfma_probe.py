"""GPU probe: Hamming top-2 on the FP4 matrix pipe (APDS_MATCH_MFMA=1, hamming_mfma.hip) against the vector-ALU kernel: keys equal, wall time
per call. Run once per setting (the switch is read once per process):  APDS_MATCH_MFMA=0|1 python3 tools/match_mfma_probe.py [out.npy]"""
import ctypes as C
import json
import os
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
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    mode = os.environ.get("APDS_MATCH_MFMA", "1")
    keep = {}
    for nq, nt in ((35312, 983616), (5000, 100000), (1000, 3000), (262143, 262143), (20000, 10000000), (282496, 125000)):   # the last: one rank's scan of an 8-GPU run
        db = torch.randint(0, 256, (nt, 64), dtype=torch.uint8, device=dev, generator=g)
        q = torch.randint(0, 256, (nq, 64), dtype=torch.uint8, device=dev, generator=g)
        for t in (db, q):
            t[:, 60] &= 0x3F
            t[:, 61:] = 0
        q[: min(nq, 2000)] = db[: min(nq, 2000)]            # exact hits and, with the duplicates below, ties
        db[nt // 2: nt // 2 + 500] = db[:500]
        out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            check(L.apds_dev_hamming_topk(q.data_ptr(), nq, db.data_ptr(), nt, 0, 2, out.data_ptr(), None))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        pairs = float(nq) * nt
        print(json.dumps({"mfma": mode, "nq": nq, "nt": nt, "ms": round(best, 3), "Tpairs_per_s": round(pairs / best / 1e9, 3),
                          "PFLOPs_fp4_equiv": round(pairs * 1024 / best / 1e12, 3)}), flush=True)
        keep[f"{nq}x{nt}"] = out.cpu().numpy()
        del db, q, out
    if len(sys.argv) > 1:
        np.savez(sys.argv[1], **keep)


if __name__ == "__main__":
    main()
