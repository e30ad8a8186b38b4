"""GPU probe: whole Hamming top-2 call (sample pass + main scan + merges) for one (queries, rows) shape, e.g. the per-rank shape of an
N-GPU run (N x 35312 queries against 1e6 / N rows). Usage: match_shape_probe.py NQ NT"""
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
nq, nt = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
db = torch.randint(0, 256, (nt, 64), dtype=torch.uint8, device=dev, generator=g)
q = torch.randint(0, 256, (nq, 64), dtype=torch.uint8, device=dev, generator=g)
for t in (db, q):
    t[:, 60] &= 0x3F
    t[:, 61:] = 0
out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
check(L.apds_dev_timing_enable(1))
for rep in range(4):
    pkg._lib.kernel_ms("hamming_topk"), pkg._lib.kernel_ms("hamming_topk_sample")
    t0 = time.perf_counter()
    check(L.apds_dev_hamming_topk(q.data_ptr(), nq, db.data_ptr(), nt, 0, 2, out.data_ptr(), None))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ms, n = pkg._lib.kernel_ms("hamming_topk")
    sms, sn = pkg._lib.kernel_ms("hamming_topk_sample")
print(json.dumps({"nq": nq, "nt": nt, "wall_ms": round(wall, 3), "main_ms": round(ms, 3), "sample_ms": round(sms, 3), "rest_ms": round(wall - ms - sms, 3),
                  "Tpairs_per_s_wall": round(float(nq) * nt / wall / 1e9, 4)}))
