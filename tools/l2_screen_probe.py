"""L2 screen mode probe: candidates per query and mode used for a few shapes (unit-norm planted sets generated on the GPU)."""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
pl = importlib.import_module("cubesat-apds_amd.pipeline")
L, check = pkg._lib.lib(), pkg._lib.check
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
g = torch.Generator(device=dev)
g.manual_seed(7)
for nq, nt in ((65536, 200000), (262144, 1000000), (1048576, 1000000)):
    db = torch.nn.functional.normalize(torch.randn((nt, 128), device=dev, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn((nq, 128), device=dev, generator=g), dim=1)
    npl = int(0.3 * nq)
    src = torch.randint(0, nt, (npl,), device=dev, generator=g)
    q[:npl] = torch.nn.functional.normalize(db[src] + 0.05 * torch.randn((npl, 128), device=dev, generator=g), dim=1)
    out = torch.empty((nq, 2), dtype=torch.int64, device=dev)
    used, cpq = C.c_int(-1), C.c_double(0)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(L.apds_dev_l2_topk_ex(q.data_ptr(), nq, db.data_ptr(), nt, 128, 0, 2, 1, out.data_ptr(), pl.torch_stream(), C.byref(used), C.byref(cpq)))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    print(f"nq {nq} nt {nt}: mode_used {used.value}, candidates/query {cpq.value:.2f}, {ms:.1f} ms", flush=True)
    del db, q, out
