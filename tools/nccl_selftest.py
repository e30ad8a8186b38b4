"""RCCL smoke test on ONE GPU (world_size 1): process-group init with device_id, all_gather_into_tensor of device tensors on an
explicit stream interleaved with libapds kernels, a gloo meta group beside it, barrier, teardown. The multi-rank choreography is
covered by tests/test_sharded_matcher_cpu.py (gloo, world 2); this checks that the RCCL backend itself comes up in this image."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
meta = dist.new_group(backend="gloo")
pkg = graft.load_package()
from cubesat_apds_amd import pipeline as pl  # noqa: E402

pkg._lib.check(pkg.lib().apds_set_device(0))
db = pkg.synth.make_descriptor_db(50000)
q, src = pkg.synth.make_queries(db, 2000)
db64 = np.zeros((len(db), 64), np.uint8); db64[:, :61] = db
q64 = np.zeros((len(q), 64), np.uint8); q64[:, :61] = q
with torch.cuda.stream(torch.cuda.Stream(dev)):
    m = pl.ShardedMatcher(torch.from_numpy(db64).to(dev), 0, group=dist.group.WORLD, meta_group=meta)
    keys_direct = m.knn(torch.from_numpy(q64).to(dev), 2).cpu().numpy().view(np.uint64)
    # the collective path by hand, world 1
    qd = torch.from_numpy(q64).to(dev)
    g = torch.empty((1, len(q), 64), dtype=torch.uint8, device=dev)
    pl._gather_into(dist, dist.group.WORLD, g, qd)
    local = pl.HipBackend().topk(g[0].contiguous(), m.rows, 0, 2)
    parts = torch.empty((1, len(q), 2), dtype=torch.int64, device=dev)
    pl._gather_into(dist, dist.group.WORLD, parts, local)
    merged = pl.HipBackend().merge(parts, 2).cpu().numpy().view(np.uint64)
    cnt = m.exchange_counts(len(q))
torch.cuda.synchronize()
assert np.array_equal(merged, keys_direct) and cnt == [len(q)]
planted = src >= 0
assert np.array_equal((merged[planted, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64), src[planted])
dist.barrier()
dist.destroy_process_group()
print("nccl selftest OK: RCCL world-1 all_gather + kernels on one stream, gloo meta group, barrier")
