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
    rows = torch.from_numpy(db64).to(dev)
    qd = torch.from_numpy(q64).to(dev)
    direct = pl.ShardedMatcher(rows, 0).knn(qd, 2).cpu().numpy().view(np.uint64)          # no collectives
    # the collective path of the multi-GPU matcher on ONE rank, through the C ABI (apds_shard_*): ncclAllGather of the queries, the scan,
    # the send/recv group of the keys, the merge; first in one call, then in the split form the pipeline uses
    m = pl.ShardedMatcher(rows, 0, group=dist.group.WORLD, meta_group=meta, always_exchange=True)
    info = m.info()
    assert info["transport"] == "rccl" and info["world"] == 1 and info["rccl_version"] > 0, info
    cnt = m.exchange_counts(len(q))
    one_call = m.knn(qd, 2, counts=cnt).cpu().numpy().view(np.uint64).copy()
    bufs = [m.make_buffers(len(q)), m.make_buffers(len(q))]
    gs = torch.cuda.Stream(dev)
    with torch.cuda.stream(gs):
        m.gather_queries(qd, cnt, bufs[0])
    m.scan_gathered(bufs[0], 2)
    with torch.cuda.stream(gs):
        m.gather_queries(torch.flip(qd, dims=(0,)).contiguous(), cnt, bufs[1])      # the next frame's gather goes out before this frame's exchange
    split_a = m.exchange_merge(bufs[0], 2).cpu().numpy().view(np.uint64).copy()
    m.scan_gathered(bufs[1], 2)
    split_b = m.exchange_merge(bufs[1], 2).cpu().numpy().view(np.uint64).copy()
    # the same exchange through torch.distributed's own RCCL communicator (APDS_TRANSPORT_DEVICE: device callbacks on this group) - the
    # fallback transport of bench.py (--transport torch) - and the strong-scaling form (one frame, all-gather of the per-shard keys)
    mt = pl.ShardedMatcher(rows, 0, group=dist.group.WORLD, meta_group=meta, always_exchange=True, transport="torch")
    info_t = mt.info()
    assert info_t["transport"] == "device-callbacks" and info_t["world"] == 1, info_t
    torch_call = mt.knn(qd, 2).cpu().numpy().view(np.uint64).copy()
    rep = torch.empty((len(q), 2), dtype=torch.int64, device=dev)
    pkg._lib.check(pkg.lib().apds_shard_knn_replicated(mt.handle, qd.data_ptr(), len(q), 0, 2, rep.data_ptr(), pl.torch_stream()))
    rep_t = rep.cpu().numpy().view(np.uint64).copy()
    pkg._lib.check(pkg.lib().apds_shard_knn_replicated(m.handle, qd.data_ptr(), len(q), -1, 2, rep.data_ptr(), pl.torch_stream()))
    rep_r = rep.cpu().numpy().view(np.uint64).copy()
torch.cuda.synchronize()
assert np.array_equal(torch_call, direct) and np.array_equal(rep_t, direct) and np.array_equal(rep_r, direct)
assert cnt == [len(q)] and np.array_equal(one_call, direct) and np.array_equal(split_a, direct) and np.array_equal(split_b[::-1], direct)
planted = src >= 0
assert np.array_equal((direct[planted, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64), src[planted])
dist.barrier()
dist.destroy_process_group()
print(f"nccl selftest OK: the library's RCCL transport (rccl {info['rccl_version']}) at world 1 - ncclAllGather of the queries + send/recv group of the keys between "
      "library kernels, one-call and split forms - under a torch nccl group with a gloo meta group beside it, barrier; the same exchange and the "
      "replicated (all-gather of keys) form through torch.distributed device collectives (device-callback transport)")
