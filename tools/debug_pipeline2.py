import sys, numpy as np, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as g
pkg = g.load_package(); fe = pkg.feature_extraction; synth = pkg.synth
from cubesat_apds_amd import pipeline as pl
L = pkg.lib(); check = pkg._lib.check
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NDB = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
f = synth.make_tile(T, T, frame_index=0)
r = np.roll(f, (37, 52), axis=(0, 1)).copy()
cap = 262143
kps = torch.empty((cap, 7), dtype=torch.float32, device=dev); desc = torch.empty((cap, 64), dtype=torch.uint8, device=dev)
rt = torch.from_numpy(r).to(dev); n = C.c_int(0)
check(L.apds_dev_akaze_extract(rt.data_ptr(), T, T, 4, rt.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
P = n.value
rows = desc[:P].clone(); xy = kps[:P, 0:2].clone()
host = fe.akaze_keypoint_descriptor_extraction_def(r, None)
print("dev K", P, "host K", len(host.keypoints), "desc equal", np.array_equal(rows.cpu().numpy()[:, :61], host.descriptors), "pad zero", int(rows[:, 61:].sum()))
rnd = synth.make_descriptor_db(NDB - P, seed=synth.DB_SEED + P); pad = np.zeros((NDB - P, 64), np.uint8); pad[:, :61] = rnd
db = torch.cat([rows, torch.from_numpy(pad).to(dev)]).contiguous()
ft = torch.from_numpy(f).to(dev)
check(L.apds_dev_akaze_extract(ft.data_ptr(), T, T, 4, ft.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
K = n.value
keys = pl.HipBackend().topk(desc[:K], db, 0, 2)
torch.cuda.synchronize()
k = keys.cpu().numpy().astype(np.uint64)
d0 = (k[:, 0] >> np.uint64(32)).astype(np.int64); i0 = (k[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
d1 = (k[:, 1] >> np.uint64(32)).astype(np.int64)
print("K", K, "d0 hist", np.bincount(np.minimum(d0 // 20, 15)), "best in planted:", (i0 < P).sum())
print("d1 hist", np.bincount(np.minimum(d1 // 20, 15)))
qa = fe.akaze_keypoint_descriptor_extraction_def(f, None)
oi, od = fe.knn_match(qa.descriptors, db.cpu().numpy()[:, :61], 2)
print("host-api equal:", np.array_equal(od[:, 0], d0), np.array_equal(oi[:, 0], i0), np.array_equal(od[:, 1], d1))
print("ratio pass", (d0.astype(np.float32) < d1.astype(np.float32) * 0.3).sum())
