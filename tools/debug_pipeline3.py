import sys, numpy as np, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as g
import oracle
pkg = g.load_package(); fe = pkg.feature_extraction; synth = pkg.synth
from cubesat_apds_amd import pipeline as pl
L = pkg.lib(); check = pkg._lib.check
T = int(sys.argv[1])
dev = torch.device("cuda:0")
f = synth.make_tile(T, T, frame_index=0)
if len(sys.argv) > 2 and "roll" in sys.argv[2]:
    f = np.roll(f, (37, 52), axis=(0, 1)).copy()
DEVFIRST = len(sys.argv) > 2 and "devfirst" in sys.argv[2]
oracle.set_threads(16)
ref = oracle.akaze(f)
if not DEVFIRST:
    host = fe.akaze_keypoint_descriptor_extraction_def(f, None)
def cmp(name, kp, d):
    print(name, "K", len(kp), len(ref.keypoints))
    if len(kp) != len(ref.keypoints): return
    for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        bad = np.nonzero(kp[fld] != ref.keypoints[fld])[0]
        print("  ", fld, "mismatch", len(bad), bad[:5])
    bad = np.nonzero((d != ref.descriptors).any(1))[0]
    print("   desc mismatch rows", len(bad), bad[:10])
if not DEVFIRST:
    cmp("host-api", host.keypoints, host.descriptors)
cap = 262143
for rep in range(2):
    kps = torch.zeros((cap, 7), dtype=torch.float32, device=dev); desc = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
    ft = torch.from_numpy(f).to(dev); n = C.c_int(0)
    check(L.apds_dev_akaze_extract(ft.data_ptr(), T, T, 4, ft.stride(0), cap, kps.data_ptr(), desc.data_ptr(), cap, C.byref(n), pl.torch_stream()))
    torch.cuda.synchronize()
    K = n.value
    kp = np.frombuffer(kps[:K].cpu().numpy().tobytes(), dtype=pkg._lib.KEYPOINT_DTYPE)
    cmp(f"dev-api rep{rep}", kp, desc[:K].cpu().numpy()[:, :61])

if DEVFIRST:
    host = fe.akaze_keypoint_descriptor_extraction_def(f, None)
    cmp("host-api (after dev)", host.keypoints, host.descriptors)
