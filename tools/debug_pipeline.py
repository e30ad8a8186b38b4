import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as g
pkg = g.load_package(); fe = pkg.feature_extraction; synth = pkg.synth
T = 1024
f = synth.make_tile(T, T, frame_index=0)
r = np.roll(f, (37, 52), axis=(0, 1)).copy()
a = fe.akaze_keypoint_descriptor_extraction_def(f, None)
b = fe.akaze_keypoint_descriptor_extraction_def(r, None)
print("K", len(a.keypoints), len(b.keypoints))
idx, dist = fe.knn_match(a.descriptors, b.descriptors, 2)
print("dist0 hist", np.bincount(np.minimum(dist[:, 0] // 20, 15)))
m = fe.get_knn_matches(a.descriptors, b.descriptors, 2, 0.3)
print("matches", len(m))
ka, kb = a.keypoints, b.keypoints
dx = kb["x"][idx[:, 0]] - ka["x"]; dy = kb["y"][idx[:, 0]] - ka["y"]
good = (np.abs(dx - 52) < 1.5) & (np.abs(dy - 37) < 1.5)
print("geometrically right 1-NN:", good.sum(), "by octave", [(o, int(good[ka['octave']==o].sum()), int((ka['octave']==o).sum())) for o in range(4)])
print("dist of right ones", np.bincount(np.minimum(dist[good, 0] // 20, 15)))
