import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as g
pkg = g.load_package()
f = pkg.synth.make_tile(4096, 4096, frame_index=0)
ex = pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(f, None)
print("K", len(ex.keypoints))
