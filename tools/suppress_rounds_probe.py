import importlib, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("cubesat-apds_amd")
img = pkg.synth.make_tile(4096, 4096, frame_index=0)
e = pkg.feature_extraction.akaze_keypoint_descriptor_extraction_def(img, None)
print(len(e.keypoints))
