"""Where one preprocessor tile's host time goes (serial): window view, fused extraction call, ordered store."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
ge, pp, fd, fe = pkg.geotiff_extractor, pkg.preprocessor, pkg.feature_database, pkg.feature_extraction
size = 8192
t = pkg.synth.make_tile(size, size, frame_index=5, channels=3).astype(np.float32)
ds = ge.MosaicedDataset(np.stack([t[:, :, 2], t[:, :, 1], t[:, :, 0]]))
mm = ds.datasets_min_max()
tile = (1024, 1024)
for rep in range(2):
    table, images = fd.KeypointTable(2_000_000), pp.ImageTable()
    tw = te = ts = 0.0
    for i in range(8):
        for j in range(8):
            a = time.perf_counter()
            win = ds.window((j * 1024, i * 1024), tile, tile)
            b = time.perf_counter()
            k = fe.tile_keypoint_descriptor_extraction(win[0], win[1], win[2], mm, None)
            c = time.perf_counter()
            pp.store_tile(table, images, k, tile, j, i, 0)
            d = time.perf_counter()
            tw += b - a; te += c - b; ts += d - c
    table.close()
print(f"per tile: window {tw / 64 * 1e3:.3f} ms, fused extract call {te / 64 * 1e3:.3f} ms, store {ts / 64 * 1e3:.3f} ms")
