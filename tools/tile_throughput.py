"""Tile-scheduler throughput: tiles/s of one level of a synthetic mosaic through to_rgb -> raster_to_mat -> AKAZE ->
resident keypoint table, for 1 .. 8 concurrent extraction workers (host-pointer C ABI: every tile crosses PCIe)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
ge, pp, fd = pkg.geotiff_extractor, pkg.preprocessor, pkg.feature_database
size, levels = 8192, 4                     # level 0: 64 tiles of 1024^2
t = pkg.synth.make_tile(size, size, frame_index=5, channels=3).astype(np.float32)
ds = ge.MosaicedDataset(np.stack([t[:, :, 2], t[:, :, 1], t[:, :, 0]]))
ds.datasets_min_max()
for workers, fused, batch in ((1, False, 1), (1, True, 1), (4, True, 1), (8, True, 1), (1, True, 4), (1, True, 8), (1, True, 16), (1, True, 32)):
    for rep in range(2):                   # first pass warms the threads' workspaces
        table, images = fd.KeypointTable(2_000_000), pp.ImageTable()
        t0 = time.perf_counter()
        out = pp.downscale_from_lod(table, images, ds, levels, 0, workers=workers, fused=fused, batch=batch)
        dt = time.perf_counter() - t0
        n = sum(k for _, k in out)
        table.close()
    print(f"workers {workers} fused {int(fused)} batch {batch:2d}: {len(out)} tiles of 1024^2 in {dt * 1e3:7.1f} ms = {len(out) / dt:6.1f} tiles/s, {n} keypoints", flush=True)
