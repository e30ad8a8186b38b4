"""Host-side mirror of the pixel math of the reference crate `geotiff_extractor`
(/root/reference/geotiff_extractor/src/image_extractor/mod.rs) — the step right before the hot path.
GDAL I/O itself is out of scope; these functions take the f32 bands GDAL would deliver."""
import numpy as np

from . import _lib
from ._lib import check, lib, ptr

GAMMA_VALUE = np.float32(1.0) / np.float32(2.2)   # mod.rs:14
U8_MAX = np.float32(255)                           # mod.rs:15


class BandsMinMax:
    """mod.rs `BandsMinMax` (f64 fields)."""

    def __init__(self, red_min, red_max, green_min, green_max, blue_min, blue_max):
        self.red_min, self.red_max, self.green_min, self.green_max, self.blue_min, self.blue_max = (
            float(red_min), float(red_max), float(green_min), float(green_max), float(blue_min), float(blue_max))

    def as_array(self):
        return np.array([self.red_min, self.red_max, self.green_min, self.green_max, self.blue_min, self.blue_max], np.float64)


def band_merger(bands, min_max, bgra=False):
    """mod.rs:346-378 — bands = [red, green, blue] f32 arrays of equal length -> n x 4 u8 (RGBA8; BGRA when bgra=True,
    i.e. fused with homographier::raster_to_mat)."""
    r, g, b = (np.ascontiguousarray(x, np.float32).ravel() for x in bands)
    if not (len(r) == len(g) == len(b)):
        raise _lib.ApdsError(_lib.ERR_ASSERT, "bands differ in length")
    out = np.zeros((len(r), 4), np.uint8)
    mm = min_max.as_array()
    check(lib().apds_band_merger(ptr(r), ptr(g), ptr(b), len(r), ptr(mm), int(bgra), ptr(out)))
    return out


def f32_to_u8(input_value, mn, mx):
    """mod.rs:410-422 — returns the u8 or None where the reference returns Err (NaN, gamma out of range)."""
    v = np.float32(input_value)
    if np.isnan(v):
        return None
    with np.errstate(all="ignore"):
        f = (v - np.float32(mn)) / (np.float32(mx) - np.float32(mn))
    if not (0.0 <= f <= 1.0):
        return None
    out = band_merger([[v], [np.nan], [np.nan]], BandsMinMax(mn, mx, 0, 1, 0, 1))
    return int(out[0, 0])
