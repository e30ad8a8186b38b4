"""Host-side mirror of the pixel math of the reference crate `geotiff_extractor`
(/root/reference/geotiff_extractor/src/image_extractor/mod.rs) — the step right before the hot path.
GDAL I/O itself is out of scope; these functions take the f32 bands GDAL would deliver."""
import struct

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


# ---- minimal TIFF I/O for synthetic mosaics (SURVEY §8f-2 "optional minimal TIFF reader/writer") ---------------------------
# Baseline TIFF, little endian, uncompressed, 32-bit IEEE float samples, strips; chunky or planar. Enough to put a synthetic
# three-band mosaic on disk and read it back the way `MosaicedDataset::to_rgb` consumes a GDAL dataset. Compressed / tiled /
# BigTIFF files (what GDAL's COG driver writes, mod.rs:140-160) are not read: GDAL I/O stays out of scope.
def write_tiff_f32(path, bands):
    """bands: array [n_bands, H, W] (or a list of HxW arrays) of float32 -> one-strip chunky TIFF."""
    b = np.ascontiguousarray(np.stack([np.asarray(x, np.float32) for x in bands]), np.float32)
    nb, h, w = b.shape
    data = np.ascontiguousarray(np.moveaxis(b, 0, 2)).tobytes()          # chunky: H x W x bands
    if len(data) >= 2 ** 32 - 1024:
        raise ValueError("classic TIFF is limited to 4 GiB")
    entries = []

    def tag(code, typ, values):
        entries.append((code, typ, list(values)))

    tag(256, 4, [w]); tag(257, 4, [h]); tag(258, 3, [32] * nb); tag(259, 3, [1])
    tag(262, 3, [2 if nb >= 3 else 1]); tag(273, 4, [0]); tag(277, 3, [nb]); tag(278, 4, [h])
    tag(279, 4, [len(data)]); tag(284, 3, [1]); tag(339, 3, [3] * nb)
    if nb > 3:
        tag(338, 3, [0] * (nb - 3))                                       # ExtraSamples: unspecified
    entries.sort()
    ifd_off = 8
    ifd_size = 2 + 12 * len(entries) + 4
    extra_off = ifd_off + ifd_size
    extra = b""
    body = b""
    fixups = {}
    for code, typ, vals in entries:
        fmt = "<" + {3: "H", 4: "I"}[typ] * len(vals)
        raw = struct.pack(fmt, *vals)
        if len(raw) <= 4:
            field = raw.ljust(4, b"\0")
        else:
            field = struct.pack("<I", extra_off + len(extra))
            extra += raw + (b"\0" if len(raw) % 2 else b"")
        fixups[code] = len(body) + 8
        body += struct.pack("<HHI", code, typ, len(vals)) + field
    data_off = extra_off + len(extra)
    body = bytearray(body)
    body[fixups[273]:fixups[273] + 4] = struct.pack("<I", data_off)       # StripOffsets
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        f.write(struct.pack("<H", len(entries)) + bytes(body) + struct.pack("<I", 0))
        f.write(extra)
        f.write(data)


def read_tiff_f32(path):
    """-> float32 array [n_bands, H, W]. Raises ValueError for anything but uncompressed float32 strips."""
    raw = np.fromfile(path, np.uint8)
    view = memoryview(raw)

    def rd(fmt, off):
        return struct.unpack_from(fmt, view, off)

    if bytes(view[0:2]) != b"II" or rd("<H", 2)[0] != 42:
        raise ValueError("not a little-endian classic TIFF")
    ifd = rd("<I", 4)[0]
    tags = {}
    for i in range(rd("<H", ifd)[0]):
        code, typ, cnt = rd("<HHI", ifd + 2 + 12 * i)
        size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 11: 4, 12: 8}.get(typ)
        if size is None:
            continue
        off = ifd + 2 + 12 * i + 8
        if size * cnt > 4:
            off = rd("<I", off)[0]
        if typ in (3, 4):
            tags[code] = list(rd("<" + ("H" if typ == 3 else "I") * cnt, off))
    w, h = tags[256][0], tags[257][0]
    nb = tags.get(277, [1])[0]
    if tags.get(259, [1])[0] != 1 or set(tags.get(258, [0])) != {32} or set(tags.get(339, [1])) != {3} or 322 in tags:
        raise ValueError("only uncompressed 32-bit float strip TIFFs are read (GDAL I/O is out of scope)")
    offs, cnts = tags[273], tags[279]
    blob = np.concatenate([raw[o:o + c] for o, c in zip(offs, cnts)]).view(np.float32)
    if tags.get(284, [1])[0] == 2:
        return np.ascontiguousarray(blob.reshape(nb, h, w))
    return np.ascontiguousarray(np.moveaxis(blob.reshape(h, w, nb), 2, 0))


class MosaicedDataset:
    """mod.rs `MosaicedDataset` + the `Datasets` methods the preprocessor uses (mod.rs:200-269, 279-288), over an in-memory
    [3, H, W] float32 raster instead of a GDAL handle."""

    def __init__(self, bands):
        self.bands = np.ascontiguousarray(bands, np.float32)
        if self.bands.ndim != 3 or self.bands.shape[0] < 3:
            raise ValueError("bands: [>=3, H, W] float32")
        self.min_max = None

    @classmethod
    def import_mosaic_dataset(cls, path):
        """mod.rs:279-288"""
        return cls(read_tiff_f32(path))

    def raster_size(self):
        """(width, height), as gdal's Dataset::raster_size (preprocessor/src/main.rs:207)"""
        return self.bands.shape[2], self.bands.shape[1]

    def datasets_min_max(self):
        """mod.rs:200-229 — per-band minimum and maximum (NaN ignored), cached."""
        if self.min_max is None:
            with np.errstate(all="ignore"):
                mm = [(float(np.nanmin(self.bands[i])), float(np.nanmax(self.bands[i]))) for i in range(3)]
            self.min_max = BandsMinMax(mm[0][0], mm[0][1], mm[1][0], mm[1][1], mm[2][0], mm[2][1])
        return self.min_max

    def to_rgb(self, window, window_size, size):
        """mod.rs:241-269 — window (x, y) and window_size (w, h) in raster pixels, resampled to size (w, h), then band_merger.
        Returns size[0]*size[1] RGBA8 pixels. The reference resamples with GDAL's Lanczos (mod.rs:339), which is not restated:
        equal sizes are copied, anything else is decimated by nearest neighbour ((i + 0.5) * window / size, floored)."""
        x0, y0 = int(window[0]), int(window[1])
        ww, wh = int(window_size[0]), int(window_size[1])
        ow, oh = int(size[0]), int(size[1])
        W, H = self.raster_size()
        if x0 < 0 or y0 < 0 or ww <= 0 or wh <= 0 or x0 + ww > W or y0 + wh > H or ow <= 0 or oh <= 0:
            raise _lib.ApdsError(_lib.ERR_OUT_OF_RANGE, "window outside the raster")
        win = self.window(window, window_size, size)
        return band_merger([np.ascontiguousarray(win[i]).ravel() for i in range(3)], self.datasets_min_max())

    def window(self, window, window_size, size):
        """The three f32 band windows `to_rgb` merges ([3, size_h, size_w]; a strided view of the mosaic when no resampling is needed)."""
        x0, y0 = int(window[0]), int(window[1])
        ww, wh = int(window_size[0]), int(window_size[1])
        ow, oh = int(size[0]), int(size[1])
        W, H = self.raster_size()
        if x0 < 0 or y0 < 0 or ww <= 0 or wh <= 0 or x0 + ww > W or y0 + wh > H or ow <= 0 or oh <= 0:
            raise _lib.ApdsError(_lib.ERR_OUT_OF_RANGE, "window outside the raster")
        win = self.bands[:3, y0:y0 + wh, x0:x0 + ww]
        if (ow, oh) != (ww, wh):
            ys = np.minimum(((np.arange(oh) + 0.5) * (wh / oh)).astype(np.int64), wh - 1)
            xs = np.minimum(((np.arange(ow) + 0.5) * (ww / ow)).astype(np.int64), ww - 1)
            win = win[:, ys][:, :, xs]
        return win
