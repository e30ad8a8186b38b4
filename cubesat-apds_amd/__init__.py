"""cubesat-apds_amd — MI355X (gfx950) implementation of the cubesat-APDS hot path.

AKAZE extraction -> Hamming brute-force match -> RANSAC homography, as hand-written HIP kernels behind the
C ABI in include/apds.h (libapds_hip.so). This package is the host-side mirror of the reference's two Rust
crates for that path:

    feature_extraction  (/root/reference/feature_extraction/src/lib.rs)
    homographier        (/root/reference/homographier/src/homographier/mod.rs)

There is no CPU fallback: importing works anywhere, but every compute call needs the HIP library and a GPU.
The directory name has a hyphen, so import it through `__graft_entry__.load_package()` (module name
`cubesat_apds_amd`).
"""
from . import _lib, feature_database, feature_extraction, geotiff_extractor, homographier, preprocessor, synth  # noqa: F401
from ._lib import ApdsError, lib, build_library  # noqa: F401

__all__ = ["feature_extraction", "homographier", "geotiff_extractor", "feature_database", "preprocessor", "synth", "ApdsError", "lib", "build_library"]
