"""CPU: the C-ABI library loads and exports every symbol include/apds.h declares; argument errors that are
detected before any device work behave like the reference. No compute calls (no GPU here)."""
import os
import re

import numpy as np
import pytest


def test_header_symbols_are_exported(pkg):
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "apds.h")).read()
    declared = set(re.findall(r"\b(apds_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg._lib.SYMBOLS), declared ^ set(pkg._lib.SYMBOLS)
    L = pkg.lib()
    for name in declared:
        assert hasattr(L, name), name
    # ... and nothing else: the dynamic symbol table's apds_* functions are exactly the header's (no undeclared entry points)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("apds_")}
    assert exported == declared, exported ^ declared


def test_struct_layouts(pkg):
    assert pkg._lib.KEYPOINT_DTYPE.itemsize == 28     # cv::KeyPoint
    assert pkg._lib.DMATCH_DTYPE.itemsize == 16       # cv::DMatch
    assert pkg.feature_extraction.MAX_POINTS == 262143  # lib.rs:12-13


def test_no_cpu_fallback(pkg):
    if pkg.lib().apds_device_count() > 0:
        pytest.skip("GPU present")
    q = np.zeros((4, 61), np.uint8)
    with pytest.raises(pkg.ApdsError) as e:
        pkg.feature_extraction.get_knn_matches(q, q, 2, 0.3)
    assert e.value.code == -216


def test_argument_errors_before_device_work(pkg):
    fe = pkg.feature_extraction
    q = np.zeros((4, 61), np.uint8)
    # empty sides: Ok(empty) without touching the device (knnMatch yields no rows)
    assert len(fe.get_knn_matches(q[:0], q, 2, 0.3)) == 0
    assert len(fe.get_knn_matches(q, q[:0], 2, 0.3)) == 0
    assert len(fe.get_bruteforce_matches(q[:0], q)) == 0
    with pytest.raises(pkg.ApdsError) as e:          # lib.rs:108 `i.get(1)?`
        fe.get_knn_matches(q, q, 1, 0.3)
    assert e.value.code == -211
    with pytest.raises(pkg.ApdsError) as e:
        fe.get_knn_matches(q, q[:1], 2, 0.3)
    assert e.value.code == -211
    with pytest.raises(pkg.ApdsError) as e:
        fe.get_knn_matches(q, q, 0, 0.3)
    assert e.value.code == -215


def test_cmat_semantics(pkg):
    hg = pkg.homographier
    # mod.rs:475-477 cmat_init: empty Mat is an error
    with pytest.raises(hg.MatError):
        hg.Cmat(np.zeros((0, 0, 4), np.uint8), np.uint8, 4)
    # mod.rs:515-553 cmat_from_slice: row-major
    rows = [[[(1 + j) * c for c in (1, 2, 3, 4)] for j in range(4)] for _ in range(4)]
    cm = hg.Cmat.from_2d_slice(rows, np.uint8, 4)
    assert tuple(cm.mat[0, 0]) == (1, 2, 3, 4) and tuple(cm.mat[3, 3]) == (4, 8, 12, 16)
    # mod.rs:606-625 cmat_at_2d_works: (3,5) and (5,3) are StsOutOfRange (-211), (3,3) is readable
    for rc in ((3, 5), (5, 3)):
        with pytest.raises(hg.MatError) as e:
            cm.at_2d(*rc)
        assert e.value.kind == "Opencv" and e.value.inner.code == -211
    assert tuple(cm.at_2d(3, 3)) == (4, 8, 12, 16)
    with pytest.raises(hg.MatError) as e:          # type mismatch -> MatError::Empty (mod.rs:115-118)
        hg.Cmat(np.zeros((2, 2), np.float32), np.float64)
    assert e.value.kind == "Empty"


def test_raster_to_mat_length_check(pkg):
    with pytest.raises(pkg.homographier.MatError) as e:    # mod.rs:185-187
        pkg.homographier.raster_to_mat(np.zeros((15, 4), np.uint8), 4, 4)
    assert e.value.kind == "Unknown"
