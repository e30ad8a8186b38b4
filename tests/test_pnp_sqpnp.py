"""CPU: the product's SQPnP (csrc/sqpnp_core.h through apds_pnp_sqpnp - host arithmetic, no device) against the oracle's separate
restatement (oracle/pnp_oracle.cpp), bit for bit: both evaluate the same IEEE double operations in the same order. Inputs cover what the
final solvePnP of pnp_solver_ransac(SOLVEPNP_SQPNP) can meet (mod.rs:327,359): few and many points, pixel noise up to gross, coplanar,
nearly flat and collinear object points (the last have no pose: both sides must say so). What the pose must BE is held by
tests/test_external_anchors.py; PARITY UNPINNED against OpenCV (no OpenCV in this image)."""
import ctypes as C

import numpy as np


def _rot(rv):
    th = np.linalg.norm(rv)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def _product(pkg, obj, img, K):
    obj, img, K = (np.ascontiguousarray(a, np.float64) for a in (obj, img, K))
    r, t, found = np.zeros(3), np.zeros(3), C.c_int(-1)
    rc = pkg.lib().apds_pnp_sqpnp(pkg._lib.ptr(obj), pkg._lib.ptr(img), len(obj), pkg._lib.ptr(K), pkg._lib.ptr(r), pkg._lib.ptr(t), C.byref(found))
    assert rc == 0, pkg.lib().apds_last_error()
    return found.value, r, t


def test_product_sqpnp_equals_the_oracle_bit_for_bit(pkg, oracle_mod):
    rng = np.random.default_rng(2026)
    K = np.array([[800.0, 0, 320], [0, 820, 240], [0, 0, 1]])
    poses = 0
    for trial in range(600):
        n = int(rng.choice([3, 4, 5, 6, 8, 20, 100, 1000]))
        rv = rng.normal(size=3) * rng.choice([0.01, 0.5, 1.5, 3.0])
        t = np.array([0.2, -0.1, 6.0]) + rng.normal(size=3) * 0.5
        obj = rng.uniform(-1, 1, size=(n, 3))
        shape = trial % 6
        if shape == 1:
            obj[:, 2] = 0.3                                   # a plane off the origin
        elif shape == 2:
            obj = obj * [1, 1, 1e-3]                          # nearly flat
        elif shape == 3:
            obj[:, 1], obj[:, 2] = obj[:, 0] * 0.5, obj[:, 0] * -0.2   # a line
        elif shape == 4:
            obj = obj @ _rot(rng.normal(size=3)).T * [1, 1, 0] @ _rot(rng.normal(size=3)) + 0.4   # a tilted plane
        cam = obj @ _rot(rv).T + t
        img = np.stack([cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2], cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]], 1)
        img = img + rng.normal(size=img.shape) * rng.choice([0, 0.1, 2.0, 30.0])
        found, r, tt = _product(pkg, obj, img, K)
        rc, ro, to = oracle_mod.solve_pnp_sqpnp(obj, img, K)
        assert found == rc, (trial, n, shape)
        if found:
            poses += 1
            assert np.array_equal(r, ro) and np.array_equal(tt, to), (trial, n, shape, r - ro, tt - to)
    assert poses >= 450


def test_product_ippe_equals_the_oracle_bit_for_bit(pkg, oracle_mod):
    """The same for SOLVEPNP_IPPE (csrc/ippe_core.h through apds_pnp_ippe against the oracle's restatement): planes through and off the
    origin, tilted planes, planes whose first three points are collinear (the SVD branch), clouds that are not planar and lines (no pose:
    both sides must say so)."""
    rng = np.random.default_rng(2027)
    K = np.array([[800.0, 0, 320], [0, 820, 240], [0, 0, 1]])
    poses = 0
    for trial in range(600):
        n = int(rng.choice([4, 5, 6, 8, 20, 100, 1000]))
        rv = rng.normal(size=3) * rng.choice([0.01, 0.5, 1.2])
        t = np.array([0.2, -0.1, 6.0]) + rng.normal(size=3) * 0.5
        obj = rng.uniform(-1, 1, size=(n, 3))
        obj[:, 2] = 0
        shape = trial % 6
        if shape == 1:
            obj[:, 2] = 0.3
        elif shape == 2:
            obj = obj @ _rot(rng.normal(size=3)).T + rng.normal(size=3)
        elif shape == 3:
            obj[:, 2] = rng.normal(size=n) * 0.02
        elif shape == 4:
            obj[1] = obj[0] + (obj[2] - obj[0]) * 0.5
            obj = obj @ _rot(rng.normal(size=3)).T + 0.2
        elif shape == 5:
            obj[:, 1] = obj[:, 0] * 0.5
        cam = obj @ _rot(rv).T + t
        img = np.stack([cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2], cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]], 1)
        img = img + rng.normal(size=img.shape) * rng.choice([0, 0.1, 2.0, 30.0])
        obj, img = np.ascontiguousarray(obj), np.ascontiguousarray(img)
        r, tt, found = np.zeros(3), np.zeros(3), C.c_int(-1)
        rc = pkg.lib().apds_pnp_ippe(pkg._lib.ptr(obj), pkg._lib.ptr(img), n, pkg._lib.ptr(K), pkg._lib.ptr(r), pkg._lib.ptr(tt), C.byref(found))
        assert rc == 0, pkg.lib().apds_last_error()
        orc, ro, to = oracle_mod.solve_pnp_ippe(obj, img, K)
        assert found.value == orc, (trial, n, shape)
        if orc:
            poses += 1
            assert np.array_equal(r, ro, equal_nan=True) and np.array_equal(tt, to, equal_nan=True), (trial, n, shape, r - ro, tt - to)
    assert poses >= 350


def test_sqpnp_argument_checks(pkg):
    K = np.eye(3)
    obj, img = np.zeros((2, 3)), np.zeros((2, 2))
    found = C.c_int(0)
    r, t = np.zeros(3), np.zeros(3)
    rc = pkg.lib().apds_pnp_sqpnp(pkg._lib.ptr(obj), pkg._lib.ptr(img), 2, pkg._lib.ptr(K), pkg._lib.ptr(r), pkg._lib.ptr(t), C.byref(found))
    assert rc == pkg._lib.ERR_ASSERT and found.value == 0
    rc = pkg.lib().apds_pnp_sqpnp(None, pkg._lib.ptr(img), 5, pkg._lib.ptr(K), pkg._lib.ptr(r), pkg._lib.ptr(t), C.byref(found))
    assert rc == pkg._lib.ERR_BAD_ARG
