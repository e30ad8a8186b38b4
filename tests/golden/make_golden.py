"""Regenerates the golden fixtures in this directory from the CPU oracle (oracle/, a restatement: the reference cannot be
built or imported here — it is Rust over OpenCV). Fixtures are DATA: seeded inputs' expected outputs.

    python tests/golden/make_golden.py

akaze_256.npz      : synthetic 256x256 BGRA tile (frame 5) -> keypoints (cv::KeyPoint rows) + 61-byte descriptors
hamming_1k_4k.npz  : 1000 queries x 4000 train rows (seeded) -> top-2 indices and distances, ratio(0.3) and cross-check matches
homography_200.npz : 200 point pairs (40 % inliers) -> H (RANSAC, thr 3) and inlier mask
ingest.npz         : band_merger / warp_perspective expected bytes for seeded inputs
pnp_400.npz        : 400 3D-2D correspondences (60 % inliers) -> solvePnPRansac pose + inlier indices, EPnP and P3P
pnp_sqpnp_400.npz  : the same correspondences -> solvePnPRansac with SOLVEPNP_SQPNP (EPnP's consensus set, SQPnP's pose over it)
pnp_ippe_400.npz   : 400 correspondences of a tilted PLANAR target (60 % inliers) -> solvePnPRansac with SOLVEPNP_IPPE
world_coordinates.npz : 500 mosaic pixels -> ECEF through two geotransforms and a seeded elevation raster
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402

synth = graft.load_package().synth


def inputs():
    tile = synth.make_tile(256, 256, frame_index=5)
    db = synth.make_descriptor_db(4000, seed=0x44420001 + 4000)
    q, _ = synth.make_queries(db, 1000, seed=0x51550001 + 1000)
    src, dst, _, _ = synth.make_ransac_set(200, seed=0x52410001 + 200)
    rng = np.random.default_rng(20261003)
    bands = [rng.normal(0.4, 0.3, 5000).astype(np.float32) for _ in range(3)]
    bands[0][::97] = np.nan
    bands[1][::97] = np.nan
    bands[2][::97] = np.nan
    bands[1][5::131] = np.nan
    mm = np.array([0.0017, 0.93, -0.2, 1.1, 0.05, 0.8])
    img = rng.integers(0, 256, (48, 64, 4), dtype=np.uint8)
    M = np.array([[0.9, -0.2, 6.0], [0.25, 1.1, -3.0], [1e-3, -2e-3, 1.0]])
    return tile, db, q, src, dst, bands, mm, img, M


def main():
    tile, db, q, src, dst, bands, mm, img, M = inputs()
    r = oracle.akaze(tile)
    np.savez_compressed(os.path.join(HERE, "akaze_256.npz"), keypoints=r.keypoints, descriptors=r.descriptors)
    idx, dist = oracle.knn_hamming(q, db, 2)
    np.savez_compressed(os.path.join(HERE, "hamming_1k_4k.npz"), idx=idx, dist=dist, ratio=oracle.get_knn_matches(q, db, 2, 0.3),
                        cross=oracle.get_bruteforce_matches(q, db))
    found, H, mask = oracle.find_homography(src, dst, 8, 3.0)
    np.savez_compressed(os.path.join(HERE, "homography_200.npz"), found=found, H=H, mask=mask)
    np.savez_compressed(os.path.join(HERE, "ingest.npz"), rgba=oracle.band_merger(*bands, mm), warped=oracle.warp_perspective(img, M))


def pnp_inputs():
    obj, img, K, _, _, _ = synth.make_pnp_set(400, seed=0x504E5000 + 400, inlier_frac=0.6, noise=0.5)
    return obj, img, K


def pnp_planar_inputs():
    """pnp_inputs() pressed onto a tilted plane off the origin; the inliers' pixels re-projected from the planted pose (+ 0.5 px noise)."""
    obj, img, K, rvec, tvec, flag = synth.make_pnp_set(400, seed=0x504E5000 + 401, inlier_frac=0.6, noise=0.5)
    rng = np.random.default_rng(0x1BBE)

    def rot(v):
        th = np.linalg.norm(v)
        k = v / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    obj = obj.copy()
    obj[:, 2] = 0.0
    obj = obj @ rot(np.array([0.3, -0.2, 0.5])).T + np.array([3.0, -2.0, 5.0])
    cam = obj[flag] @ rot(rvec).T + tvec
    img = img.copy()
    img[flag] = cam[:, :2] / cam[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, 0.5, (int(flag.sum()), 2))
    return np.ascontiguousarray(obj), np.ascontiguousarray(img), K


def world_inputs():
    rng = np.random.default_rng(20261004)
    dgt = [9.0, 1e-4, 0.0, 57.0, 0.0, -1e-4]
    egt = [8.99, 3e-4, 1e-7, 57.01, 2e-7, -3e-4]
    yy, xx = np.mgrid[0:300, 0:400]
    elevation = 80 + 60 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.uniform(0, 3, (300, 400))
    xy = rng.uniform(0, 700, (500, 2))
    return xy, dgt, egt, elevation


def main_extra():
    obj, img, K = pnp_inputs()
    out = {}
    for name, method in (("epnp", 1), ("p3p", 2)):
        rc, r, t, idx = oracle.solve_pnp_ransac(obj, img, K, 500, 3.0, 0.99, method)
        out.update({name + "_rc": rc, name + "_rvec": r, name + "_tvec": t, name + "_inliers": idx})
    np.savez_compressed(os.path.join(HERE, "pnp_400.npz"), **out)
    rc, r, t, idx = oracle.solve_pnp_ransac(obj, img, K, 500, 3.0, 0.99, 8)
    np.savez_compressed(os.path.join(HERE, "pnp_sqpnp_400.npz"), sqpnp_rc=rc, sqpnp_rvec=r, sqpnp_tvec=t, sqpnp_inliers=idx)
    obj, img, K = pnp_planar_inputs()
    rc, r, t, idx = oracle.solve_pnp_ransac(obj, img, K, 500, 3.0, 0.99, 6)
    np.savez_compressed(os.path.join(HERE, "pnp_ippe_400.npz"), ippe_rc=rc, ippe_rvec=r, ippe_tvec=t, ippe_inliers=idx)
    xy, dgt, egt, elevation = world_inputs()
    rc, xyz = oracle.world_coordinates(xy, dgt, egt, elevation)
    np.savez_compressed(os.path.join(HERE, "world_coordinates.npz"), rc=rc, xyz=xyz)


if __name__ == "__main__":
    main()
    main_extra()
