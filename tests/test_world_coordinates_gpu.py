"""GPU: get_world_coordinates (feature_database/src/elevationdb.rs:64-104), batched kernel vs the oracle: bit-identical (same IEEE f64
operations, fixed sin/cos polynomials). Then the chain it exists for: DB keypoint pixels -> ECEF object points -> pnp_solver_ransac."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DGT = [9.0, 1e-4, 0, 57.0, 0, -1e-4]
EGT = [8.99, 3e-4, 0, 57.01, 0, -3e-4]


def _table(pkg, seed=4):
    t = pkg.feature_database.ElevationTable()
    t.create_geotransform("dataset", DGT)
    t.create_geotransform("elevation", EGT)
    yy, xx = np.mgrid[0:500, 0:600]
    t.add_elevation_data(80 + 60 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + np.random.default_rng(seed).uniform(0, 3, (500, 600)))
    return t


def test_batch_equals_oracle(gpu_pkg, oracle_mod):
    t = _table(gpu_pkg)
    xy = np.random.default_rng(1).uniform(0, 1300, (20000, 2))
    got = t.get_world_coordinates_batch(xy)
    rc, want = oracle_mod.world_coordinates(xy, DGT, EGT, t.elevation)
    assert rc == 0 and np.array_equal(got, want)
    assert t.get_world_coordinates(12.5, 99.25) == tuple(oracle_mod.world_coordinates([[12.5, 99.25]], DGT, EGT, t.elevation)[1][0])
    # rotated / sheared transforms take the general inverse
    t.create_geotransform("dataset", [-70.6, 2e-4, 1e-6, -33.3, -2e-6, -2e-4])
    t.create_geotransform("elevation", [-70.7, 1e-3, 1e-7, -33.2, 2e-7, -1e-3])
    got = t.get_world_coordinates_batch(xy[:3000])
    rc, want = oracle_mod.world_coordinates(xy[:3000], t.transforms["dataset"], t.transforms["elevation"], t.elevation)
    assert rc == 0 and np.array_equal(got, want)


def test_missing_elevation_and_no_elevation(gpu_pkg, oracle_mod):
    t = _table(gpu_pkg)
    with pytest.raises(gpu_pkg.ApdsError) as e:
        t.get_world_coordinates(1e7, 1e7)                       # Err(Diesel NotFound) in the reference
    assert e.value.code == -211
    bare = gpu_pkg.feature_database.ElevationTable()
    bare.create_geotransform("dataset", DGT)
    got = bare.get_world_coordinates_batch([[0, 0], [100, 100]])  # elevationdb.rs:74-77: no elevation transform -> height 0
    assert np.array_equal(got, oracle_mod.world_coordinates([[0, 0], [100, 100]], DGT)[1])
    t.create_geotransform("elevation", [0, 0, 0, 0, 0, 0])
    with pytest.raises(gpu_pkg.ApdsError) as e:
        t.get_world_coordinates(1, 1)
    assert e.value.code == -5


def test_pose_from_world_points(gpu_pkg):
    # DB keypoint pixels -> ECEF points; a camera 12 km above the terrain looks down at them; PnP recovers it. solvePnPRansac converts
    # its inputs to f32 (6.4e6 m has 0.5 m resolution there), so the points are expressed relative to their centroid first.
    hg = gpu_pkg.homographier
    t = _table(gpu_pkg)
    rng = np.random.default_rng(8)
    px = rng.uniform(100, 1300, (600, 2))
    world = t.get_world_coordinates_batch(px)
    c0 = world.mean(0)
    local = world - c0
    up = c0 / np.linalg.norm(c0)
    east = np.cross([0, 0, 1.0], up); east /= np.linalg.norm(east)
    north = np.cross(up, east)
    R = np.stack([east, -north, -up])                            # camera axes: x east, y south, z down (towards the ground)
    cam_pos = 12000.0 * up + 300.0 * east
    tvec = -R @ cam_pos
    K = np.array([[4000.0, 0, 2048], [0, 4000.0, 2048], [0, 0, 1]])
    pc = local @ R.T + tvec
    img = pc[:, :2] / pc[:, 2:3] * 4000.0 + 2048.0 + rng.normal(0, 0.3, (600, 2))
    img[:120] = rng.uniform(0, 4096, (120, 2))                   # 20 % wrong correspondences
    sol = hg.pnp_solver_ransac([hg.ImgObjCorrespondence(o, i) for o, i in zip(local, img)], hg.Cmat(K, np.float64), 500, 3.0, 0.99)
    assert sol is not None and len(sol.inliers.mat) > 450 and (sol.inliers.mat.ravel() >= 120).mean() > 0.99
    rv = sol.rvec.mat.ravel()
    th = np.linalg.norm(rv); k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Rg = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    cam_est = -Rg.T @ sol.tvec.mat.ravel()
    assert np.linalg.norm(cam_est - cam_pos) < 60.0 and np.abs(Rg - R).max() < 5e-3
