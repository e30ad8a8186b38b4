"""Seeded synthetic inputs for tests and bench (SURVEY.md §8d recipes). numpy only; no oracle, no GPU.

Everything is derived from SplitMix64 counter streams so the same seed gives the same bytes on every box.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
TILE_SEED = 0x4150445300000000
DB_SEED = 0x44420001
QUERY_SEED = 0x51550001
RANSAC_SEED = 0x52410001
L2_SEED = 0x4C320001
PNP_SEED = 0x504E0001


def splitmix64(seed, n, offset=0):
    """n outputs of SplitMix64 started at `seed` (counter form), as uint64."""
    with np.errstate(over="ignore"):
        i = np.arange(1 + offset, n + 1 + offset, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n, offset=0):
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _value_noise(seed, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    lat = uniform01(seed, gh * gw).reshape(gh, gw) * 2.0 - 1.0
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = ys - y0
    fx = xs - x0
    fy = fy * fy * (3 - 2 * fy)
    fx = fx * fx * (3 - 2 * fx)
    a = lat[y0][:, x0]
    b = lat[y0][:, x0 + 1]
    c = lat[y0 + 1][:, x0]
    d = lat[y0 + 1][:, x0 + 1]
    top = a + (b - a) * fx[None, :]
    bot = c + (d - c) * fx[None, :]
    return top + (bot - top) * fy[:, None]


def make_tile(h, w, frame_index=0, channels=4, blobs_per_mpx=2000):
    """u8 tile, grey content replicated into B,G,R (alpha 255): 128 + 48*fBm(6 octaves) + Gaussian blobs."""
    seed = TILE_SEED + frame_index
    img = np.zeros((h, w), np.float64)
    amp, cell, norm = 1.0, 128, 0.0
    for o in range(6):
        img += amp * _value_noise(seed ^ (0x1000 + o), h, w, max(cell, 2))
        norm += amp
        amp *= 0.5
        cell //= 2
    img = 128.0 + 48.0 * img / norm * 1.6
    nb = int(round(blobs_per_mpx * (w * h) / float(1024 * 1024)))
    u = uniform01(seed ^ 0xB10B, nb * 4).reshape(nb, 4)
    cx = u[:, 0] * w
    cy = u[:, 1] * h
    sg = 1.5 + u[:, 2] * 10.5
    am = -80.0 + u[:, 3] * 160.0
    for i in range(nb):
        r = int(np.ceil(4 * sg[i]))
        x0, x1 = max(int(cx[i]) - r, 0), min(int(cx[i]) + r + 1, w)
        y0, y1 = max(int(cy[i]) - r, 0), min(int(cy[i]) + r + 1, h)
        if x0 >= x1 or y0 >= y1:
            continue
        gx = np.exp(-0.5 * ((np.arange(x0, x1) - cx[i]) / sg[i]) ** 2)
        gy = np.exp(-0.5 * ((np.arange(y0, y1) - cy[i]) / sg[i]) ** 2)
        img[y0:y1, x0:x1] += am[i] * gy[:, None] * gx[None, :]
    g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if channels == 1:
        return g
    out = np.empty((h, w, channels), np.uint8)
    out[..., 0] = g
    out[..., 1] = g
    out[..., 2] = g
    if channels == 4:
        out[..., 3] = 255
    return out


def make_descriptor_db(n, seed=DB_SEED, desc_bytes=61):
    """n x desc_bytes u8, bits iid Bernoulli(1/2); bits 486,487 (top two of byte 60) forced to 0."""
    words = splitmix64(seed, n * 8)
    rows = words.view(np.uint8).reshape(n, 64)[:, :desc_bytes].copy()
    if desc_bytes == 61:
        rows[:, 60] &= 0x3F
    return rows


def make_queries(db, nq, seed=QUERY_SEED, planted=0.3, flip=0.05):
    """nq query rows: `planted` of them are noisy copies of random DB rows (each bit flipped w.p. flip)."""
    n, nb = db.shape
    q = make_descriptor_db(nq, seed ^ 0x7777, nb)
    u = uniform01(seed, nq * 2).reshape(nq, 2)
    is_planted = u[:, 0] < planted
    src = np.minimum((u[:, 1] * n).astype(np.int64), n - 1)
    pi = np.nonzero(is_planted)[0]
    if len(pi):
        noise = uniform01(seed ^ 0x5151, len(pi) * nb * 8).reshape(len(pi), nb, 8) < flip
        nbits = np.packbits(noise, axis=2, bitorder="little").reshape(len(pi), nb)
        q[pi] = db[src[pi]] ^ nbits
        if nb == 61:
            q[:, 60] &= 0x3F
    return q, np.where(is_planted, src, -1)


def make_ransac_set(n=50000, seed=RANSAC_SEED, inlier_frac=0.4, noise=0.5, extent=4096.0):
    """Point pairs under a near-identity homography: returns src, dst (n x 2 f32), H_true, inlier flag."""
    u = uniform01(seed, n * 8).reshape(n, 8)
    p = uniform01(seed ^ 0xABCD, 8)
    ang = np.deg2rad(-5 + 10 * p[0])
    sc = 0.95 + 0.1 * p[1]
    tx, ty = -40 + 80 * p[2], -40 + 80 * p[3]
    H = np.array([[sc * np.cos(ang), -sc * np.sin(ang), tx], [sc * np.sin(ang), sc * np.cos(ang), ty],
                  [(-1 + 2 * p[4]) * 1e-5, (-1 + 2 * p[5]) * 1e-5, 1.0]])
    src = u[:, 0:2] * extent
    hom = np.concatenate([src, np.ones((n, 1))], 1) @ H.T
    dst = hom[:, :2] / hom[:, 2:3]
    # Box-Muller noise
    r = np.sqrt(-2 * np.log(np.maximum(u[:, 2], 1e-300)))
    dst = dst + noise * np.stack([r * np.cos(2 * np.pi * u[:, 3]), r * np.sin(2 * np.pi * u[:, 3])], 1)
    inl = u[:, 4] < inlier_frac
    dst[~inl] = u[~inl, 5:7] * extent
    return src.astype(np.float32), dst.astype(np.float32), H, inl


def make_l2_set(n_db, n_query, dim=128, seed=L2_SEED, planted=0.3, noise=0.05):
    """SURVEY §8d synthetic L2 data: unit-norm N(0,1) rows; `planted` of the queries are a DB row + N(0, noise^2), renormalised."""
    def normal(sd, n):
        u = uniform01(sd, 2 * n).reshape(2, n)
        return np.sqrt(-2 * np.log(np.maximum(u[0], 1e-300))) * np.cos(2 * np.pi * u[1])
    db = normal(seed, n_db * dim).reshape(n_db, dim)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = normal(seed ^ 0x1111, n_query * dim).reshape(n_query, dim)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    u = uniform01(seed ^ 0x2222, n_query * 2).reshape(n_query, 2)
    src = np.minimum((u[:, 1] * n_db).astype(np.int64), n_db - 1)
    pl = u[:, 0] < planted
    pq = db[src[pl]] + noise * normal(seed ^ 0x3333, int(pl.sum()) * dim).reshape(-1, dim)
    q[pl] = pq / np.linalg.norm(pq, axis=1, keepdims=True)
    return db.astype(np.float32), q.astype(np.float32), np.where(pl, src, -1)


def make_pnp_set(n=2000, seed=PNP_SEED, inlier_frac=0.6, noise=0.5, width=4096.0, height=4096.0, focal=3500.0):
    """3D-2D correspondences for pnp_solver_ransac: object points in a 400 x 400 x 60 slab (terrain-like: wide, shallow), a camera about
    900 units above it with a small tilt, Gaussian pixel noise on the inliers and uniformly random image points for the outliers.
    Returns obj (n x 3 f64), img (n x 2 f64), K (3 x 3), rvec_true, tvec_true, inlier flag."""
    u = uniform01(seed, n * 8).reshape(n, 8)
    p = uniform01(seed ^ 0xBEEF, 8)
    K = np.array([[focal, 0.0, width / 2], [0.0, focal * 0.99, height / 2], [0.0, 0.0, 1.0]])
    rvec = np.array([-0.12 + 0.24 * p[0], -0.12 + 0.24 * p[1], -0.5 + 1.0 * p[2]])
    tvec = np.array([-20 + 40 * p[3], -20 + 40 * p[4], 850 + 100 * p[5]])
    th = np.linalg.norm(rvec)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    obj = np.stack([-200 + 400 * u[:, 0], -200 + 400 * u[:, 1], -30 + 60 * u[:, 2]], 1)
    cam = obj @ R.T + tvec
    img = cam[:, :2] / cam[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    r = np.sqrt(-2 * np.log(np.maximum(u[:, 3], 1e-300)))
    img = img + noise * np.stack([r * np.cos(2 * np.pi * u[:, 4]), r * np.sin(2 * np.pi * u[:, 4])], 1)
    inl = u[:, 5] < inlier_frac
    img[~inl] = u[~inl, 6:8] * np.array([width, height])
    return obj, img, K, rvec, tvec, inl
