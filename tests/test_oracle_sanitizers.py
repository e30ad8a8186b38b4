"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY §5's sanitizer line; GPU sanitizers are not available on
the pool, so the CPU restatement - which shares its indexing and border rules with the kernels - is what can be checked this way).
oracle/Makefile builds liboracle_asan.so; a child process with the sanitizer runtime preloaded pushes the golden inputs and three
odd-sized tiles (odd width / height, 1-, 3- and 4-channel, a level that drops an octave) through every entry point. Clean = the child
exits 0 with no sanitizer report, and its AKAZE output equals the golden fixture."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as graft
import oracle
synth = graft.load_package().synth
oracle.set_threads(2)
g = np.load(os.path.join(%(root)r, "tests", "golden", "akaze_256.npz"))
tile = synth.make_tile(256, 256, frame_index=5)
r = oracle.akaze(tile)
assert np.array_equal(r.descriptors, g["descriptors"]) and len(r.keypoints) == len(g["keypoints"])
for (h, w, ch) in ((131, 97, 1), (173, 211, 3), (90, 333, 4)):
    img = synth.make_tile(max(h, w), max(h, w), frame_index=h)[:h, :w]
    img = np.ascontiguousarray(img[..., 0] if ch == 1 else img[..., :ch])
    oracle.akaze(img)
oracle.akaze(tile, max_points=10)
db = synth.make_descriptor_db(4000, seed=0x44420001 + 4000)
q, _ = synth.make_queries(db, 300, seed=0x51550001 + 1000)
oracle.knn_hamming(q, db, 2)
oracle.get_knn_matches(q, db, 2, 0.3)
oracle.get_bruteforce_matches(q, db)
src, dst, _, _ = synth.make_ransac_set(200, seed=0x52410001 + 200)
for method in (0, 4, 8, 16):
    oracle.find_homography(src, dst, method, 3.0)
obj, img2, K, _, _, _ = synth.make_pnp_set(200)
oracle.solve_pnp_ransac(obj, img2, K, 100, 3.0, 0.99)
rng = np.random.default_rng(3)
bands = [rng.normal(0.4, 0.3, 777).astype(np.float32) for _ in range(3)]
bands[0][::5] = np.nan
oracle.band_merger(*bands, np.array([0.0, 1.0, -0.2, 1.1, 0.05, 0.8]))
oracle.warp_perspective(rng.integers(0, 256, (37, 53, 4), dtype=np.uint8), np.array([[0.9, -0.2, 6.0], [0.25, 1.1, -3.0], [1e-3, -2e-3, 1.0]]))
print("SANITIZED RUN OK")
'''


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT})
    env = dict(os.environ, LD_PRELOAD=libasan, APDS_ORACLE_LIB=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=67", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    report = r.stdout[-2000:] + r.stderr[-4000:]
    assert r.returncode == 0 and "SANITIZED RUN OK" in r.stdout, report
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, report
