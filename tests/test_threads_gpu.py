"""GPU: the C ABI is re-entrant (SURVEY §8b Threading): the reference calls the extraction from a rayon pool with no lock
(/root/reference/preprocessor/src/main.rs:86-89,233-243,277) and builds a fresh AKAZE / BFMatcher per call. Several host threads
run different entry points at once, each on its own per-thread stream and workspace; every result must equal the one the same
call gives when it runs alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_callers_get_their_serial_results(gpu_pkg):
    fe, hg, ge, synth = gpu_pkg.feature_extraction, gpu_pkg.homographier, gpu_pkg.geotiff_extractor, gpu_pkg.synth
    images = [synth.make_tile(384 + 64 * i, 512 - 32 * i, frame_index=20 + i, channels=(4, 3, 1, 4)[i]) for i in range(4)]
    db = synth.make_descriptor_db(60000)
    queries = [synth.make_queries(db, 700 + 100 * i, seed=0x5155_0001 + i)[0] for i in range(4)]
    sets = [synth.make_ransac_set(3000 + 500 * i, seed=0x5241_0001 + i, inlier_frac=0.5) for i in range(4)]
    bands = [synth.uniform01(77 + i, 3 * 50000).astype(np.float32).reshape(3, 50000) for i in range(4)]
    mm = ge.BandsMinMax(0.0, 1.0, 0.0, 1.0, 0.0, 1.0)

    def job(kind, i):
        if kind == 0:
            e = fe.akaze_keypoint_descriptor_extraction_def(images[i], None)
            return e.keypoints.tobytes(), e.descriptors.tobytes()
        if kind == 1:
            return fe.get_knn_matches(queries[i], db, 2, 0.8).tobytes(),
        if kind == 2:
            H, mask = hg.find_homography_mat(sets[i][0], sets[i][1], hg.HomographyMethod.RANSAC, 3.0)
            return H.mat.tobytes(), mask.mat.tobytes()
        return ge.band_merger(list(bands[i]), mm).tobytes(),

    serial = {(kind, i): job(kind, i) for kind in range(4) for i in range(4)}
    assert len(serial[(0, 0)][0]) > 28 * 50 and len(serial[(1, 0)][0]) > 16 * 50
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for step in range(16):
                    kind, i = (step + t) % 4, (step // 4 + t + rep) % 4       # threads are in different entry points at the same time
                    got = job(kind, i)
                    if got != serial[(kind, i)]:
                        errors.append((t, rep, kind, i))
        except Exception as e:                                                # pragma: no cover
            errors.append((t, repr(e)))
        finally:
            gpu_pkg.lib().apds_thread_release()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
