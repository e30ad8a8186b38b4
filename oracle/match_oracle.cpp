// oracle/match_oracle.cpp — Hamming brute-force matching restated on the CPU. TEST INFRASTRUCTURE ONLY.
//
// Reference call sites: feature_extraction/src/lib.rs:94-114 (get_knn_matches), :116-126
// (get_bruteforce_matches), :161-180 (get_points_from_matches); homographier mod.rs:183-220 (raster_to_mat).
// Arithmetic behind them: OpenCV BFMatcher(NORM_HAMMING) -> core batchDistance / normHamming
// (features2d/src/matchers.cpp, core/src/batch_distance.cpp), not in /root/reference. PARITY UNPINNED.
#include "oracle.h"

#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
inline int hamming(const uint8_t* a, const uint8_t* b, int n) {
    int d = 0, i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        std::memcpy(&x, a + i, 8);
        std::memcpy(&y, b + i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    for (; i < n; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}
}  // namespace

extern "C" {

// batchDistance with K>0: per query keep the K smallest, strict '<' insertion => lower index wins ties.
void oracle_knn_hamming(const uint8_t* q, int nq, size_t q_stride, const uint8_t* t, int nt, size_t t_stride,
                        int desc_bytes, int k, int32_t* idx, int32_t* dist) {
    const int nth = oracle_get_threads();
    (void)nth;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int i = 0; i < nq; i++) {
        int32_t* bi = idx + (size_t)i * k;
        int32_t* bd = dist + (size_t)i * k;
        for (int j = 0; j < k; j++) {
            bi[j] = -1;
            bd[j] = INT_MAX;
        }
        const uint8_t* qi = q + (size_t)i * q_stride;
        for (int r = 0; r < nt; r++) {
            const int d = hamming(qi, t + (size_t)r * t_stride, desc_bytes);
            if (d < bd[k - 1]) {
                int j = k - 2;
                for (; j >= 0 && bd[j] > d; j--) {
                    bd[j + 1] = bd[j];
                    bi[j + 1] = bi[j];
                }
                bd[j + 1] = d;
                bi[j + 1] = r;
            }
        }
    }
}

int oracle_get_knn_matches(const uint8_t* q, int nq, size_t q_stride, const uint8_t* t, int nt, size_t t_stride,
                           int desc_bytes, int k, float filter_strength, oracle_dmatch* out) {
    if (k < 1 || desc_bytes <= 0 || nq < 0 || nt < 0) return -215;
    if (nq == 0 || nt == 0) return 0;                 // knnMatch returns no rows; the loop body never runs
    if (k < 2 || nt < 2) return -211;                 // i.get(1)? -> StsOutOfRange on the first query
    std::vector<int32_t> idx((size_t)nq * k), dist((size_t)nq * k);
    oracle_knn_hamming(q, nq, q_stride, t, nt, t_stride, desc_bytes, k, idx.data(), dist.data());
    int n = 0;
    for (int i = 0; i < nq; i++) {
        const float d0 = (float)dist[(size_t)i * k], d1 = (float)dist[(size_t)i * k + 1];
        if (d0 < d1 * filter_strength) {
            out[n].query_idx = i;
            out[n].train_idx = idx[(size_t)i * k];
            out[n].img_idx = 0;
            out[n].distance = d0;
            n++;
        }
    }
    return n;
}

// batchDistance(..., crosscheck=true): 1-NN of every TRAIN row over the queries, then per query keep the
// train row with the smallest such distance (first train row wins ties).
int oracle_get_bruteforce_matches(const uint8_t* q, int nq, size_t q_stride, const uint8_t* t, int nt, size_t t_stride,
                                  int desc_bytes, oracle_dmatch* out) {
    if (desc_bytes <= 0 || nq < 0 || nt < 0) return -215;
    if (nq == 0 || nt == 0) return 0;
    std::vector<int32_t> tidx(nt), tdist(nt);
    oracle_knn_hamming(t, nt, t_stride, q, nq, q_stride, desc_bytes, 1, tidx.data(), tdist.data());
    std::vector<int32_t> nidx(nq, -1), dist(nq, INT_MAX);
    for (int i = 0; i < nt; i++) {
        const int id = tidx[i];
        const int d = tdist[i];
        if (d < dist[id]) {
            dist[id] = d;
            nidx[id] = i;
        }
    }
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (nidx[i] < 0) continue;
        out[n].query_idx = i;
        out[n].train_idx = nidx[i];
        out[n].img_idx = 0;
        out[n].distance = (float)dist[i];
        n++;
    }
    return n;
}

int oracle_get_points_from_matches(const oracle_keypoint* kp1, int n1, const oracle_keypoint* kp2, int n2,
                                   const oracle_dmatch* m, int nm, int bug_compatible, float* pts1, float* pts2) {
    for (int i = 0; i < nm; i++) {
        const int i1 = bug_compatible ? m[i].img_idx : m[i].query_idx;   // lib.rs:169 uses img_idx
        const int i2 = m[i].train_idx;
        if (i1 < 0 || i1 >= n1 || i2 < 0 || i2 >= n2) return -211;
        pts1[2 * i] = kp1[i1].x;
        pts1[2 * i + 1] = kp1[i1].y;
        if (bug_compatible) {   // lib.rs:176-177 converts img1's keypoints twice
            pts2[2 * i] = kp1[i1].x;
            pts2[2 * i + 1] = kp1[i1].y;
        } else {
            pts2[2 * i] = kp2[i2].x;
            pts2[2 * i + 1] = kp2[i2].y;
        }
    }
    return 0;
}

void oracle_knn_l2(const float* q, int nq, const float* t, int nt, int dim, int k, int32_t* idx, float* dist) {
    const int nth = oracle_get_threads();
    (void)nth;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int i = 0; i < nq; i++) {
        int32_t* bi = idx + (size_t)i * k;
        float* bd = dist + (size_t)i * k;
        for (int j = 0; j < k; j++) {
            bi[j] = -1;
            bd[j] = INFINITY;
        }
        const float* qi = q + (size_t)i * dim;
        for (int r = 0; r < nt; r++) {
            const float* tr = t + (size_t)r * dim;
            float s = 0.f;
            for (int c = 0; c < dim; c++) {
                const float d = qi[c] - tr[c];
                s += d * d;
            }
            const float d = sqrtf(s);
            if (d < bd[k - 1]) {
                int j = k - 2;
                for (; j >= 0 && bd[j] > d; j--) {
                    bd[j + 1] = bd[j];
                    bi[j + 1] = bi[j];
                }
                bd[j + 1] = d;
                bi[j + 1] = r;
            }
        }
    }
}

int oracle_raster_to_mat(const uint8_t* rgba, size_t n_pixels, int w, int h, uint8_t* bgra) {
    if (w <= 0 || h <= 0 || n_pixels != (size_t)w * (size_t)h) return -1;
    for (size_t i = 0; i < n_pixels; i++) {
        bgra[4 * i + 0] = rgba[4 * i + 2];
        bgra[4 * i + 1] = rgba[4 * i + 1];
        bgra[4 * i + 2] = rgba[4 * i + 0];
        bgra[4 * i + 3] = rgba[4 * i + 3];
    }
    return 0;
}

}  // extern "C"
