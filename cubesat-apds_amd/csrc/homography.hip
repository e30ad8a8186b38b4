// csrc/homography.hip — findHomography (least squares / RANSAC / LMEDS) on gfx950.
//
// Replaces opencv::calib3d::find_homography behind /root/reference/homographier/src/homographier/mod.rs:243-250.
//
// Split of the work (speculative batched RANSAC that reproduces OpenCV's sequential result exactly):
//   host   : cv::RNG sample stream + subset checks (control flow, O(iterations)), replay of the adaptive
//            termination rule over per-hypothesis inlier counts, and the O(1)-size linear algebra of the final
//            refit (9x9 Jacobi, 8x8 Levenberg-Marquardt solves);
//   device : one thread per hypothesis solves the normalised 4-point DLT in f64 (same operation order as the
//            scalar algorithm => bit-identical models); the scoring kernel evaluates hypotheses x points
//            reprojection errors in f32 (points streamed from L2, 8 hypotheses held in SGPRs per block, inlier
//            counts by wave ballot + popcount, integer atomics => deterministic); inlier mask; all per-point
//            sums of the refit (centroids, scales, 9x9 normal equations, J^T J, J^T r, residuals) as
//            deterministic two-stage reductions.
// Everything sequential in OpenCV stays sequential in meaning: hypotheses are scored speculatively in batches
// and the winner is chosen by replaying `goodCount > max(best, 3)` / RANSACUpdateNumIters in iteration order.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "config.h"
#include "kernels.h"

namespace apds {

struct P2 {
    float x, y;
};

// ---- small dense helpers shared by host and device ---------------------------------------------------------
__host__ __device__ inline double hypot_cv(double a, double b) {
    a = fabs(a);
    b = fabs(b);
    if (a > b) {
        b /= a;
        return a * sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * sqrt(1 + a * a);
    }
    return 0;
}

// Array accessors: the same algorithm text runs on host arrays and on per-thread arrays kept in LDS
// (element-major, thread-minor: consecutive lanes hit consecutive banks, no scratch memory round trips).
template <class T>
struct PlainArr {
    T* p;
    __host__ __device__ T& operator[](int i) const { return p[i]; }
};
template <class T, int STRIDE>
struct StridedArr {
    T* p;
    __host__ __device__ T& operator[](int i) const { return p[i * STRIDE]; }
};

// Symmetric eigen decomposition, pivoted Jacobi rotations (cv::eigen's JacobiImpl_ restated). A: N x N, upper
// triangle used and destroyed. W: eigenvalues descending. V rows: eigenvectors. indR/indC: N ints of scratch each.
template <int N, class MA, class MW, class MV, class MI>
__host__ __device__ inline void jacobi_eigen(MA A, MW W, MV V, MI indR, MI indC) {
    const double eps = 2.2204460492503131e-16;
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < N; j++) V[i * N + j] = 0;
        V[i * N + i] = 1;
    }
    for (int k = 0; k < N; k++) {
        W[k] = A[(N + 1) * k];
        if (k < N - 1) {
            int m = k + 1;
            double mv = fabs(A[N * k + m]);
            for (int i = k + 2; i < N; i++) {
                const double val = fabs(A[N * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            int m = 0;
            double mv = fabs(A[k]);
            for (int i = 1; i < k; i++) {
                const double val = fabs(A[N * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    const int maxIters = N * N * 30;
    for (int iters = 0; iters < maxIters; iters++) {
        int k = 0;
        double mv = fabs(A[indR[0]]);
        for (int i = 1; i < N - 1; i++) {
            const double val = fabs(A[N * i + indR[i]]);
            if (mv < val) mv = val, k = i;
        }
        int l = indR[k];
        for (int i = 1; i < N; i++) {
            const double val = fabs(A[N * indC[i] + i]);
            if (mv < val) mv = val, k = indC[i], l = i;
        }
        const double p = A[N * k + l];
        if (fabs(p) <= eps) break;
        const double y = (W[l] - W[k]) * 0.5;
        double t = fabs(y) + hypot_cv(p, y);
        double s = hypot_cv(p, t);
        const double c = t / s;
        s = p / s;
        t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[N * k + l] = 0;
        W[k] -= t;
        W[l] += t;
        double a0, b0;
#define APDS_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (int i = 0; i < k; i++) APDS_ROT(A[N * i + k], A[N * i + l]);
        for (int i = k + 1; i < l; i++) APDS_ROT(A[N * k + i], A[N * i + l]);
        for (int i = l + 1; i < N; i++) APDS_ROT(A[N * k + i], A[N * l + i]);
        for (int i = 0; i < N; i++) APDS_ROT(V[N * k + i], V[N * l + i]);
#undef APDS_ROT
        for (int j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : l;
            if (idx < N - 1) {
                int m = idx + 1;
                double mv2 = fabs(A[N * idx + m]);
                for (int i = idx + 2; i < N; i++) {
                    const double val = fabs(A[N * idx + i]);
                    if (mv2 < val) mv2 = val, m = i;
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                int m = 0;
                double mv2 = fabs(A[idx]);
                for (int i = 1; i < idx; i++) {
                    const double val = fabs(A[N * i + idx]);
                    if (mv2 < val) mv2 = val, m = i;
                }
                indC[idx] = m;
            }
        }
    }
    for (int k = 0; k < N - 1; k++) {
        int m = k;
        for (int i = k + 1; i < N; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            const double tw = W[m];
            W[m] = W[k];
            W[k] = tw;
            for (int i = 0; i < N; i++) {
                const double tv = V[N * m + i];
                V[N * m + i] = V[N * k + i];
                V[N * k + i] = tv;
            }
        }
    }
}

// smallest eigenvector of LtL -> denormalised H with H[8] == 1
template <class MA, class MW, class MV, class MI>
__host__ __device__ inline void homography_from_ltl(MA LtL /*81, full symmetric*/, MW W, MV V, MI indR, MI indC,
                                                    const double* norm /*cmx,cmy,cMx,cMy,smx,smy,sMx,sMy*/, double* Hout) {
    jacobi_eigen<9>(LtL, W, V, indR, indC);
    double H0[9];
    for (int i = 0; i < 9; i++) H0[i] = V[72 + i];
    const double invHnorm[9] = {1. / norm[4], 0, norm[0], 0, 1. / norm[5], norm[1], 0, 0, 1};
    const double Hnorm2[9] = {norm[6], 0, -norm[2] * norm[6], 0, norm[7], -norm[3] * norm[7], 0, 0, 1};
    double Ht[9], H[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += invHnorm[r * 3 + k] * H0[k * 3 + c];
            Ht[r * 3 + c] = s;
        }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Ht[r * 3 + k] * Hnorm2[k * 3 + c];
            H[r * 3 + c] = s;
        }
    const double sc = 1. / H[8];
    for (int i = 0; i < 9; i++) Hout[i] = H[i] * sc;
}

// HomographyEstimatorCallback::runKernel for a small point set held by one thread
template <class MA, class MW, class MV, class MI>
__host__ __device__ inline int run_kernel_small(const P2* M, const P2* m, int count, MA LtL, MW W, MV V, MI indR, MI indC, double* Hout) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
        cmx += m[i].x; cmy += m[i].y;
        cMx += M[i].x; cMy += M[i].y;
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[i].x - cmx); smy += fabs(m[i].y - cmy);
        sMx += fabs(M[i].x - cMx); sMy += fabs(M[i].y - cMy);
    }
    const double eps = 2.2204460492503131e-16;
    if (fabs(smx) < eps || fabs(smy) < eps || fabs(sMx) < eps || fabs(sMy) < eps) return 0;
    smx = count / smx; smy = count / smy;
    sMx = count / sMx; sMy = count / sMy;
    for (int i = 0; i < 81; i++) LtL[i] = 0;
    for (int i = 0; i < count; i++) {
        const double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        const double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    const double norm[8] = {cmx, cmy, cMx, cMy, smx, smy, sMx, sMy};
    homography_from_ltl(LtL, W, V, indR, indC, norm, Hout);
    return 1;
}

// ---- device kernels ------------------------------------------------------------------------------------------
// (Round 1 ran one thread per 4-point sample, the 9x9 system of every thread in LDS; the cooperative form below replaced it as the default
// in round 2 and the old kernel, by then launched by no test, was deleted in round 3.)
// The 4-point solve, 16 lanes per sample (four samples per wave). The matrices stay in LDS, but
// the work inside one Jacobi rotation -- the row/column updates, the eigenvector update, the four pivot-index rescans -- is
// spread over the lanes, and every lane evaluates the (cheap, uniform) pivot scan and rotation parameters itself. Each
// element goes through exactly the operations of jacobi_eigen<9> in the same order, so the result is bit-identical to one
// thread running jacobi_eigen<9> alone (the host refit, the oracle). What it cannot change is the rotation COUNT: the stopping rule
// is |pivot| <= DBL_EPSILON in absolute terms, which the rounding noise of a 9x9 system with O(10) entries rarely reaches, so
// most samples run the full 9*9*30 = 2430 rotations (as they do in OpenCV); a rotation costs ~0.2 us here against ~0.3 us.
static constexpr int COOP_LANES = 16, COOP_PER_BLOCK = 16;

struct CoopSlot {
    double A[81], V[81], W[9];
    int indR[9], indC[9];
    int pad[2];
};

__device__ __forceinline__ double coop_l_entry(int row, int j, double x, double y, double X, double Y) {
    // row 0: Lx = {X, Y, 1, 0, 0, 0, -x*X, -x*Y, -x}; row 1: Ly = {0, 0, 0, X, Y, 1, -y*X, -y*Y, -y}
    const double u = row == 0 ? x : y;
    const int jj = row == 0 ? j : j - 3;
    if (j >= 6) return j == 6 ? -u * X : (j == 7 ? -u * Y : -u);
    if (jj < 0 || jj > 2) return 0.0;
    return jj == 0 ? X : (jj == 1 ? Y : 1.0);
}

__global__ __launch_bounds__(COOP_LANES* COOP_PER_BLOCK) void hypothesis_coop_kernel(const P2* __restrict__ M, const P2* __restrict__ m,
                                                                                    const int* __restrict__ idx4, int B, double* __restrict__ models,
                                                                                    uint8_t* __restrict__ valid) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ CoopSlot s_slot[COOP_PER_BLOCK];
    const int g = threadIdx.x / COOP_LANES, li = threadIdx.x % COOP_LANES;
    const int h = blockIdx.x * COOP_PER_BLOCK + g;
    const bool live = h < B;
    volatile double* A = s_slot[g].A;
    volatile double* V = s_slot[g].V;
    volatile double* W = s_slot[g].W;
    volatile int* indR = s_slot[g].indR;
    volatile int* indC = s_slot[g].indC;
    constexpr int N = 9;
    const double eps = 2.2204460492503131e-16;
    // every lane of the group reads the four correspondences and normalises them (uniform work, identical results)
    P2 ms1[4], ms2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int id = live ? idx4[h * 4 + j] : 0;
        ms1[j] = M[id];
        ms2[j] = m[id];
    }
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        cmx += ms2[i].x; cmy += ms2[i].y;
        cMx += ms1[i].x; cMy += ms1[i].y;
    }
    cmx /= 4; cmy /= 4; cMx /= 4; cMy /= 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        smx += fabs(ms2[i].x - cmx); smy += fabs(ms2[i].y - cmy);
        sMx += fabs(ms1[i].x - cMx); sMy += fabs(ms1[i].y - cMy);
    }
    const bool ok = live && !(fabs(smx) < eps || fabs(smy) < eps || fabs(sMx) < eps || fabs(sMy) < eps);
    smx = 4 / smx; smy = 4 / smy;
    sMx = 4 / sMx; sMy = 4 / sMy;
    // normal equations: 45 upper-triangle entries over 16 lanes, each summed over the points in order
    for (int e = li; e < 45; e += COOP_LANES) {
        int j = 0, rem = e;
        while (rem >= N - j) {
            rem -= N - j;
            j++;
        }
        const int k = j + rem;
        double acc = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double x = (ms2[i].x - cmx) * smx, y = (ms2[i].y - cmy) * smy;
            const double X = (ms1[i].x - cMx) * sMx, Y = (ms1[i].y - cMy) * sMy;
            acc += coop_l_entry(0, j, x, y, X, Y) * coop_l_entry(0, k, x, y, X, Y) + coop_l_entry(1, j, x, y, X, Y) * coop_l_entry(1, k, x, y, X, Y);
        }
        A[j * N + k] = acc;
        A[k * N + j] = acc;
    }
    for (int e = li; e < 81; e += COOP_LANES) V[e] = (e % 10 == 0) ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (li < N) {   // initial eigenvalues and pivot indices, one row / column per lane
        const int k = li;
        W[k] = A[(N + 1) * k];
        if (k < N - 1) {
            int mi = k + 1;
            double mv = fabs(A[N * k + mi]);
            for (int i = k + 2; i < N; i++) {
                const double val = fabs(A[N * k + i]);
                if (mv < val) mv = val, mi = i;
            }
            indR[k] = mi;
        }
        if (k > 0) {
            int mi = 0;
            double mv = fabs(A[k]);
            for (int i = 1; i < k; i++) {
                const double val = fabs(A[N * i + k]);
                if (mv < val) mv = val, mi = i;
            }
            indC[k] = mi;
        }
    }
    __builtin_amdgcn_wave_barrier();
    bool done = !ok;
    for (int iters = 0; iters < N * N * 30; iters++) {
        if (!__any(!done)) break;   // wave-uniform exit: every group of this wave has converged
        if (!done) {
            // pivot: the sequential scan of jacobi_eigen, evaluated by every lane (broadcast LDS reads). All loads of a level
            // are issued before the first compare, so the scan costs two LDS latencies instead of sixteen.
            int ir[N - 1], ic[N];
            double vr[N - 1], vc[N];
#pragma unroll
            for (int i = 0; i < N - 1; i++) ir[i] = indR[i];
#pragma unroll
            for (int i = 1; i < N; i++) ic[i] = indC[i];
#pragma unroll
            for (int i = 0; i < N - 1; i++) vr[i] = A[N * i + ir[i]];
#pragma unroll
            for (int i = 1; i < N; i++) vc[i] = A[N * ic[i] + i];
            int k = 0;
            double mv = fabs(vr[0]);
#pragma unroll
            for (int i = 1; i < N - 1; i++) {
                const double val = fabs(vr[i]);
                if (mv < val) mv = val, k = i;
            }
            int l = 0;
#pragma unroll
            for (int i = 0; i < N - 1; i++)
                if (i == k) l = ir[i];
#pragma unroll
            for (int i = 1; i < N; i++) {
                const double val = fabs(vc[i]);
                if (mv < val) mv = val, k = ic[i], l = i;
            }
            const double p = A[N * k + l];
            if (fabs(p) <= eps) {
                done = true;
            } else {
                const double y = (W[l] - W[k]) * 0.5;
                double t = fabs(y) + hypot_cv(p, y);
                double sn = hypot_cv(p, t);
                const double c = t / sn;
                sn = p / sn;
                t = (p / t) * p;
                if (y < 0) sn = -sn, t = -t;
                __builtin_amdgcn_wave_barrier();   // all lanes have read the pivot data before anything is modified
                if (li == 0) {
                    A[N * k + l] = 0;
                    W[k] = W[k] - t;
                    W[l] = W[l] + t;
                }
                if (li < N) {
                    const int i = li;
                    double a0, b0;
                    if (i < k) {
                        a0 = A[N * i + k], b0 = A[N * i + l];
                        A[N * i + k] = a0 * c - b0 * sn;
                        A[N * i + l] = a0 * sn + b0 * c;
                    } else if (i > k && i < l) {
                        a0 = A[N * k + i], b0 = A[N * i + l];
                        A[N * k + i] = a0 * c - b0 * sn;
                        A[N * i + l] = a0 * sn + b0 * c;
                    } else if (i > l) {
                        a0 = A[N * k + i], b0 = A[N * l + i];
                        A[N * k + i] = a0 * c - b0 * sn;
                        A[N * l + i] = a0 * sn + b0 * c;
                    }
                    a0 = V[N * k + i], b0 = V[N * l + i];
                    V[N * k + i] = a0 * c - b0 * sn;
                    V[N * l + i] = a0 * sn + b0 * c;
                }
                __builtin_amdgcn_wave_barrier();
                if (li < 4) {   // lanes 0..3: indR[k], indC[k], indR[l], indC[l]; loads first, then the scan in registers
                    const int idx = li < 2 ? k : l;
                    const bool rows = (li & 1) == 0;
                    double v[N - 1];
#pragma unroll
                    for (int q = 0; q < N - 1; q++) {
                        // row scan: elements (idx, q + 1) with q + 1 > idx; column scan: elements (q, idx) with q < idx
                        const int e = rows ? N * idx + min(max(q + 1, idx + 1), N - 1) : N * min(q, max(idx - 1, 0)) + idx;
                        v[q] = A[e];
                    }
                    if (rows) {
                        if (idx < N - 1) {
                            int mi = idx + 1;
                            double mv2 = 0;
                            bool first = true;
#pragma unroll
                            for (int q = 0; q < N - 1; q++) {
                                const int i = q + 1;
                                if (i > idx) {
                                    const double val = fabs(v[q]);
                                    if (first) mv2 = val, mi = i, first = false;
                                    else if (mv2 < val) mv2 = val, mi = i;
                                }
                            }
                            indR[idx] = mi;
                        }
                    } else if (idx > 0) {
                        int mi = 0;
                        double mv2 = fabs(v[0]);
#pragma unroll
                        for (int q = 1; q < N - 1; q++) {
                            if (q < idx) {
                                const double val = fabs(v[q]);
                                if (mv2 < val) mv2 = val, mi = q;
                            }
                        }
                        indC[idx] = mi;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // eigenvalues descending: selection sort, the row swaps of V spread over the lanes
    if (ok) {
        for (int k = 0; k < N - 1; k++) {
            int mi = k;
            for (int i = k + 1; i < N; i++)
                if (W[mi] < W[i]) mi = i;
            __builtin_amdgcn_wave_barrier();
            if (k != mi) {
                if (li == 0) {
                    const double tw = W[mi];
                    W[mi] = W[k];
                    W[k] = tw;
                }
                if (li < N) {
                    const double tv = V[N * mi + li];
                    V[N * mi + li] = V[N * k + li];
                    V[N * k + li] = tv;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!live) return;
    double H[9];
    if (ok) {
        double H0[9];
        for (int i = 0; i < 9; i++) H0[i] = V[72 + i];
        const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
        const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
        double Ht[9], Hd[9];
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++) {
                double sacc = 0;
                for (int kk = 0; kk < 3; kk++) sacc += invHnorm[r * 3 + kk] * H0[kk * 3 + cc];
                Ht[r * 3 + cc] = sacc;
            }
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++) {
                double sacc = 0;
                for (int kk = 0; kk < 3; kk++) sacc += Ht[r * 3 + kk] * Hnorm2[kk * 3 + cc];
                Hd[r * 3 + cc] = sacc;
            }
        const double sc = 1. / Hd[8];
        for (int i = 0; i < 9; i++) H[i] = Hd[i] * sc;
    }
    if (li == 0) valid[h] = ok ? 1 : 0;
    if (li < 9) models[(size_t)h * 9 + li] = ok ? H[li] : 0.0;
}

__device__ __forceinline__ float reproj_err(const float (&Hf)[8], float Mx, float My, float mx, float my) {
    const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
    const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - mx;
    const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - my;
    return dx * dx + dy * dy;
}

static constexpr int HT = 8;   // hypotheses per block of the scoring kernel

// grid: x = ceil(B / HT), y = point parts. good[h] += #points with err <= t
__global__ __launch_bounds__(256) void score_kernel(const P2* __restrict__ M, const P2* __restrict__ m, int n, const double* __restrict__ models, int B,
                                                    float t, int* __restrict__ good) {
    APDS_RAISE_WAVE_PRIORITY();
    const int h0 = blockIdx.x * HT;
    float Hf[HT][8];
#pragma unroll
    for (int h = 0; h < HT; h++) {
        const int hh = min(h0 + h, B - 1);
#pragma unroll
        for (int j = 0; j < 8; j++) Hf[h][j] = (float)models[(size_t)hh * 9 + j];   // wave-uniform -> scalar registers
    }
    int cnt[HT];
#pragma unroll
    for (int h = 0; h < HT; h++) cnt[h] = 0;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = min(n, i0 + per);
    for (int base = i0; base < i1; base += 256) {
        const int i = base + threadIdx.x;
        const bool in = i < i1;
        const P2 a = M[in ? i : i0], b = m[in ? i : i0];
#pragma unroll
        for (int h = 0; h < HT; h++) {
            const float e = reproj_err(Hf[h], a.x, a.y, b.x, b.y);
            cnt[h] += __popcll(__ballot(in && e <= t));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int h = 0; h < HT; h++)
            if (h0 + h < B && cnt[h]) atomicAdd(&good[h0 + h], cnt[h]);
    }
}

__global__ void inlier_mask_kernel(const P2* __restrict__ M, const P2* __restrict__ m, int n, const double* __restrict__ model, float t,
                                   uint8_t* __restrict__ mask) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float Hf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) Hf[j] = (float)model[j];
    mask[i] = reproj_err(Hf, M[i].x, M[i].y, m[i].x, m[i].y) <= t;
}

// all errors of one hypothesis (LMEDS): err[h*n + i]
__global__ void errors_kernel(const P2* __restrict__ M, const P2* __restrict__ m, int n, const double* __restrict__ models, float* __restrict__ err) {
    APDS_RAISE_WAVE_PRIORITY();
    const int h = blockIdx.y;
    float Hf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) Hf[j] = (float)models[(size_t)h * 9 + j];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        err[(size_t)h * n + i] = reproj_err(Hf, M[i].x, M[i].y, m[i].x, m[i].y);
}

// exact k-th smallest of n non-negative floats per hypothesis (radix select on the bit pattern, 3 passes of 11/11/10 bits)
__global__ __launch_bounds__(1024) void kth_select_kernel(const float* __restrict__ err, int n, int kth, float* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ unsigned int hist[2048];
    __shared__ unsigned int s_prefix, s_k;
    const unsigned int* e = reinterpret_cast<const unsigned int*>(err + (size_t)blockIdx.x * n);
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_k = (unsigned int)kth;
    }
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    unsigned int mask_hi = 0;
    for (int pass = 0; pass < 3; pass++) {
        for (int i = threadIdx.x; i < 2048; i += 1024) hist[i] = 0;
        __syncthreads();
        const unsigned int prefix = s_prefix;
        const int sh = shifts[pass];
        const unsigned int bm = (1u << bits[pass]) - 1;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const unsigned int v = e[i];
            if ((v & mask_hi) == prefix) atomicAdd(&hist[(v >> sh) & bm], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int k = s_k, b = 0;
            for (; b <= bm; b++) {
                if (k < hist[b]) break;
                k -= hist[b];
            }
            s_k = k;
            s_prefix = prefix | (b << sh);
        }
        __syncthreads();
        mask_hi |= bm << sh;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __uint_as_float(s_prefix);
}

// ---- deterministic reductions over (masked) points: per-block partial sums of K doubles ---------------------------
static constexpr int RED_BLOCKS = 64, RED_THREADS = 256, RED_MAXK = 46;

struct RedParams {
    int kind;        // 0 centroid sums, 1 abs deviations, 2 LtL, 3 LM normal equations, 4 LM residual only
    double p[8];     // kind 1: centroids (cmx,cmy,cMx,cMy); kind 2: cmx,cmy,cMx,cMy,smx,smy,sMx,sMy; kind 3/4: h[8]
};

template <int K, class F>
__device__ __forceinline__ void block_reduce_store(double (&acc)[K], double* __restrict__ partials, F) {
    __shared__ double s_part[RED_THREADS / 64][RED_MAXK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) s_part[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double v = 0;
        for (int w = 0; w < RED_THREADS / 64; w++) v += s_part[w][threadIdx.x];
        partials[(size_t)blockIdx.x * RED_MAXK + threadIdx.x] = v;
    }
}

__device__ __forceinline__ void block_reduce_max_store(double v, double* __restrict__ partials, int slot) {
    __shared__ double s_max[RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
    if (lane == 0) s_max[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
        for (int w = 0; w < RED_THREADS / 64; w++) r = fmax(r, s_max[w]);
        partials[(size_t)blockIdx.x * RED_MAXK + slot] = r;
    }
}

__global__ __launch_bounds__(RED_THREADS) void reduce_kernel(const P2* __restrict__ M, const P2* __restrict__ m, const uint8_t* __restrict__ mask, int n,
                                                             RedParams P, double* __restrict__ partials) {
    APDS_RAISE_WAVE_PRIORITY();
    const int stride = gridDim.x * RED_THREADS;
    const int first = blockIdx.x * RED_THREADS + threadIdx.x;
    if (P.kind == 0) {
        double acc[5] = {0, 0, 0, 0, 0};
        for (int i = first; i < n; i += stride)
            if (!mask || mask[i]) {
                acc[0] += m[i].x; acc[1] += m[i].y; acc[2] += M[i].x; acc[3] += M[i].y; acc[4] += 1.0;
            }
        block_reduce_store<5>(acc, partials, 0);
    } else if (P.kind == 1) {
        double acc[4] = {0, 0, 0, 0};
        for (int i = first; i < n; i += stride)
            if (!mask || mask[i]) {
                acc[0] += fabs(m[i].x - P.p[0]); acc[1] += fabs(m[i].y - P.p[1]);
                acc[2] += fabs(M[i].x - P.p[2]); acc[3] += fabs(M[i].y - P.p[3]);
            }
        block_reduce_store<4>(acc, partials, 0);
    } else if (P.kind == 2) {
        double acc[45];
#pragma unroll
        for (int k = 0; k < 45; k++) acc[k] = 0;
        for (int i = first; i < n; i += stride)
            if (!mask || mask[i]) {
                const double x = (m[i].x - P.p[0]) * P.p[4], y = (m[i].y - P.p[1]) * P.p[5];
                const double X = (M[i].x - P.p[2]) * P.p[6], Y = (M[i].y - P.p[3]) * P.p[7];
                const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
                const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
                int q = 0;
#pragma unroll
                for (int j = 0; j < 9; j++)
#pragma unroll
                    for (int k = j; k < 9; k++) acc[q++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
            }
        block_reduce_store<45>(acc, partials, 0);
    } else {
        // HomographyRefineCallback::compute: residuals and Jacobian rows of the 8-parameter model
        double acc[45];
#pragma unroll
        for (int k = 0; k < 45; k++) acc[k] = 0;
        const double* h = P.p;
        double rinf = 0;
        for (int i = first; i < n; i += stride)
            if (!mask || mask[i]) {
                const double Mx = M[i].x, My = M[i].y;
                double ww = h[6] * Mx + h[7] * My + 1.;
                ww = fabs(ww) > 2.2204460492503131e-16 ? 1. / ww : 0;
                const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
                const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
                const double r0 = xi - m[i].x, r1 = yi - m[i].y;
                acc[44] += r0 * r0 + r1 * r1;
                rinf = fmax(rinf, fmax(fabs(r0), fabs(r1)));
                if (P.kind == 3) {
                    const double J0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
                    const double J1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 8; a++)
#pragma unroll
                        for (int b = a; b < 8; b++) acc[q++] += J0[a] * J0[b] + J1[a] * J1[b];
#pragma unroll
                    for (int a = 0; a < 8; a++) acc[36 + a] += J0[a] * r0 + J1[a] * r1;
                }
            }
        block_reduce_store<45>(acc, partials, 0);
        block_reduce_max_store(rinf, partials, 45);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
namespace {

void launch_hypotheses(const P2* M, const P2* m, const int* idx_dev, int B, double* models_dev, uint8_t* valid_dev, hipStream_t s) {
    hipLaunchKernelGGL(hypothesis_coop_kernel, dim3(ceil_div(B, COOP_PER_BLOCK)), dim3(COOP_LANES * COOP_PER_BLOCK), 0, s, M, m, idx_dev, B, models_dev, valid_dev);
}

struct RNG {
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

bool have_collinear(const P2* ptr, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = ptr[j].x - ptr[i].x, dy1 = ptr[j].y - ptr[i].y;
        for (int k = 0; k < j; k++) {
            const double dx2 = ptr[k].x - ptr[i].x, dy2 = ptr[k].y - ptr[i].y;
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}

double det3(const double* a) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

bool check_subset(const P2* s, const P2* d, int count) {
    if (have_collinear(s, count) || have_collinear(d, count)) return false;
    if (count == 4) {
        static const int tt[][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int* t = tt[i];
            const double A[9] = {s[t[0]].x, s[t[0]].y, 1., s[t[1]].x, s[t[1]].y, 1., s[t[2]].x, s[t[2]].y, 1.};
            const double B[9] = {d[t[0]].x, d[t[0]].y, 1., d[t[1]].x, d[t[1]].y, 1., d[t[2]].x, d[t[2]].y, 1.};
            negative += det3(A) * det3(B) < 0;
        }
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}

bool get_subset(const P2* m1, const P2* m2, int count, int* idx, RNG& rng, int maxAttempts) {
    P2 ms1[4], ms2[4];
    for (int iters = 0; iters < maxAttempts; ++iters) {
        int i;
        for (i = 0; i < 4; ++i) {
            int idx_i;
            for (idx_i = rng.uniform(0, count); std::find(idx, idx + i, idx_i) != idx + i; idx_i = rng.uniform(0, count)) {
            }
            idx[i] = idx_i;
            ms1[i] = m1[idx_i];
            ms2[i] = m2[idx_i];
        }
        if (check_subset(ms1, ms2, i)) return true;
    }
    return false;
}

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

// sum the per-block partials in block order (deterministic)
void run_reduce(const P2* M, const P2* m, const uint8_t* mask, int n, const RedParams& P, int K, double* out, double* partials_dev,
                double* partials_host, hipStream_t s) {
    hipLaunchKernelGGL(reduce_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, M, m, mask, n, P, partials_dev);
    HIP_CHECK(hipMemcpyAsync(partials_host, partials_dev, sizeof(double) * RED_BLOCKS * RED_MAXK, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < K; k++) {
        double v = 0;
        for (int b = 0; b < RED_BLOCKS; b++) v += partials_host[b * RED_MAXK + k];
        out[k] = v;
    }
    if (P.kind >= 3) {   // slot 45 carries max |r| instead of a sum
        double v = 0;
        for (int b = 0; b < RED_BLOCKS; b++) v = std::max(v, partials_host[b * RED_MAXK + 45]);
        out[45] = v;
    }
}

// x = sum_i (v_i . b / w_i) v_i over eigenpairs with |w_i| > 2 eps sum(w)  (cv::solve / cv::invert, DECOMP_EIG)
void eig_solve8(const double* Asym, const double* b, int nb, double* x) {
    double A[64], W[8], V[64];
    int indR[8], indC[8];
    std::memcpy(A, Asym, sizeof(A));
    jacobi_eigen<8>(PlainArr<double>{A}, PlainArr<double>{W}, PlainArr<double>{V}, PlainArr<int>{indR}, PlainArr<int>{indC});
    double threshold = 0;
    for (int i = 0; i < 8; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 8 * nb; i++) x[i] = 0;
    for (int i = 0; i < 8; i++) {
        double wi = W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int c = 0; c < nb; c++) {
            double sdot = 0;
            for (int j = 0; j < 8; j++) sdot += V[i * 8 + j] * b[j * nb + c];
            sdot *= wi;
            for (int j = 0; j < 8; j++) x[j * nb + c] += sdot * V[i * 8 + j];
        }
    }
}

// Sums over the selected points. Large selections: two-stage parallel f64 reductions on the device (reduce_kernel). Small
// ones (<= HOST_REFIT_MAX points, where the refit is poorly conditioned and every rounding shows): a plain loop on the host
// over the compressed points in index order, with the associations of the sequential algorithm (J^T J and |r|^2 accumulate
// row by row, i.e. the x and y residual of a point in two steps), which makes the result bit-identical to the CPU restatement.
constexpr int HOST_REFIT_MAX = 256;

struct Refit {
    const P2 *M, *m;
    const uint8_t* mask;
    int n;
    double *pd, *ph;
    hipStream_t s;
    bool host = false;
    std::vector<P2> selM, selm;   // host: the selected (inlier) pairs, compressed, index order

    void sums(const RedParams& P, int K, double* r) {
        if (!host) {
            run_reduce(M, m, mask, n, P, K, r, pd, ph, s);
            return;
        }
        for (int k = 0; k < 46; k++) r[k] = 0;
        double rinf = 0;
        const double* h = P.p;
        for (size_t i = 0; i < selM.size(); i++) {
            const P2 Mi = selM[i], mi = selm[i];
            if (P.kind == 0) {
                r[0] += mi.x; r[1] += mi.y; r[2] += Mi.x; r[3] += Mi.y; r[4] += 1.0;
            } else if (P.kind == 1) {
                r[0] += std::fabs(mi.x - P.p[0]); r[1] += std::fabs(mi.y - P.p[1]);
                r[2] += std::fabs(Mi.x - P.p[2]); r[3] += std::fabs(Mi.y - P.p[3]);
            } else if (P.kind == 2) {
                const double x = (mi.x - P.p[0]) * P.p[4], y = (mi.y - P.p[1]) * P.p[5];
                const double X = (Mi.x - P.p[2]) * P.p[6], Y = (Mi.y - P.p[3]) * P.p[7];
                const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
                const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
                int q = 0;
                for (int j = 0; j < 9; j++)
                    for (int k = j; k < 9; k++) r[q++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
            } else {
                const double Mx = Mi.x, My = Mi.y;
                double ww = h[6] * Mx + h[7] * My + 1.;
                ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
                const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
                const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
                const double r0 = xi - mi.x, r1 = yi - mi.y;
                r[44] += r0 * r0;
                r[44] += r1 * r1;
                rinf = std::max(rinf, std::max(std::fabs(r0), std::fabs(r1)));
                if (P.kind == 3) {
                    const double J0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
                    const double J1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
                    int q = 0;
                    for (int a = 0; a < 8; a++)
                        for (int b = a; b < 8; b++) {
                            r[q] += J0[a] * J0[b];
                            r[q] += J1[a] * J1[b];
                            q++;
                        }
                    for (int a = 0; a < 8; a++) {
                        r[36 + a] += J0[a] * r0;
                        r[36 + a] += J1[a] * r1;
                    }
                }
            }
        }
        r[45] = rinf;
    }

    // runKernel over the masked points: returns 0 when degenerate
    int run_kernel(double* H) {
        RedParams P{};
        double r[46];
        P.kind = 0;
        sums(P, 5, r);
        const double count = r[4];
        if (count < 1) return 0;
        const double cmx = r[0] / count, cmy = r[1] / count, cMx = r[2] / count, cMy = r[3] / count;
        P.kind = 1;
        P.p[0] = cmx; P.p[1] = cmy; P.p[2] = cMx; P.p[3] = cMy;
        sums(P, 4, r);
        double smx = r[0], smy = r[1], sMx = r[2], sMy = r[3];
        if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON) return 0;
        smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
        P.kind = 2;
        P.p[4] = smx; P.p[5] = smy; P.p[6] = sMx; P.p[7] = sMy;
        sums(P, 45, r);
        double LtL[81];
        int q = 0;
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] = LtL[k * 9 + j] = r[q++];
        const double norm[8] = {cmx, cmy, cMx, cMy, smx, smy, sMx, sMy};
        double W9[9], V81[81];
        int indR[9], indC[9];
        homography_from_ltl(PlainArr<double>{LtL}, PlainArr<double>{W9}, PlainArr<double>{V81}, PlainArr<int>{indR}, PlainArr<int>{indC}, norm, H);
        return 1;
    }

    double rinf_last = 0;   // |r|_inf of the last normal_eq call

    // kind 3: A = J^T J, v = J^T r, S = |r|^2 ; kind 4: S only
    void normal_eq(const double* h, double* A, double* v, double& S, bool need_J) {
        RedParams P{};
        P.kind = need_J ? 3 : 4;
        for (int i = 0; i < 8; i++) P.p[i] = h[i];
        double r[46];
        sums(P, 45, r);
        S = r[44];
        rinf_last = r[45];
        if (need_J) {
            int q = 0;
            for (int a = 0; a < 8; a++)
                for (int b = a; b < 8; b++) A[a * 8 + b] = A[b * 8 + a] = r[q++];
            for (int a = 0; a < 8; a++) v[a] = r[36 + a];
        }
    }

    // LMSolver::run (levmarq.cpp), 8 parameters, maxIters 10, eps FLT_EPSILON
    void lm_refine(double* H, int maxIters) {
        const int lx = 8;
        double x[8], xd[8], A[64], Ap[64], v[8], d[8], temp_d[8], D[8];
        for (int i = 0; i < 8; i++) x[i] = H[i];
        double S;
        normal_eq(x, A, v, S, true);
        double rinf = rinf_last;
        for (int i = 0; i < lx; i++) D[i] = A[i * 8 + i];
        const double Rlo = 0.25, Rhi = 0.75;
        double lambda = 1, lc = 0.75;
        int iter = 0;
        for (;;) {
            std::memcpy(Ap, A, sizeof(A));
            for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
            eig_solve8(Ap, v, 1, d);
            for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
            double Sd, dummyA[1], dummyv[1];
            normal_eq(xd, dummyA, dummyv, Sd, false);
            for (int a = 0; a < 8; a++) {
                double sacc = 0;
                for (int b = 0; b < 8; b++) sacc += A[a * 8 + b] * d[b];
                temp_d[a] = -sacc + 2 * v[a];
            }
            double dS = 0;
            for (int a = 0; a < 8; a++) dS += d[a] * temp_d[a];
            const double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
            if (R > Rhi) {
                lambda *= 0.5;
                if (lambda < lc) lambda = 0;
            } else if (R < Rlo) {
                double t = 0;
                for (int a = 0; a < 8; a++) t += d[a] * v[a];
                double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
                nu = std::min(std::max(nu, 2.), 10.);
                if (lambda == 0) {
                    double I8[64] = {0};
                    for (int i = 0; i < 8; i++) I8[i * 8 + i] = 1;
                    eig_solve8(A, I8, 8, Ap);
                    double maxval = DBL_EPSILON;
                    for (int i = 0; i < lx; i++) maxval = std::max(maxval, std::fabs(Ap[i * 8 + i]));
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            if (Sd < S) {
                std::memcpy(x, xd, sizeof(x));
                normal_eq(x, A, v, S, true);
                rinf = rinf_last;
            }
            iter++;
            double dinf = 0;
            for (int i = 0; i < 8; i++) dinf = std::max(dinf, std::fabs(d[i]));
            const bool proceed = iter < maxIters && dinf >= FLT_EPSILON && rinf >= FLT_EPSILON;
            if (!proceed) break;
        }
        for (int i = 0; i < 8; i++) H[i] = x[i];
    }
};

}  // namespace

// returns 1 (model found; H_host filled, mask_dev filled if non-null) or 0 (none)
int find_homography_device(const float* src, const float* dst, int n, int method, double thr, int max_iters, double confidence, double* H_host,
                           uint8_t* mask_dev, hipStream_t s) {
    APDS_REQUIRE(src && dst && H_host, APDS_ERR_BAD_ARG, "null argument");
    APDS_REQUIRE(n >= 4, APDS_ERR_ASSERT, "at least 4 point pairs are required");
    APDS_REQUIRE(method == 0 || method == APDS_HOMOGRAPHY_LMEDS || method == APDS_HOMOGRAPHY_RANSAC || method == APDS_HOMOGRAPHY_RHO, APDS_ERR_BAD_ARG,
                 "unknown homography method");
    APDS_REQUIRE(confidence > 0 && confidence < 1, APDS_ERR_ASSERT, "confidence must be in (0,1)");
    if (method == APDS_HOMOGRAPHY_RHO && n > 4)   // its own estimator and refinement (homography_rho.hip); n == 4 is the plain solve, as in OpenCV
        return find_homography_rho_device(src, dst, n, thr, max_iters, confidence, H_host, mask_dev, s);
    if (thr <= 0) thr = 3;
    ThreadCtx& c = ctx();
    const P2* M = reinterpret_cast<const P2*>(src);
    const P2* m = reinterpret_cast<const P2*>(dst);
    uint8_t* mask = mask_dev ? mask_dev : c.alloc_n<uint8_t>(n);
    double* partials_dev = c.alloc_n<double>((size_t)RED_BLOCKS * RED_MAXK);
    std::vector<double> partials_host((size_t)RED_BLOCKS * RED_MAXK);
    Refit refit{M, m, nullptr, n, partials_dev, partials_host.data(), s};
    bool result = false;
    // host copy of the points: drives the cv::RNG sample stream (subset checks need the coordinates) and small refits
    std::vector<P2> hM, hm;
    auto fetch_points = [&]() {
        if (!hM.empty()) return;
        hM.resize(n);
        hm.resize(n);
        HIP_CHECK(hipMemcpyAsync(hM.data(), M, (size_t)n * sizeof(P2), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(hm.data(), m, (size_t)n * sizeof(P2), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    };
    int selected = n;   // points the final refit runs over (inliers), when known

    if (method == 0 || n == 4) {
        HIP_CHECK(hipMemsetAsync(mask, 1, n, s));
        if (n <= HOST_REFIT_MAX) {
            fetch_points();
            refit.host = true;
            refit.selM = hM;
            refit.selm = hm;
        }
        result = refit.run_kernel(H_host) > 0;
    } else {
        fetch_points();
        RNG rng((uint64_t)-1);
        const int batch_env = config().ransac_batch;
        double best_model[9] = {0};

        if (method == APDS_HOMOGRAPHY_RANSAC) {
            int niters = std::max(max_iters, 1), maxGood = 0, iter = 0;
            bool stop = false, any = false;
            const float t = (float)(thr * thr);
            // First batch: batch_env samples (with a fair inlier ratio the budget collapses to a few dozen iterations after the
            // first good model). Later batches cover the whole remaining budget, up to 4096 at once: a batch costs about the
            // same wall time whatever its size (one thread per sample, latency-bound), so low-inlier inputs that keep the
            // full budget finish in two round trips instead of max_iters / 512.
            const int first_batch = std::max(8, std::min(batch_env, std::max(max_iters, 8)));
            const int max_batch = std::max(first_batch, std::min(4096, std::max(max_iters, 8)));
            int* idx_dev = c.alloc_n<int>((size_t)max_batch * 4);
            double* models_dev = c.alloc_n<double>((size_t)max_batch * 9);
            uint8_t* valid_dev = c.alloc_n<uint8_t>(max_batch);
            int* good_dev = c.alloc_n<int>(max_batch);
            std::vector<int> idx((size_t)max_batch * 4), good(max_batch);
            std::vector<uint8_t> valid(max_batch);
            std::vector<double> models((size_t)max_batch * 9);
            while (!stop && iter < niters) {
                // speculate: samples for the next `batch` iterations (the RNG stream does not depend on the scores)
                const int batch = iter == 0 ? first_batch : max_batch;
                int B = 0;
                bool subset_failed = false;
                for (; B < batch && iter + B < niters; B++)
                    if (!get_subset(hM.data(), hm.data(), n, &idx[(size_t)B * 4], rng, 10000)) {
                        subset_failed = true;
                        break;
                    }
                if (B > 0) {
                    HIP_CHECK(hipMemcpyAsync(idx_dev, idx.data(), (size_t)B * 4 * sizeof(int), hipMemcpyHostToDevice, s));
                    HIP_CHECK(hipMemsetAsync(good_dev, 0, (size_t)B * sizeof(int), s));
                    launch_hypotheses(M, m, idx_dev, B, models_dev, valid_dev, s);
                    {
                        KernelTimer timer("ransac_score", s);
                        const int parts = std::max(1, std::min(64, ceil_div(256 * 8, ceil_div(B, HT))));
                        hipLaunchKernelGGL(score_kernel, dim3(ceil_div(B, HT), parts), dim3(256), 0, s, M, m, n, (const double*)models_dev, B, t, good_dev);
                    }
                    HIP_CHECK(hipMemcpyAsync(good.data(), good_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipMemcpyAsync(valid.data(), valid_dev, (size_t)B, hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipMemcpyAsync(models.data(), models_dev, (size_t)B * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipStreamSynchronize(s));
                    // replay OpenCV's sequential loop over the speculated iterations
                    for (int b = 0; b < B && iter < niters; b++, iter++) {
                        if (!valid[b]) continue;
                        if (good[b] > std::max(maxGood, 3)) {
                            std::memcpy(best_model, &models[(size_t)b * 9], sizeof(best_model));
                            maxGood = good[b];
                            any = true;
                            niters = ransac_update_num_iters(confidence, (double)(n - good[b]) / n, 4, niters);
                        }
                    }
                }
                if (subset_failed) stop = true;   // iter == 0 -> no model; otherwise keep the best so far
            }
            if (any) {
                HIP_CHECK(hipMemcpyAsync(models_dev, best_model, sizeof(best_model), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(inlier_mask_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, M, m, n, (const double*)models_dev, t, mask);
                std::memcpy(H_host, best_model, sizeof(best_model));
                result = true;
                selected = maxGood;
            }
        } else {   // LMEDS
            const double outlierRatio = 0.45;
            int niters = ransac_update_num_iters(confidence, outlierRatio, 4, max_iters);
            niters = std::max(niters, 3);
            std::vector<int> idx((size_t)niters * 4);
            int B = 0;
            for (; B < niters; B++)
                if (!get_subset(hM.data(), hm.data(), n, &idx[(size_t)B * 4], rng, 1000)) break;
            if (B > 0) {
                int* idx_dev = c.alloc_n<int>((size_t)B * 4);
                double* models_dev = c.alloc_n<double>((size_t)B * 9);
                uint8_t* valid_dev = c.alloc_n<uint8_t>(B);
                float* err_dev = c.alloc_n<float>((size_t)B * n);
                float* med_dev = c.alloc_n<float>(B);
                HIP_CHECK(hipMemcpyAsync(idx_dev, idx.data(), (size_t)B * 4 * sizeof(int), hipMemcpyHostToDevice, s));
                launch_hypotheses(M, m, idx_dev, B, models_dev, valid_dev, s);
                hipLaunchKernelGGL(errors_kernel, dim3(std::min(64, ceil_div(n, 256)), B), dim3(256), 0, s, M, m, n, (const double*)models_dev, err_dev);
                hipLaunchKernelGGL(kth_select_kernel, dim3(B), dim3(1024), 0, s, (const float*)err_dev, n, n / 2, med_dev);
                std::vector<float> med(B);
                std::vector<uint8_t> valid(B);
                std::vector<double> models((size_t)B * 9);
                HIP_CHECK(hipMemcpyAsync(med.data(), med_dev, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(valid.data(), valid_dev, (size_t)B, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(models.data(), models_dev, (size_t)B * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                double minMedian = DBL_MAX;
                for (int b = 0; b < B; b++) {
                    if (!valid[b]) continue;
                    const double median = med[b];
                    if (median < minMedian) {
                        minMedian = median;
                        std::memcpy(best_model, &models[(size_t)b * 9], sizeof(best_model));
                    }
                }
                if (minMedian < DBL_MAX) {
                    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 4)) * std::sqrt(minMedian);
                    sigma = std::max(sigma, 0.001);
                    const float t = (float)(sigma * sigma);
                    HIP_CHECK(hipMemcpyAsync(models_dev, best_model, sizeof(best_model), hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(inlier_mask_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, M, m, n, (const double*)models_dev, t, mask);
                    RedParams P{};
                    P.kind = 0;
                    double r[8];
                    run_reduce(M, m, mask, n, P, 5, r, partials_dev, partials_host.data(), s);
                    std::memcpy(H_host, best_model, sizeof(best_model));
                    result = r[4] >= 4;
                    selected = (int)r[4];
                }
            }
        }
    }
    HIP_CHECK(hipGetLastError());

    if (result && n > 4) {
        refit.mask = mask;
        if (!refit.host && selected <= HOST_REFIT_MAX) {   // few inliers: refit on the host in index order
            fetch_points();
            std::vector<uint8_t> hmask(n);
            HIP_CHECK(hipMemcpyAsync(hmask.data(), mask, n, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            refit.host = true;
            refit.selM.clear();
            refit.selm.clear();
            for (int i = 0; i < n; i++)
                if (hmask[i]) {
                    refit.selM.push_back(hM[i]);
                    refit.selm.push_back(hm[i]);
                }
        }
        if (method == APDS_HOMOGRAPHY_RANSAC || method == APDS_HOMOGRAPHY_LMEDS) refit.run_kernel(H_host);
        refit.lm_refine(H_host, 10);
    }
    if (!result) {
        HIP_CHECK(hipMemsetAsync(mask, 0, n, s));
        for (int i = 0; i < 9; i++) H_host[i] = 0;
    }
    HIP_CHECK(hipStreamSynchronize(s));
    return result ? 1 : 0;
}

}  // namespace apds
