// oracle/rho_oracle.cpp — HomographyMethod::RHO restated on the CPU. TEST INFRASTRUCTURE ONLY.
//
// Reference call site: homographier/src/homographier/mod.rs:25-31 (enum value RHO = 16) passed straight to
// opencv::calib3d::find_homography at mod.rs:241-250. The arithmetic is OpenCV's calib3d/src/rho.cpp (Bazargani, Bilaniuk,
// Laganiere: "A fast and robust homography scheme for real-time planar target detection"): PROSAC sampling, SPRT verification,
// the non-randomness bound on the iteration count, and a final Levenberg-Marquardt refinement, all in binary32. That file is not
// in /root/reference and OpenCV is not installed: restated from the published algorithm and from memory of rho.cpp's structure.
// PARITY UNPINNED: "equal to the oracle" means equal to THIS restatement, not to cv::findHomography(RHO). Controller points restated
// from rho.cpp as recalled (round 3, after review): the loop runs `i < maxI || i < 100` (a floor of 100 iterations); verify() is
// evaluateModelSPRT -> updateSPRT -> if (curr.numInl > best.numInl) {saveBestModel (array SWAP); updateBounds; nStarOptimize}, i.e.
// the best-model test ignores the SPRT verdict and a rejected model's inlier array keeps stale flags behind its last tested point. Known deviation: the 4-point solve is a generic Gauss-Jordan elimination with partial pivoting on the 8 x 9
// system (rho.cpp's hFuncRefC eliminates a hand-reduced form of it): the same homography up to binary32 rounding, which only
// decides inlier flags of points that lie on the threshold.
//
// This is a plain sequential program: one hypothesis at a time, every decision taken in the order rho.cpp takes it. The product
// (csrc/homography_rho.hip) reaches the same result by scoring speculated batches on the GPU and replaying this loop over them.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

constexpr int SMPL = 4;
constexpr double SPRT_T_M = 25, SPRT_M_S = 1, SPRT_EPSILON = 0.1, SPRT_DELTA = 0.01;
constexpr double MIN_DELTA_CHNG = 0.1, CHI_SQ = 1.645;
constexpr float LM_GAIN_LO = 0.25f, LM_GAIN_HI = 0.75f;
constexpr int MAX_LM_ITERS = 100;

struct Rho {
    const float* src;
    const float* dst;
    int N;
    float maxD;
    unsigned maxI, rConvg;
    double cfd, beta;
    unsigned minInl;

    // xorshift128+
    uint64_t s0, s1;
    void seed(uint64_t v) {
        s0 = v;
        s1 = ~v;
        for (int i = 0; i < 20; i++) rnd();
    }
    double rnd() {
        uint64_t x = s0;
        const uint64_t y = s1;
        x ^= x << 23;
        x ^= x >> 17;
        x ^= y ^ (y >> 26);
        s0 = y;
        s1 = x;
        return (double)(x + y) * 5.421010862427522e-20;   // 2^-64
    }

    // PROSAC state
    unsigned it = 0, phNum = SMPL, phEndI = 1, phMax = 0, phNumInl = 0, smpl[4] = {0, 0, 0, 0};
    double phEndFpI = 0;
    // SPRT state
    double eps = SPRT_EPSILON, delta = SPRT_DELTA, A = 0, lamReject = 0, lamAccept = 0;
    bool good = false;
    unsigned nTested = 0;
    // models
    float curH[9], bestH[9];
    std::vector<uint8_t> curInl, bestInl;
    unsigned curNum = 0, bestNum = 0;
    std::vector<unsigned> nrTbl;

    void design_sprt() {
        const double C = (1 - delta) * std::log((1 - delta) / (1 - eps)) + delta * std::log(delta / eps);
        const double K = SPRT_T_M * C / SPRT_M_S + 1;
        double An = K, prev;
        unsigned i = 0;
        do {
            prev = An;
            An = K + std::log(An);
        } while ((An - prev > 1.5e-8) && (++i < 10));
        A = An;
        lamReject = (1.0 - delta) / (1.0 - eps);
        lamAccept = delta / eps;
    }

    static unsigned iter_bound(double confidence, double inlierRate, unsigned sampleSize, unsigned maxIterBound) {
        confidence = confidence <= 0 ? 0 : (confidence >= 1 ? 1 : confidence);
        inlierRate = inlierRate <= 0 ? 0 : (inlierRate >= 1 ? 1 : inlierRate);
        const double pOut = 1. - std::pow(inlierRate, (double)sampleSize);
        unsigned r;
        if (pOut >= 1.) r = maxIterBound;
        else if (pOut <= 0.) r = 1;
        else {
            const double v = std::ceil(std::log(1. - confidence) / std::log(pOut));
            r = v >= (double)maxIterBound ? maxIterBound : (unsigned)v;
        }
        return r <= maxIterBound ? r : maxIterBound;
    }

    void rnd_sample(unsigned k, unsigned* out, unsigned setSize) {
        if (k * 2 > setSize) {   // selection sampling (TAOCP 3.4.2 S)
            unsigned j = 0;
            for (unsigned i = 0; i < setSize && j < k; i++) {
                const double U = rnd(), a = k - j, b = setSize - i;
                if (a > b * U) out[j++] = i;
            }
        } else {
            for (unsigned i = 0; i < k; i++) {
                bool dup;
                do {
                    out[i] = (unsigned)(setSize * rnd());
                    dup = false;
                    for (unsigned j = 0; j < i; j++)
                        if (out[i] == out[j]) {
                            dup = true;
                            break;
                        }
                } while (dup);
            }
        }
    }

    bool sample_degenerate(float* pk) const {   // pk: 4 source points then 4 destination points (x, y)
        for (int k = 0; k < 4; k++) {
            pk[2 * k] = src[2 * smpl[k]];
            pk[2 * k + 1] = src[2 * smpl[k] + 1];
            pk[8 + 2 * k] = dst[2 * smpl[k]];
            pk[8 + 2 * k + 1] = dst[2 * smpl[k] + 1];
        }
        for (int a = 0; a < 4; a++)
            for (int b = a + 1; b < 4; b++)
                if (pk[2 * a] == pk[2 * b] || pk[2 * a + 1] == pk[2 * b + 1]) return true;   // source points share an x or a y
        auto side = [&](int p, int q, int r, int base) {   // (p x q) . r in homogeneous coordinates
            const float* P = pk + base;
            const float c0 = P[2 * p + 1] - P[2 * q + 1];
            const float c1 = P[2 * q] - P[2 * p];
            const float c2 = P[2 * p] * P[2 * q + 1] - P[2 * p + 1] * P[2 * q];
            return c0 * P[2 * r] + c1 * P[2 * r + 1] + c2;
        };
        const int tests[4][3] = {{0, 1, 2}, {0, 1, 3}, {2, 3, 0}, {2, 3, 1}};
        for (auto& t : tests) {
            const float a = side(t[0], t[1], t[2], 0), b = side(t[0], t[1], t[2], 8);
            if ((((int)a) ^ ((int)b)) < 0) return true;   // the two quadrilaterals are oriented differently
        }
        return false;
    }

    // 4-point homography with H[8] = 1: Gauss-Jordan with partial pivoting on the 8 x 9 system, binary32, one IEEE operation per
    // source operation. false: singular.
    static bool solve4(const float* pk, float* H) {
        float M[8][9];
        for (int k = 0; k < 4; k++) {
            const float x = pk[2 * k], y = pk[2 * k + 1], X = pk[8 + 2 * k], Y = pk[8 + 2 * k + 1];
            float* r0 = M[2 * k];
            float* r1 = M[2 * k + 1];
            r0[0] = x, r0[1] = y, r0[2] = 1, r0[3] = 0, r0[4] = 0, r0[5] = 0, r0[6] = -(x * X), r0[7] = -(y * X), r0[8] = X;
            r1[0] = 0, r1[1] = 0, r1[2] = 0, r1[3] = x, r1[4] = y, r1[5] = 1, r1[6] = -(x * Y), r1[7] = -(y * Y), r1[8] = Y;
        }
        for (int c = 0; c < 8; c++) {
            int p = c;
            float best = std::fabs(M[c][c]);
            for (int r = c + 1; r < 8; r++)
                if (std::fabs(M[r][c]) > best) {
                    best = std::fabs(M[r][c]);
                    p = r;
                }
            if (!(best > 0.0f)) return false;
            if (p != c)
                for (int j = 0; j < 9; j++) {
                    const float t = M[c][j];
                    M[c][j] = M[p][j];
                    M[p][j] = t;
                }
            for (int r = 0; r < 8; r++) {
                if (r == c) continue;
                const float f = M[r][c] / M[c][c];
                for (int j = c; j < 9; j++) M[r][j] = M[r][j] - f * M[c][j];
            }
        }
        for (int i = 0; i < 8; i++) H[i] = M[i][8] / M[i][i];
        H[8] = 1.0f;
        for (int i = 0; i < 8; i++)
            if (!std::isfinite(H[i])) return false;
        return true;
    }

    static bool inlier(const float* H, float x, float y, float X, float Y, float maxDsq) {
        float rx = H[0] * x + H[1] * y + H[2];
        float ry = H[3] * x + H[4] * y + H[5];
        const float rz = H[6] * x + H[7] * y + 1.0f;
        rx /= rz;
        ry /= rz;
        rx -= X;
        ry -= Y;
        rx *= rx;
        ry *= ry;
        return rx + ry <= maxDsq;
    }

    void evaluate_sprt() {
        double lambda = 1.0;
        const float maxDsq = maxD * maxD;
        curNum = 0;
        good = true;
        int i = 0;
        for (; i < N && good; i++) {
            const bool in = inlier(curH, src[2 * i], src[2 * i + 1], dst[2 * i], dst[2 * i + 1], maxDsq);
            curNum += in;
            curInl[i] = in;
            lambda *= in ? lamAccept : lamReject;
            good = lambda <= A;
        }
        nTested = i;
    }

    void update_sprt() {
        if (good) {
            if (curNum > bestNum) {
                eps = (double)curNum / N;
                design_sprt();
            }
        } else {
            const double nd = (double)curNum / nTested;
            if (nd > 0) {
                const double rel = std::fabs(delta - nd) / delta;
                if (rel > MIN_DELTA_CHNG) {
                    delta = nd;
                    design_sprt();
                }
            }
        }
    }

    void nstar_optimize() {
        const unsigned min_len = 10 * 2;
        unsigned best_n = N, test_n = N, bestInlN = bestNum, testInl = bestNum;
        for (; test_n > min_len && testInl; test_n--) {
            if ((uint64_t)testInl * best_n > (uint64_t)bestInlN * test_n) {
                if (testInl < nrTbl[test_n]) break;
                best_n = test_n;
                bestInlN = testInl;
            }
            testInl -= bestInl[test_n - 1] ? 1 : 0;
        }
        if ((uint64_t)bestInlN * phMax > (uint64_t)phNumInl * best_n) {
            phMax = best_n;
            phNumInl = bestInlN;
            maxI = iter_bound(cfd, (double)phNumInl / phMax, SMPL, maxI);
        }
    }

    // sum of squared reprojection errors over the inliers and, if asked, the normal equations of the 8-parameter problem (binary32)
    void jacobian_errors(const float* H, float* JtJ, float* Jte, float* Sp) const {
        float S = 0.0f;
        if (JtJ) std::memset(JtJ, 0, 64 * sizeof(float));
        if (Jte) std::memset(Jte, 0, 8 * sizeof(float));
        for (int i = 0; i < N; i++) {
            if (!bestInl[i]) continue;
            const float x = src[2 * i], y = src[2 * i + 1], X = dst[2 * i], Y = dst[2 * i + 1];
            const float W = H[6] * x + H[7] * y + 1.0f;
            float iW = std::fabs(W) > FLT_EPSILON ? 1.0f / W : 0.0f;
            const float reprojX = (H[0] * x + H[1] * y + H[2]) * iW;
            const float reprojY = (H[3] * x + H[4] * y + H[5]) * iW;
            const float eX = reprojX - X, eY = reprojY - Y;
            const float e = eX * eX + eY * eY;
            S += e;
            if (JtJ || Jte) {
                const float dxh11 = x * iW, dxh12 = y * iW, dxh13 = iW, dxh31 = -reprojX * x * iW, dxh32 = -reprojX * y * iW;
                const float dyh21 = x * iW, dyh22 = y * iW, dyh23 = iW, dyh31 = -reprojY * x * iW, dyh32 = -reprojY * y * iW;
                const float jx[8] = {dxh11, dxh12, dxh13, 0, 0, 0, dxh31, dxh32};
                const float jy[8] = {0, 0, 0, dyh21, dyh22, dyh23, dyh31, dyh32};
                if (Jte)
                    for (int a = 0; a < 8; a++) Jte[a] += eX * jx[a] + eY * jy[a];
                if (JtJ)
                    for (int a = 0; a < 8; a++)
                        for (int b = 0; b <= a; b++) JtJ[a * 8 + b] += jx[a] * jx[b] + jy[a] * jy[b];
            }
        }
        if (Sp) *Sp = S;
    }

    // Cholesky factor (lower) of JtJ + lambda * diag(JtJ)... rho.cpp damps the diagonal multiplicatively; false if not positive definite
    static bool chol8_damped(const float* A, float lambda, float* L) {
        const float lambdap1 = lambda + 1.0f;
        for (int i = 0; i < 8; i++)
            for (int j = 0; j <= i; j++) {
                float x = A[i * 8 + j];
                if (i == j) x *= lambdap1;
                for (int k = 0; k < j; k++) x -= L[i * 8 + k] * L[j * 8 + k];
                if (i == j) {
                    if (!(x > 0.0f)) return false;
                    L[i * 8 + i] = std::sqrt(x);
                } else {
                    L[i * 8 + j] = x / L[j * 8 + j];
                }
            }
        return true;
    }
    static void tri_solve8(const float* L, const float* b, float* x) {   // L L^T x = b
        float y[8];
        for (int i = 0; i < 8; i++) {
            float v = b[i];
            for (int k = 0; k < i; k++) v -= L[i * 8 + k] * y[k];
            y[i] = v / L[i * 8 + i];
        }
        for (int i = 7; i >= 0; i--) {
            float v = y[i];
            for (int k = i + 1; k < 8; k++) v -= L[k * 8 + i] * x[k];
            x[i] = v / L[i * 8 + i];
        }
    }

    void refine() {
        float S, newS, L = 100.0f, dH[8], newH[9], JtJ[64], Jte[8], Lf[64];
        jacobian_errors(bestH, JtJ, Jte, &S);
        for (int i = 0; i < MAX_LM_ITERS; i++) {
            while (!chol8_damped(JtJ, L, Lf)) L *= 2.0f;
            tri_solve8(Lf, Jte, dH);
            for (int k = 0; k < 8; k++) newH[k] = bestH[k] - dH[k];
            newH[8] = 1.0f;
            jacobian_errors(newH, nullptr, nullptr, &newS);
            const float dS = S - newS;
            float dL = 0.0f;
            for (int k = 0; k < 8; k++) dL += dH[k] * (L * dH[k] + Jte[k]);
            const float gain = std::fabs(dL) < FLT_EPSILON ? dS : dS / dL;
            if (gain < LM_GAIN_LO) {
                L *= 8;
                if (L > 1000.0f / FLT_EPSILON) break;
            } else if (gain > LM_GAIN_HI) {
                L *= 0.5f;
            }
            if (gain > 0) {
                S = newS;
                std::memcpy(bestH, newH, sizeof(newH));
                jacobian_errors(bestH, JtJ, Jte, &S);
            }
        }
    }

    bool run(double* Hout, uint8_t* mask) {
        curInl.assign(N, 0);
        bestInl.assign(N, 0);
        std::memset(curH, 0, sizeof(curH));
        std::memset(bestH, 0, sizeof(bestH));
        seed(~(uint64_t)0);
        phMax = N;
        {   // expected number of iterations before a sample of the top-4 is all inliers (PROSAC growth function seed)
            double numer = 1, denom = 1;
            for (unsigned i = 0; i < SMPL; i++) {
                numer *= SMPL - i;
                denom *= N - i;
            }
            phEndFpI = rConvg * numer / denom;
        }
        nrTbl.assign(N + 1, 0);
        {
            const double bb = std::sqrt(beta * (1.0 - beta)) * CHI_SQ;
            for (unsigned n = SMPL + 1; n < (unsigned)N + 1; n++) nrTbl[n] = (unsigned)std::ceil(SMPL + n * beta + std::sqrt((double)n) * bb);
        }
        design_sprt();
        // rho.cpp: `for(ctrl.i = 0; ctrl.i < arg.maxI || ctrl.i < 100; ctrl.i++)`: at least 100 iterations whatever the confidence bound says
        for (it = 0; it < maxI || it < 100; it++) {
            if (it >= phEndI && phNum < phMax) {   // next PROSAC phase: one more (lower-ranked) point enters the pool
                phNum++;
                const double next = (phEndFpI * phNum) / (phNum - SMPL);
                phEndI += (unsigned)std::ceil(next - phEndFpI);
                phEndFpI = next;
            }
            if (it > phEndI) {
                rnd_sample(4, smpl, phNum);
            } else {
                rnd_sample(3, smpl, phNum - 1);
                smpl[3] = phNum - 1;
            }
            float pk[16];
            if (sample_degenerate(pk)) continue;
            if (!solve4(pk, curH)) continue;
            evaluate_sprt();
            update_sprt();
            // rho.cpp verify(): evaluateModelSPRT(); updateSPRT(); if (isBestModel()) { saveBestModel(); updateBounds(); nStarOptimize(); } with
            // isBestModel() = curr.numInl > best.numInl - the SPRT's verdict is NOT part of it, so a model rejected after `nTested` points
            // becomes the best one if the inliers counted that far beat the best count. saveBestModel() swaps the two inlier arrays and
            // evaluateModelSPRT() writes only the first nTested flags, so the tail of such a model's array holds whatever an earlier
            // evaluation left there (both arrays start zeroed): curInl / bestInl reproduce exactly that (prefix writes, swap).
            if (curNum > bestNum) {
                std::memcpy(bestH, curH, sizeof(curH));
                bestInl.swap(curInl);
                bestNum = curNum;
                maxI = iter_bound(cfd, (double)bestNum / N, SMPL, maxI);
                nstar_optimize();
            }
        }
        const bool ok = bestNum >= minInl;
        if (ok && bestNum > (unsigned)SMPL) refine();
        for (int i = 0; i < 9; i++) Hout[i] = ok ? (double)bestH[i] : 0.0;
        if (mask)
            for (int i = 0; i < N; i++) mask[i] = ok ? bestInl[i] : 0;
        return ok;
    }
};

}  // namespace

extern "C" int oracle_rho_homography(const float* src_xy, const float* dst_xy, int n, double thr, int max_iters, double confidence, double* H,
                                     uint8_t* mask) {
    if (!src_xy || !dst_xy || !H || n < 4) return -215;
    Rho r{};
    r.src = src_xy;
    r.dst = dst_xy;
    r.N = n;
    r.maxD = (float)(thr <= 0 ? 3 : thr);
    r.maxI = r.rConvg = (unsigned)(max_iters < 1 ? 1 : max_iters);
    r.cfd = confidence;
    r.beta = 0.35;
    r.minInl = 4;
    return r.run(H, mask) ? 1 : 0;
}
