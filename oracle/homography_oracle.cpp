// oracle/homography_oracle.cpp — findHomography restated on the CPU. TEST INFRASTRUCTURE ONLY.
//
// Reference call site: homographier/src/homographier/mod.rs:231-259 (find_homography_mat ->
// opencv::calib3d::find_homography(src, dst, mask, method, thr); 5-argument form => maxIters 2000,
// confidence 0.995). Arithmetic: OpenCV calib3d fundam.cpp (HomographyEstimatorCallback,
// HomographyRefineCallback), ptsetreg.cpp (RANSAC / LMedS registrators, RANSACUpdateNumIters),
// levmarq.cpp (LMSolver), core lapack.cpp (Jacobi eigen), core rand (cv::RNG). Not in /root/reference.
// Pinned by the reference's own test homography_success (mod.rs:437-472), see tests/test_oracle_kat.py.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct P2f { float x, y; };
inline double epsilon_f() { return FLT_EPSILON; }

double hypot_cv(double a, double b) {
    a = std::fabs(a);
    b = std::fabs(b);
    if (a > b) {
        b /= a;
        return a * std::sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * std::sqrt(1 + a * a);
    }
    return 0;
}

// Symmetric eigen decomposition by pivoted Jacobi rotations. A is n x n (upper triangle used, destroyed),
// W = eigenvalues sorted descending, V rows = eigenvectors.
void jacobi_eigen(double* A, int n, double* W, double* V) {
    const double eps = DBL_EPSILON;
    std::vector<int> indR(n), indC(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) V[i * n + j] = 0;
        V[i * n + i] = 1;
    }
    auto scan_row = [&](int k) {
        int m = k + 1;
        double mv = std::fabs(A[n * k + m]);
        for (int i = k + 2; i < n; i++) {
            double val = std::fabs(A[n * k + i]);
            if (mv < val) mv = val, m = i;
        }
        indR[k] = m;
    };
    auto scan_col = [&](int k) {
        int m = 0;
        double mv = std::fabs(A[k]);
        for (int i = 1; i < k; i++) {
            double val = std::fabs(A[n * i + k]);
            if (mv < val) mv = val, m = i;
        }
        indC[k] = m;
    };
    for (int k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) scan_row(k);
        if (k > 0) scan_col(k);
    }
    const int maxIters = n * n * 30;
    if (n > 1)
        for (int iters = 0; iters < maxIters; iters++) {
            int k = 0;
            double mv = std::fabs(A[indR[0]]);
            for (int i = 1; i < n - 1; i++) {
                double val = std::fabs(A[n * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (int i = 1; i < n; i++) {
                double val = std::fabs(A[n * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            double p = A[n * k + l];
            if (std::fabs(p) <= eps) break;
            double y = (W[l] - W[k]) * 0.5;
            double t = std::fabs(y) + hypot_cv(p, y);
            double s = hypot_cv(p, t);
            double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[n * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
            for (int i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
            for (int i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
            for (int i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
            for (int i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
            for (int j = 0; j < 2; j++) {
                int idx = j == 0 ? k : l;
                if (idx < n - 1) scan_row(idx);
                if (idx > 0) scan_col(idx);
            }
        }
    for (int k = 0; k < n - 1; k++) {
        int m = k;
        for (int i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            std::swap(W[m], W[k]);
            for (int i = 0; i < n; i++) std::swap(V[n * m + i], V[n * k + i]);
        }
    }
}

// HomographyEstimatorCallback::runKernel — normalised DLT through the 9x9 normal equations.
int run_kernel(const P2f* M, const P2f* m, int count, double* Hout) {
    double LtL[9][9], W[9], V[9][9];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
        cmx += m[i].x; cmy += m[i].y;
        cMx += M[i].x; cMy += M[i].y;
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += std::fabs(m[i].x - cmx); smy += std::fabs(m[i].y - cmy);
        sMx += std::fabs(M[i].x - cMx); sMy += std::fabs(M[i].y - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON ||
        std::fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx; smy = count / smy;
    sMx = count / sMx; sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    std::memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; i++) {
        double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        double Lx[] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j][k] = LtL[k][j];
    jacobi_eigen(&LtL[0][0], 9, W, &V[0][0]);
    const double* H0 = V[8];
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
    return 1;
}

void compute_error(const P2f* M, const P2f* m, int count, const double* H, float* err) {
    const float Hf[] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    for (int i = 0; i < count; i++) {
        float ww = 1.f / (Hf[6] * M[i].x + Hf[7] * M[i].y + 1.f);
        float dx = (Hf[0] * M[i].x + Hf[1] * M[i].y + Hf[2]) * ww - m[i].x;
        float dy = (Hf[3] * M[i].x + Hf[4] * M[i].y + Hf[5]) * ww - m[i].y;
        err[i] = dx * dx + dy * dy;
    }
}

int find_inliers(const P2f* M, const P2f* m, int count, const double* H, std::vector<float>& err, uint8_t* mask, double thresh) {
    err.resize(count);
    compute_error(M, m, count, H, err.data());
    const float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < count; i++) {
        int f = err[i] <= t;
        mask[i] = (uint8_t)f;
        nz += f;
    }
    return nz;
}

bool have_collinear(const P2f* ptr, int count) {
    int i = count - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = ptr[j].x - ptr[i].x;
        double dy1 = ptr[j].y - ptr[i].y;
        for (int k = 0; k < j; k++) {
            double dx2 = ptr[k].x - ptr[i].x;
            double dy2 = ptr[k].y - ptr[i].y;
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

double det3(const double* a) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

bool check_subset(const P2f* s, const P2f* d, int count) {
    if (have_collinear(s, count) || have_collinear(d, count)) return false;
    if (count == 4) {
        static const int tt[][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int* t = tt[i];
            double A[9] = {s[t[0]].x, s[t[0]].y, 1., s[t[1]].x, s[t[1]].y, 1., s[t[2]].x, s[t[2]].y, 1.};
            double B[9] = {d[t[0]].x, d[t[0]].y, 1., d[t[1]].x, d[t[1]].y, 1., d[t[2]].x, d[t[2]].y, 1.};
            negative += det3(A) * det3(B) < 0;
        }
        if (negative != 0 && negative != 4) return false;
    }
    return true;
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

bool get_subset(const P2f* m1, const P2f* m2, int count, P2f* ms1, P2f* ms2, int* idx, RNG& rng, int maxAttempts) {
    const int modelPoints = 4;
    for (int iters = 0; iters < maxAttempts; ++iters) {
        int i;
        for (i = 0; i < modelPoints; ++i) {
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
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

bool ransac_run(const P2f* m1, const P2f* m2, int count, double threshold, double confidence, int maxIters, double* Hbest,
                uint8_t* bestMaskOut) {
    const int modelPoints = 4;
    int niters = std::max(maxIters, 1);
    int maxGoodCount = 0;
    RNG rng((uint64_t)-1);
    if (count < modelPoints) return false;
    std::vector<uint8_t> mask(count), bestMask(count);
    if (count == modelPoints) {
        if (run_kernel(m1, m2, count, Hbest) <= 0) return false;
        std::memset(bestMaskOut, 1, count);
        return true;
    }
    std::vector<float> err;
    P2f ms1[4], ms2[4];
    int idx[4];
    double model[9];
    for (int iter = 0; iter < niters; iter++) {
        bool found = get_subset(m1, m2, count, ms1, ms2, idx, rng, 10000);
        if (!found) {
            if (iter == 0) return false;
            break;
        }
        if (run_kernel(ms1, ms2, 4, model) <= 0) continue;
        int goodCount = find_inliers(m1, m2, count, model, err, mask.data(), threshold);
        if (goodCount > std::max(maxGoodCount, modelPoints - 1)) {
            std::swap(mask, bestMask);
            std::memcpy(Hbest, model, sizeof(model));
            maxGoodCount = goodCount;
            niters = ransac_update_num_iters(confidence, (double)(count - goodCount) / count, modelPoints, niters);
        }
    }
    if (maxGoodCount > 0) {
        std::memcpy(bestMaskOut, bestMask.data(), count);
        return true;
    }
    return false;
}

bool lmeds_run(const P2f* m1, const P2f* m2, int count, double confidence, int maxIters, double* Hbest, uint8_t* maskOut) {
    const int modelPoints = 4;
    const double outlierRatio = 0.45;
    double minMedian = DBL_MAX;
    RNG rng((uint64_t)-1);
    if (count < modelPoints) return false;
    if (count == modelPoints) {
        if (run_kernel(m1, m2, count, Hbest) <= 0) return false;
        std::memset(maskOut, 1, count);
        return true;
    }
    int niters = ransac_update_num_iters(confidence, outlierRatio, modelPoints, maxIters);
    niters = std::max(niters, 3);
    std::vector<float> err(count);
    P2f ms1[4], ms2[4];
    int idx[4];
    double model[9];
    for (int iter = 0; iter < niters; iter++) {
        bool found = get_subset(m1, m2, count, ms1, ms2, idx, rng, 1000);
        if (!found) {
            if (iter == 0) return false;
            break;
        }
        if (run_kernel(ms1, ms2, 4, model) <= 0) continue;
        compute_error(m1, m2, count, model, err.data());
        // OpenCV sorts the float errors through their int bit patterns (all errors are >= 0)
        int32_t* ie = reinterpret_cast<int32_t*>(err.data());
        std::nth_element(ie, ie + count / 2, ie + count);
        double median = err[count / 2];
        if (median < minMedian) {
            minMedian = median;
            std::memcpy(Hbest, model, sizeof(model));
        }
    }
    if (minMedian < DBL_MAX) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (count - modelPoints)) * std::sqrt(minMedian);
        sigma = std::max(sigma, 0.001);
        std::vector<float> e2;
        int good = find_inliers(m1, m2, count, Hbest, e2, maskOut, sigma);
        return good >= modelPoints;
    }
    return false;
}

// ---- LMSolver (levmarq.cpp), 8 parameters -------------------------------------------------------
void refine_compute(const P2f* M, const P2f* m, int count, const double* h, double* err, double* J) {
    for (int i = 0; i < count; i++) {
        double Mx = M[i].x, My = M[i].y;
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[i * 2] = xi - m[i].x;
        err[i * 2 + 1] = yi - m[i].y;
        if (J) {
            double* Jp = J + (size_t)i * 16;
            Jp[0] = Mx * ww; Jp[1] = My * ww; Jp[2] = ww;
            Jp[3] = Jp[4] = Jp[5] = 0.;
            Jp[6] = -Mx * ww * xi; Jp[7] = -My * ww * xi;
            Jp[8] = Jp[9] = Jp[10] = 0.;
            Jp[11] = Mx * ww; Jp[12] = My * ww; Jp[13] = ww;
            Jp[14] = -Mx * ww * yi; Jp[15] = -My * ww * yi;
        }
    }
}

// x = sum_i (v_i . b / w_i) v_i over eigenpairs with |w_i| > 2*eps*sum(w)   (solve/invert with DECOMP_EIG)
void eig_solve(const double* Asym, int n, const double* b, int nb, double* x) {
    std::vector<double> A(Asym, Asym + n * n), W(n), V(n * n);
    jacobi_eigen(A.data(), n, W.data(), V.data());
    double threshold = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n * nb; i++) x[i] = 0;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int c = 0; c < nb; c++) {
            double s = 0;
            for (int j = 0; j < n; j++) s += V[i * n + j] * b[j * nb + c];
            s *= wi;
            for (int j = 0; j < n; j++) x[j * nb + c] += s * V[i * n + j];
        }
    }
}

void lm_refine(const P2f* M, const P2f* m, int count, double* H, int maxIters) {
    const int lx = 8, lr = count * 2;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    std::vector<double> x(H, H + 8), xd(8), r(lr), rd(lr), J((size_t)lr * 8), A(64), Ap(64), v(8), d(8), temp_d(8), D(8);
    auto normal_eq = [&]() {
        std::fill(A.begin(), A.end(), 0.0);
        std::fill(v.begin(), v.end(), 0.0);
        for (int i = 0; i < lr; i++) {
            const double* Ji = &J[(size_t)i * 8];
            for (int a = 0; a < 8; a++) {
                for (int b = 0; b < 8; b++) A[a * 8 + b] += Ji[a] * Ji[b];
                v[a] += Ji[a] * r[i];
            }
        }
    };
    auto sq = [](const std::vector<double>& z) {
        double s = 0;
        for (double t : z) s += t * t;
        return s;
    };
    auto ninf = [](const std::vector<double>& z) {
        double s = 0;
        for (double t : z) s = std::max(s, std::fabs(t));
        return s;
    };
    refine_compute(M, m, count, x.data(), r.data(), J.data());
    double S = sq(r);
    normal_eq();
    for (int i = 0; i < lx; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        Ap = A;
        for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
        eig_solve(Ap.data(), 8, v.data(), 1, d.data());
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        refine_compute(M, m, count, xd.data(), rd.data(), nullptr);
        double Sd = sq(rd);
        for (int a = 0; a < 8; a++) {   // temp_d = 2*v - A*d
            double s = 0;
            for (int b = 0; b < 8; b++) s += A[a * 8 + b] * d[b];
            temp_d[a] = -s + 2 * v[a];
        }
        double dS = 0;
        for (int a = 0; a < 8; a++) dS += d[a] * temp_d[a];
        double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
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
                eig_solve(A.data(), 8, I8, 8, Ap.data());
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) maxval = std::max(maxval, std::fabs(Ap[i * 8 + i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            std::swap(x, xd);
            refine_compute(M, m, count, x.data(), r.data(), J.data());
            normal_eq();
        }
        iter++;
        bool proceed = iter < maxIters && ninf(d) >= epsx && ninf(r) >= epsf;
        if (!proceed) break;
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
}

// ---- the refit over MANY points: the summation order of the GPU's reduction ---------------------------------------------------
// Above 256 selected points the product sums the per-point terms of runKernel and of the LM normal equations with a parallel
// reduction (csrc/homography.hip reduce_kernel: 64 blocks x 256 threads, thread (b, t) takes the points b*256 + t + 16384 j in
// order; a wave folds its 64 partial sums with the shuffle-down tree 32, 16, ..., 1; the four waves of a block, then the 64 blocks,
// are added in order). Floating-point sums depend on their order, so a sequential loop here could only agree to ~1e-6 (round 1).
// This restates that ORDER ("mirror the device reduction tree on the CPU side"), with the per-point terms written
// as the reference algorithm defines them, so that H is comparable bit for bit at every size. The control flow around the sums
// (runKernel, LMSolver::run) is the sequential code above.
constexpr int MIRROR_MIN = 257, MIRROR_BLOCKS = 64, MIRROR_THREADS = 256;

template <int K, class Term>
void mirrored_sums(int n, const uint8_t* mask, Term term, double* out) {
    std::vector<double> block((size_t)MIRROR_BLOCKS * K, 0.0);
    double lanes[K][64], part[4][K];
    for (int b = 0; b < MIRROR_BLOCKS; b++) {
        for (int w = 0; w < MIRROR_THREADS / 64; w++) {
            for (int l = 0; l < 64; l++) {
                double acc[K];
                for (int k = 0; k < K; k++) acc[k] = 0;
                for (long long i = (long long)b * MIRROR_THREADS + w * 64 + l; i < n; i += (long long)MIRROR_BLOCKS * MIRROR_THREADS)
                    if (!mask || mask[i]) term((int)i, acc);
                for (int k = 0; k < K; k++) lanes[k][l] = acc[k];
            }
            for (int k = 0; k < K; k++) {
                double* v = lanes[k];
                for (int off = 32; off > 0; off >>= 1) {
                    double nv[64];
                    for (int l = 0; l < 64; l++) nv[l] = v[l] + (l + off < 64 ? v[l + off] : v[l]);   // shfl_down: out of range = own value
                    for (int l = 0; l < 64; l++) v[l] = nv[l];
                }
                part[w][k] = v[0];
            }
        }
        for (int k = 0; k < K; k++) {
            double v = 0;
            for (int w = 0; w < MIRROR_THREADS / 64; w++) v += part[w][k];
            block[(size_t)b * K + k] = v;
        }
    }
    for (int k = 0; k < K; k++) {
        double v = 0;
        for (int b = 0; b < MIRROR_BLOCKS; b++) v += block[(size_t)b * K + k];
        out[k] = v;
    }
}

int run_kernel_mirrored(const P2f* M, const P2f* m, int n, const uint8_t* mask, double* Hout) {
    double r[5];
    mirrored_sums<5>(n, mask, [&](int i, double* a) { a[0] += m[i].x; a[1] += m[i].y; a[2] += M[i].x; a[3] += M[i].y; a[4] += 1.0; }, r);
    const double count = r[4];
    if (count < 1) return 0;
    const double cmx = r[0] / count, cmy = r[1] / count, cMx = r[2] / count, cMy = r[3] / count;
    double sc[4];
    mirrored_sums<4>(n, mask, [&](int i, double* a) {
        a[0] += std::fabs(m[i].x - cmx); a[1] += std::fabs(m[i].y - cmy);
        a[2] += std::fabs(M[i].x - cMx); a[3] += std::fabs(M[i].y - cMy);
    }, sc);
    double smx = sc[0], smy = sc[1], sMx = sc[2], sMy = sc[3];
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON) return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double tri[45];
    mirrored_sums<45>(n, mask, [&](int i, double* a) {
        const double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        const double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        int q = 0;
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) a[q++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }, tri);
    double LtL[9][9], W[9], V[9][9];
    int q = 0;
    for (int j = 0; j < 9; j++)
        for (int k = j; k < 9; k++) LtL[j][k] = LtL[k][j] = tri[q++];
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    jacobi_eigen(&LtL[0][0], 9, W, &V[0][0]);
    const double* H0 = V[8];
    double Ht[9], H[9];
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) {
            double sacc = 0;
            for (int k = 0; k < 3; k++) sacc += invHnorm[rr * 3 + k] * H0[k * 3 + c];
            Ht[rr * 3 + c] = sacc;
        }
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) {
            double sacc = 0;
            for (int k = 0; k < 3; k++) sacc += Ht[rr * 3 + k] * Hnorm2[k * 3 + c];
            H[rr * 3 + c] = sacc;
        }
    const double scl = 1. / H[8];
    for (int i = 0; i < 9; i++) Hout[i] = H[i] * scl;
    return 1;
}

// LMSolver::run with the normal equations J^T J, J^T r and |r|^2, |r|_inf summed in the mirrored order
void lm_refine_mirrored(const P2f* M, const P2f* m, int n, const uint8_t* mask, double* H, int maxIters) {
    const int lx = 8;
    double x[8], xd[8], A[64], Ap[64], v[8], d[8], temp_d[8], D[8];
    for (int i = 0; i < 8; i++) x[i] = H[i];
    auto normal_eq = [&](const double* h, bool needJ, double* Aout, double* vout, double& S, double& rinf) {
        double r[45];
        double rmax = 0;
        mirrored_sums<45>(n, mask, [&](int i, double* a) {
            const double Mx = M[i].x, My = M[i].y;
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
            const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            const double r0 = xi - m[i].x, r1 = yi - m[i].y;
            a[44] += r0 * r0 + r1 * r1;
            rmax = std::max(rmax, std::max(std::fabs(r0), std::fabs(r1)));
            if (needJ) {
                const double J0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
                const double J1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
                int q = 0;
                for (int aa = 0; aa < 8; aa++)
                    for (int bb = aa; bb < 8; bb++) a[q++] += J0[aa] * J0[bb] + J1[aa] * J1[bb];
                for (int aa = 0; aa < 8; aa++) a[36 + aa] += J0[aa] * r0 + J1[aa] * r1;
            }
        }, r);
        S = r[44];
        rinf = rmax;
        if (needJ) {
            int q = 0;
            for (int aa = 0; aa < 8; aa++)
                for (int bb = aa; bb < 8; bb++) Aout[aa * 8 + bb] = Aout[bb * 8 + aa] = r[q++];
            for (int aa = 0; aa < 8; aa++) vout[aa] = r[36 + aa];
        }
    };
    double S, rinf, rinf_d;
    normal_eq(x, true, A, v, S, rinf);
    for (int i = 0; i < lx; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        std::memcpy(Ap, A, sizeof(A));
        for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
        eig_solve(Ap, 8, v, 1, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        double Sd;
        normal_eq(xd, false, nullptr, nullptr, Sd, rinf_d);
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
                eig_solve(A, 8, I8, 8, Ap);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) maxval = std::max(maxval, std::fabs(Ap[i * 8 + i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            std::memcpy(x, xd, sizeof(x));
            normal_eq(x, true, A, v, S, rinf);
        }
        iter++;
        double dinf = 0;
        for (int i = 0; i < 8; i++) dinf = std::max(dinf, std::fabs(d[i]));
        const bool proceed = iter < maxIters && dinf >= epsilon_f() && rinf >= epsilon_f();
        if (!proceed) break;
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
}

}  // namespace

extern "C" {

int oracle_homography_4pt(const float* src_xy, const float* dst_xy, int count, double* H) {
    return run_kernel(reinterpret_cast<const P2f*>(src_xy), reinterpret_cast<const P2f*>(dst_xy), count, H);
}

int oracle_ransac_samples(const float* src_xy, const float* dst_xy, int n, int iters, int32_t* idx4) {
    const P2f* m1 = reinterpret_cast<const P2f*>(src_xy);
    const P2f* m2 = reinterpret_cast<const P2f*>(dst_xy);
    RNG rng((uint64_t)-1);
    P2f ms1[4], ms2[4];
    int it = 0;
    for (; it < iters; it++) {
        int idx[4];
        if (!get_subset(m1, m2, n, ms1, ms2, idx, rng, 10000)) break;
        for (int j = 0; j < 4; j++) idx4[it * 4 + j] = idx[j];
    }
    return it;
}

int oracle_find_homography(const float* src_xy, const float* dst_xy, int n, int method, double thr, int max_iters,
                           double confidence, double* H, uint8_t* mask_out) {
    if (!src_xy || !dst_xy || !H) return -215;
    if (n < 4) return -215;   // StsVecLengthErr in OpenCV; the shim maps any error to MatError::Opencv
    // RHO: its own estimator and refinement (rho_oracle.cpp); findHomography skips the generic refit for it. With exactly four pairs every
    // method is the plain 4-point solve: `if( method == 0 || npoints == 4 )` comes first in cv::findHomography (fundam.cpp).
    if (method == 16 && n > 4) {
        std::vector<uint8_t> m(n, 0);
        const int rc = oracle_rho_homography(src_xy, dst_xy, n, thr, max_iters, confidence, H, m.data());
        if (mask_out) std::memcpy(mask_out, m.data(), n);
        return rc;
    }
    if (method != 0 && method != 4 && method != 8 && method != 16) return -5;
    if (thr <= 0) thr = 3;
    const P2f* src = reinterpret_cast<const P2f*>(src_xy);
    const P2f* dst = reinterpret_cast<const P2f*>(dst_xy);
    std::vector<uint8_t> mask(n, 1);
    bool result = false;
    if ((method == 0 || n == 4) && n >= MIRROR_MIN)
        result = run_kernel_mirrored(src, dst, n, nullptr, H) > 0;
    else if (method == 0 || n == 4)
        result = run_kernel(src, dst, n, H) > 0;
    else if (method == 8)
        result = ransac_run(src, dst, n, thr, confidence, max_iters, H, mask.data());
    else
        result = lmeds_run(src, dst, n, confidence, max_iters, H, mask.data());
    int selected = 0;
    for (int i = 0; i < n; i++) selected += mask[i] ? 1 : 0;
    if (result && n > 4 && selected >= MIRROR_MIN) {   // many points: the sums in the order of the GPU's reduction (see mirrored_sums)
        if (method == 8 || method == 4) run_kernel_mirrored(src, dst, n, mask.data(), H);
        lm_refine_mirrored(src, dst, n, mask.data(), H, 10);
    } else if (result && n > 4) {
        std::vector<P2f> s, d;
        for (int i = 0; i < n; i++)
            if (mask[i]) {
                s.push_back(src[i]);
                d.push_back(dst[i]);
            }
        if (!s.empty()) {
            if (method == 8 || method == 4) run_kernel(s.data(), d.data(), (int)s.size(), H);
            lm_refine(s.data(), d.data(), (int)s.size(), H, 10);
        }
    }
    if (!result) {
        std::fill(mask.begin(), mask.end(), 0);
        for (int i = 0; i < 9; i++) H[i] = 0;
    }
    if (mask_out) std::memcpy(mask_out, mask.data(), n);
    return result ? 1 : 0;
}

}  // extern "C"
