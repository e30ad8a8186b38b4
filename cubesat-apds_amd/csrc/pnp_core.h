// csrc/pnp_core.h — EPnP pose from n >= 4 object/image correspondences and the Rodrigues maps, written once for host and device.
//
// Replaces what cv::solvePnP(..., SOLVEPNP_EPNP) does inside cv::solvePnPRansac, the call behind
// homographier::pnp_solver_ransac (homographier/src/homographier/mod.rs:347-361). The same text runs (a) one thread per
// 5-point RANSAC sample with its 12x12 system in LDS (pnp.hip: pnp_hypothesis_kernel) and (b) on the host for the single
// all-inlier solve at the end. Arithmetic contract shared with oracle/pnp_oracle.cpp: IEEE double, one operation per
// source operation (-ffp-contract=off), sums in index order, hypot as sqrt(a*a+b*b), the fixed sin/cos/atan polynomials
// below instead of libm.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace apds {
namespace pnp {

#define PNP_HD __host__ __device__ inline

struct Camera {
    double fu, fv, uc, vc;
};

// Accessors: plain arrays (registers / host memory) or element-major per-thread arrays in LDS
template <class T>
struct Plain {
    T* p;
    PNP_HD T& operator[](int i) const { return p[i]; }
    PNP_HD Plain sub(int off) const { return Plain{p + off}; }
};
template <class T, int STRIDE>
struct Strided {
    T* p;
    PNP_HD T& operator[](int i) const { return p[i * STRIDE]; }
    PNP_HD Strided sub(int off) const { return Strided{p + off * STRIDE}; }
};

struct MwcRng {   // cv::RNG
    uint64_t state;
    PNP_HD unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
};

// ---- elementary functions with a fixed evaluation order ---------------------------------------------------------------
PNP_HD void sincos_fixed(double a, double& s, double& c) {
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const int k = (int)(a * two_over_pi + 0.5);
    const double r = (a - k * pio2_hi) - k * pio2_lo;
    const double r2 = r * r;
    double ps = -7.6471637318198164759e-13;
    ps = ps * r2 + 1.6059043836821614599e-10;
    ps = ps * r2 + -2.5052108385441718775e-08;
    ps = ps * r2 + 2.7557319223985890653e-06;
    ps = ps * r2 + -1.9841269841269841270e-04;
    ps = ps * r2 + 8.3333333333333333333e-03;
    ps = ps * r2 + -1.6666666666666666667e-01;
    const double sr = r + r * (r2 * ps);
    double pc = 4.7794773323873852974e-14;
    pc = pc * r2 + -1.1470745597729724714e-11;
    pc = pc * r2 + 2.0876756987868098979e-09;
    pc = pc * r2 + -2.7557319223985890653e-07;
    pc = pc * r2 + 2.4801587301587301587e-05;
    pc = pc * r2 + -1.3888888888888888889e-03;
    pc = pc * r2 + 4.1666666666666666667e-02;
    pc = pc * r2 + -0.5;
    const double cr = 1.0 + r2 * pc;
    const int q = k & 3;
    s = q == 0 ? sr : (q == 1 ? cr : (q == 2 ? -sr : -cr));
    c = q == 0 ? cr : (q == 1 ? -sr : (q == 2 ? -cr : sr));
}

PNP_HD double atan_fixed(double t) {   // t >= 0
    const bool inv = t > 1.0;
    const double u = inv ? 1.0 / t : t;
    const int k = (int)(u * 8.0 + 0.5);
    // atan(k/8), k = 0..8
    const double tk = k == 0   ? 0.0
                      : k == 1 ? 0.12435499454676144
                      : k == 2 ? 0.24497866312686414
                      : k == 3 ? 0.35877067027057225
                      : k == 4 ? 0.4636476090008061
                      : k == 5 ? 0.5585993153435624
                      : k == 6 ? 0.6435011087932844
                      : k == 7 ? 0.7188299996216245
                               : 0.7853981633974483;
    const double a = k * 0.125;
    const double v = (u - a) / (1.0 + u * a);
    const double v2 = v * v;
    double p = 0.058823529411764705;
    p = p * v2 + -0.06666666666666667;
    p = p * v2 + 0.07692307692307693;
    p = p * v2 + -0.09090909090909091;
    p = p * v2 + 0.1111111111111111;
    p = p * v2 + -0.14285714285714285;
    p = p * v2 + 0.2;
    p = p * v2 + -0.3333333333333333;
    const double r = tk + (v + v * (v2 * p));
    return inv ? 1.5707963267948966 - r : r;
}

PNP_HD double acos_fixed(double c) { return 2.0 * atan_fixed(sqrt((1.0 - c) / (1.0 + c))); }

// ---- one-sided Jacobi SVD on the rows of At (n rows, each m long): cv JacobiSVDImpl_<double> ---------------------------
// Out: W descending, rows of At = left singular vectors, rows of Vt = right singular vectors (skipped when !WANT_V; the
// rotation sequence does not depend on Vt).
template <bool WANT_V, class AT, class WA, class VT>
PNP_HD void svd_rows(AT At, int m, int n, WA W, VT Vt) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    const int max_iter = m > 30 ? m : 30;
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const double t = At[i * m + k];
            sd += t * t;
        }
        W[i] = sd;
        if (WANT_V) {
            for (int k = 0; k < n; k++) Vt[i * n + k] = 0;
            Vt[i * n + i] = 1;
        }
    }
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += At[i * m + k] * At[j * m + k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const double x = At[i * m + k], y = At[j * m + k];
                    const double t0 = c * x + s * y;
                    const double t1 = -s * x + c * y;
                    At[i * m + k] = t0;
                    At[j * m + k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (WANT_V)
                    for (int k = 0; k < n; k++) {
                        const double x = Vt[i * n + k], y = Vt[j * n + k];
                        Vt[i * n + k] = c * x + s * y;
                        Vt[j * n + k] = -s * x + c * y;
                    }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const double t = At[i * m + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            const double wi = W[i];
            W[i] = W[j];
            W[j] = wi;
            for (int k = 0; k < m; k++) {
                const double x = At[i * m + k];
                At[i * m + k] = At[j * m + k];
                At[j * m + k] = x;
            }
            if (WANT_V)
                for (int k = 0; k < n; k++) {
                    const double x = Vt[i * n + k];
                    Vt[i * n + k] = Vt[j * n + k];
                    Vt[j * n + k] = x;
                }
        }
    }
    MwcRng rng{0x12345678};
    for (int i = 0; i < n; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            // exactly-zero singular value: random +-1/m row, made orthogonal to the rows before it, then normalised
            const double val0 = 1. / m;
            for (int k = 0; k < m; k++) At[i * m + k] = (rng.next() & 256) != 0 ? val0 : -val0;
            for (int it = 0; it < 2; it++) {
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < m; k++) sd += At[i * m + k] * At[j * m + k];
                    double asum = 0;
                    for (int k = 0; k < m; k++) {
                        const double t = At[i * m + k] - sd * At[j * m + k];
                        At[i * m + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < m; k++) At[i * m + k] *= asum;
                }
                sd = 0;
                for (int k = 0; k < m; k++) {
                    const double t = At[i * m + k];
                    sd += t * t;
                }
                sd = sqrt(sd);
            }
        }
        const double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

// 3x3 SVD in plain arrays: A row-major -> W[3], Ut rows = left vectors, Vt rows = right vectors
PNP_HD void svd3(const double* A, double* W, double* Ut, double* Vt) {
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Ut[i * 3 + k] = A[k * 3 + i];
    svd_rows<true>(Plain<double>{Ut}, 3, 3, Plain<double>{W}, Plain<double>{Vt});
}

// least squares through the SVD (cv::solve DECOMP_SVD, one right-hand side): A is m x n with m = 6, n <= 5.
// wrk: at least n*6 + n*n + n doubles.
template <class WRK>
PNP_HD void svd_lstsq6(const double* A, int n, const double* b, double* x, WRK wrk) {
    const int m = 6;
    WRK At = wrk, Vt = wrk.sub(n * m), W = wrk.sub(n * m + n * n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++) At[i * m + k] = A[k * n + i];
    svd_rows<true>(At, m, n, W, Vt);
    double threshold = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < n; j++) x[j] = 0;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++) s += At[i * m + j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] = x[j] + s * Vt[i * n + j];
    }
}

PNP_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PNP_HD double sqdist3(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// ---- Rodrigues (calibration.cpp) ----------------------------------------------------------------------------------------
PNP_HD void rotation_from_rvec(const double* rv, double* R) {
    const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i == 0 || i == 4 || i == 8) ? 1. : 0.;
        return;
    }
    double s, c;
    sincos_fixed(theta, s, c);
    const double c1 = 1. - c, itheta = 1. / theta;
    const double rx = rv[0] * itheta, ry = rv[1] * itheta, rz = rv[2] * itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double skew[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int i = 0; i < 9; i++) {
        const double eye = (i == 0 || i == 4 || i == 8) ? 1. : 0.;
        R[i] = (c * eye + c1 * rrt[i]) + s * skew[i];
    }
}

PNP_HD void rvec_from_rotation(const double* Rin, double* rv) {
    double W[3], Ut[9], Vt[9], R[9];
    svd3(Rin, W, Ut, Vt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + Ut[6 + i] * Vt[6 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos_fixed(c);
    if (s < 1e-5) {
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    rv[0] = rx;
    rv[1] = ry;
    rv[2] = rz;
}

// PnPRansacCallback::computeError for one point: projectPoints with zero distortion, float output, float squared distance
PNP_HD float reprojection_sqerr(const double* R, const double* t, const Camera& cam, float X, float Y, float Z, float u, float v) {
    const double dX = X, dY = Y, dZ = Z;
    double x = R[0] * dX + R[1] * dY + R[2] * dZ + t[0];
    double y = R[3] * dX + R[4] * dY + R[5] * dZ + t[1];
    double z = R[6] * dX + R[7] * dY + R[8] * dZ + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    const float px = (float)(x * cam.fu + cam.uc), py = (float)(y * cam.fv + cam.vc);
    const float dx = u - px, dy = v - py;
    return dx * dx + dy * dy;
}

// ---- EPnP (epnp.cpp) ------------------------------------------------------------------------------------------------
// element (k, c) of the 2n x 12 matrix M: rows 2p / 2p+1 belong to point p
template <class AL, class US>
PNP_HD double m_entry(const AL& alphas, const US& us, const Camera& cam, int k, int c) {
    const int p = k >> 1, j = c / 3, comp = c - 3 * j;
    const double a = alphas[4 * p + j];
    if ((k & 1) == 0) return comp == 0 ? a * cam.fu : (comp == 1 ? 0.0 : a * (cam.uc - us[2 * p]));
    return comp == 0 ? 0.0 : (comp == 1 ? a * cam.fv : a * (cam.vc - us[2 * p + 1]));
}

// Householder least squares for the 6x4 Gauss-Newton step, as epnp.cpp writes it: the pivot scan looks at rows k..nr-2
// only, and an all-zero column returns silently leaving X as it was.
PNP_HD void qr_solve_6x4(double* A, double* b, double* X) {
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    for (int k = 0; k < nc; k++) {
        double eta = fabs(A[k * nc + k]);
        for (int i = k; i < nr - 1; i++) {
            const double elt = fabs(A[i * nc + k]);
            if (eta < elt) eta = elt;
        }
        if (eta == 0) return;
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) {
            A[i * nc + k] *= inv_eta;
            sum2 += A[i * nc + k] * A[i * nc + k];
        }
        double sigma = sqrt(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
            double sum = 0;
            for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
            const double tau = sum / A1[k];
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    for (int j = 0; j < nc; j++) {
        double tau = 0;
        for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double sum = 0;
        for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

PNP_HD void gauss_newton_betas(const double* L, const double* rho, double* betas) {
    double A[24], b[6], x[4] = {0, 0, 0, 0};
    for (int it = 0; it < 5; it++) {
        for (int i = 0; i < 6; i++) {
            const double* rowL = L + i * 10;
            double* rowA = A + i * 4;
            rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
            rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
            rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
            rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
            b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                             rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                             rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                             rowL[9] * betas[3] * betas[3]);
        }
        qr_solve_6x4(A, b, x);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// Pose for one beta vector: control points in the camera frame -> points in the camera frame -> absolute orientation.
// Returns the mean reprojection error (pixels).
template <class PW, class US, class AL, class PC, class UT>
PNP_HD double pose_from_betas(int n, const PW& pws, const US& us, const AL& alphas, PC pcs, const Camera& cam, const UT& ut, const double* betas,
                              double* R /*9*/, double* t /*3*/) {
    double ccs[12];
    for (int i = 0; i < 12; i++) ccs[i] = 0.0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) ccs[3 * j + k] += betas[i] * ut[12 * (11 - i) + 3 * j + k];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++)
            pcs[3 * i + j] = alphas[4 * i] * ccs[j] + alphas[4 * i + 1] * ccs[3 + j] + alphas[4 * i + 2] * ccs[6 + j] + alphas[4 * i + 3] * ccs[9 + j];
    if (pcs[2] < 0.0)   // the first point must be in front of the camera (ccs is not used again: only pcs flips)
        for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            pc0[j] += pcs[3 * i + j];
            pw0[j] += pws[3 * i + j];
        }
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            const double dc = pcs[3 * i + j] - pc0[j];
            abt[3 * j] += dc * (pws[3 * i] - pw0[0]);
            abt[3 * j + 1] += dc * (pws[3 * i + 1] - pw0[1]);
            abt[3 * j + 2] += dc * (pws[3 * i + 2] - pw0[2]);
        }
    double d[3], u_t[9], v_t[9];
    svd3(abt, d, u_t, v_t);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = u_t[i] * v_t[j] + u_t[3 + i] * v_t[3 + j] + u_t[6 + i] * v_t[6 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) {
        R[6] = -R[6];
        R[7] = -R[7];
        R[8] = -R[8];
    }
    t[0] = pc0[0] - dot3(R, pw0);
    t[1] = pc0[1] - dot3(R + 3, pw0);
    t[2] = pc0[2] - dot3(R + 6, pw0);
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
        const double pw[3] = {pws[3 * i], pws[3 * i + 1], pws[3 * i + 2]};
        const double Xc = dot3(R, pw) + t[0];
        const double Yc = dot3(R + 3, pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        const double ue = cam.uc + cam.fu * Xc * inv_Zc;
        const double ve = cam.vc + cam.fv * Yc * inv_Zc;
        const double u = us[2 * i], v = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

// pws: 3n object coordinates; us: 2n pixel coordinates (already "undistorted and re-projected", see load step of the
// callers); alphas: 4n and pcs: 3n of workspace. big: 156 doubles (12x12 system, then its left singular vectors, + 12
// singular values). wrk: 64 doubles.
template <class PW, class US, class AL, class PC, class BIG, class WRK>
PNP_HD void epnp_pose(int n, const PW& pws, const US& us, AL alphas, PC pcs, const Camera& cam, BIG big, WRK wrk, double* R, double* t) {
    // control points: centroid + principal directions scaled by sqrt(lambda / n)
    double cws[12];
    cws[0] = cws[1] = cws[2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) cws[j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[j] /= n;
    {
        double cov[9], dc[3], uct[9], vt[9];
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < n; k++) s += (pws[3 * k + i] - cws[i]) * (pws[3 * k + j] - cws[j]);
                cov[i * 3 + j] = cov[j * 3 + i] = s;
            }
        svd3(cov, dc, uct, vt);
        for (int i = 1; i < 4; i++) {
            const double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[3 * i + j] = cws[j] + k * uct[3 * (i - 1) + j];
        }
    }
    // barycentric coordinates: inverse of the control-point frame through its SVD (cvInvert CV_SVD)
    {
        double cc[9], ci[9], W[3], Ut[9], Vt[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[3 * j + i] - cws[i];
        svd3(cc, W, Ut, Vt);
        const double threshold = (W[0] + W[1] + W[2]) * (DBL_EPSILON * 2);
        for (int j = 0; j < 9; j++) ci[j] = 0;
        for (int i = 0; i < 3; i++) {
            double wi = W[i];
            if (fabs(wi) <= threshold) continue;
            wi = 1 / wi;
            const double buf[3] = {Ut[i * 3] * wi, Ut[i * 3 + 1] * wi, Ut[i * 3 + 2] * wi};
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) ci[j * 3 + k] = ci[j * 3 + k] + Vt[i * 3 + j] * buf[k];
        }
        for (int i = 0; i < n; i++) {
            const double d0 = pws[3 * i] - cws[0], d1 = pws[3 * i + 1] - cws[1], d2 = pws[3 * i + 2] - cws[2];
            double a[4];
            for (int j = 0; j < 3; j++) a[1 + j] = ci[3 * j] * d0 + ci[3 * j + 1] * d1 + ci[3 * j + 2] * d2;
            a[0] = 1.0f - a[1] - a[2] - a[3];
            for (int j = 0; j < 4; j++) alphas[4 * i + j] = a[j];
        }
    }
    // M^T M and its singular vectors; the four with the smallest singular values span the candidate solutions
    BIG ut = big, d = big.sub(144);
    // (row-outer accumulation: every entry still sums its products in row order k = 0..2n-1, as cvMulTransposed does)
    for (int i = 0; i < 12; i++)
        for (int j = i; j < 12; j++) ut[i * 12 + j] = 0;
    for (int k = 0; k < 2 * n; k++) {
        double row[12];
        for (int c = 0; c < 12; c++) row[c] = m_entry(alphas, us, cam, k, c);
        for (int i = 0; i < 12; i++)
            for (int j = i; j < 12; j++) ut[i * 12 + j] += row[i] * row[j];
    }
    for (int i = 0; i < 12; i++)
        for (int j = i + 1; j < 12; j++) ut[j * 12 + i] = ut[i * 12 + j];
    svd_rows<false>(ut, 12, 12, d, d);
    double L[60], rho[6];
    {
        for (int r = 0, a = 0, b = 1; r < 6; r++) {
            double dv[4][3];
            for (int i = 0; i < 4; i++)
                for (int c = 0; c < 3; c++) dv[i][c] = ut[12 * (11 - i) + 3 * a + c] - ut[12 * (11 - i) + 3 * b + c];
            double* row = L + 10 * r;
            row[0] = dot3(dv[0], dv[0]);
            row[1] = 2.0f * dot3(dv[0], dv[1]);
            row[2] = dot3(dv[1], dv[1]);
            row[3] = 2.0f * dot3(dv[0], dv[2]);
            row[4] = 2.0f * dot3(dv[1], dv[2]);
            row[5] = dot3(dv[2], dv[2]);
            row[6] = 2.0f * dot3(dv[0], dv[3]);
            row[7] = 2.0f * dot3(dv[1], dv[3]);
            row[8] = 2.0f * dot3(dv[2], dv[3]);
            row[9] = dot3(dv[3], dv[3]);
            b++;
            if (b > 3) {
                a++;
                b = a + 1;
            }
        }
        rho[0] = sqdist3(cws, cws + 3);
        rho[1] = sqdist3(cws, cws + 6);
        rho[2] = sqdist3(cws, cws + 9);
        rho[3] = sqdist3(cws + 3, cws + 6);
        rho[4] = sqdist3(cws + 3, cws + 9);
        rho[5] = sqdist3(cws + 6, cws + 9);
    }
    double best_err = 0;
    for (int approx = 1; approx <= 3; approx++) {
        double betas[4], sub[30], sol[5];
        if (approx == 1) {   // betas10 columns [B11 B12 B13 B14]
            for (int i = 0; i < 6; i++) {
                sub[4 * i] = L[10 * i];
                sub[4 * i + 1] = L[10 * i + 1];
                sub[4 * i + 2] = L[10 * i + 3];
                sub[4 * i + 3] = L[10 * i + 6];
            }
            svd_lstsq6(sub, 4, rho, sol, wrk);
            if (sol[0] < 0) {
                betas[0] = sqrt(-sol[0]);
                betas[1] = -sol[1] / betas[0];
                betas[2] = -sol[2] / betas[0];
                betas[3] = -sol[3] / betas[0];
            } else {
                betas[0] = sqrt(sol[0]);
                betas[1] = sol[1] / betas[0];
                betas[2] = sol[2] / betas[0];
                betas[3] = sol[3] / betas[0];
            }
        } else {             // [B11 B12 B22] or [B11 B12 B22 B13 B23]
            const int nc = approx == 2 ? 3 : 5;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < nc; j++) sub[nc * i + j] = L[10 * i + j];
            svd_lstsq6(sub, nc, rho, sol, wrk);
            if (sol[0] < 0) {
                betas[0] = sqrt(-sol[0]);
                betas[1] = (sol[2] < 0) ? sqrt(-sol[2]) : 0.0;
            } else {
                betas[0] = sqrt(sol[0]);
                betas[1] = (sol[2] > 0) ? sqrt(sol[2]) : 0.0;
            }
            if (sol[1] < 0) betas[0] = -betas[0];
            betas[2] = approx == 2 ? 0.0 : sol[3] / betas[0];
            betas[3] = 0.0;
        }
        gauss_newton_betas(L, rho, betas);
        double Rc[9], tc[3];
        const double err = pose_from_betas(n, pws, us, alphas, pcs, cam, ut, betas, Rc, tc);
        if (approx == 1 || err < best_err) {
            best_err = err;
            for (int i = 0; i < 9; i++) R[i] = Rc[i];
            for (int i = 0; i < 3; i++) t[i] = tc[i];
        }
    }
}

}  // namespace pnp
}  // namespace apds
