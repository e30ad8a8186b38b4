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

// ---- P3P (p3p.cpp: Gao, Hou, Tang, Cheng 2003; polynom_solver.cpp) -------------------------------------------------------
// The kernel solvePnPRansac uses for SOLVEPNP_P3P and whenever only four correspondences exist: up to four poses from three
// points (a quartic in the ratio of two of the camera-point distances, Horn's quaternion alignment), ranked by the fourth.
PNP_HD double cbrt_fixed(double x) {   // x > 0; OpenCV calls pow(x, 1/3.): bit-level first guess + 6 Newton steps, fixed order
    union {
        double d;
        uint64_t u;
    } v;
    v.d = x;
    v.u = v.u / 3 + 0x2A9F7893782DA1CEull;
    double y = v.d;
    for (int k = 0; k < 6; k++) y = y - (y * y * y - x) / (3.0 * (y * y));
    return y;
}
PNP_HD double cos_fixed(double a) {   // a in [0, 2 pi]
    double s, c;
    sincos_fixed(a, s, c);
    return c;
}
PNP_HD double nan_value() {
    union {
        uint64_t u;
        double d;
    } v;
    v.u = 0x7FF8000000000000ull;
    return v.d;
}

PNP_HD int poly_deg2(double a, double b, double c, double& x1, double& x2) {
    const double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    const double inv_2a = 0.5 / a;
    if (delta == 0) {
        x1 = -b * inv_2a;
        x2 = x1;
        return 1;
    }
    const double sqrt_delta = sqrt(delta);
    x1 = (-b + sqrt_delta) * inv_2a;
    x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

PNP_HD int poly_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        return poly_deg2(b, c, d, x0, x1);
    }
    const double inv_a = 1. / a;
    const double b_a = inv_a * b, b_a2 = b_a * b_a;
    const double c_a = inv_a * c;
    const double d_a = inv_a * d;
    const double Q = (3 * c_a - b_a2) / 9;
    const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) {
            x0 = x1 = x2 = -b_a_3;
            return 3;
        }
        x0 = (2 * R > 0 ? cbrt_fixed(2 * R) : nan_value()) - b_a_3;   // pow(negative, 1/3.) is NaN
        return 1;
    }
    if (D <= 0) {   // three real roots
        const double theta = acos_fixed(R / sqrt(-Q3));
        const double sqrt_Q = sqrt(-Q);
        x0 = 2 * sqrt_Q * cos_fixed(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * cos_fixed((theta + 2 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * cos_fixed((theta + 4 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        return 3;
    }
    const double AD = cbrt_fixed(fabs(R) + sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

PNP_HD int poly_deg4(double a, double b, double c, double d, double e, double& x0, double& x1, double& x2, double& x3) {
    if (a == 0) {
        x3 = 0;
        return poly_deg3(b, c, d, e, x0, x1, x2);
    }
    const double inv_a = 1. / a;
    b *= inv_a;
    c *= inv_a;
    d *= inv_a;
    e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    const int n = poly_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
    if (n == 0) return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double R = sqrt(R2);
    const double inv_R = 1. / R;
    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            const double sqrt_temp = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
            E2 = D2 - 4 * sqrt_temp;
        }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        const double D = sqrt(D2);
        nb_real_roots = 2;
        const double D_2 = 0.5 * D;
        x0 = R_2 + D_2 - b_4;
        x1 = x0 - D;
    }
    if (E2 >= 0) {
        const double E = sqrt(E2);
        const double E_2 = 0.5 * E;
        if (nb_real_roots == 0) {
            x0 = -R_2 + E_2 - b_4;
            x1 = x0 - E;
            nb_real_roots = 2;
        } else {
            x2 = -R_2 + E_2 - b_4;
            x3 = x2 - E;
            nb_real_roots = 4;
        }
    }
    return nb_real_roots;
}

// cyclic Jacobi on a symmetric 4x4 (upper triangle of A used and destroyed): eigenvalues D, eigenvectors in the columns of U
PNP_HD bool eigen_sym4(double* A, double* D, double* U) {
    double B[4], Z[4];
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 4; i++) {
        B[i] = A[5 * i];
        D[i] = B[i];
        Z[i] = 0;
    }
    for (int iter = 0; iter < 50; iter++) {
        const double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return true;
        const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++)
            for (int j = i + 1; j < 4; j++) {
                const double Aij = A[4 * i + j];
                const double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) {
                    A[4 * i + j] = 0.0;
                } else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
                    else {
                        const double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh;
                    Z[j] += hh;
                    D[i] -= hh;
                    D[j] += hh;
                    A[4 * i + j] = 0.0;
                    const double c = 1.0 / sqrt(1 + t * t);
                    const double s = t * c;
                    const double tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; k++) {
                        const double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = i + 1; k <= j - 1; k++) {
                        const double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = j + 1; k < 4; k++) {
                        const double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (int k = 0; k < 4; k++) {
                        const double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
            }
        for (int i = 0; i < 4; i++) {
            B[i] += Z[i];
            D[i] = B[i];
            Z[i] = 0;
        }
    }
    return false;
}

// Horn: rotation + translation taking the three object points P0..P2 onto the camera-frame points M (rows)
PNP_HD void p3p_align(const double* M /*3x3*/, const double* P0, const double* P1, const double* P2, double* R /*9*/, double* T /*3*/) {
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; i++) C_end[i] = (M[i] + M[3 + i] + M[6 + i]) / 3;
    C_start[0] = (P0[0] + P1[0] + P2[0]) / 3;
    C_start[1] = (P0[1] + P1[1] + P2[1]) / 3;
    C_start[2] = (P0[2] + P1[2] + P2[2]) / 3;
    double s[9];
    for (int j = 0; j < 3; j++) {
        s[0 * 3 + j] = (P0[0] * M[j] + P1[0] * M[3 + j] + P2[0] * M[6 + j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (P0[1] * M[j] + P1[1] * M[3 + j] + P2[1] * M[6 + j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (P0[2] * M[j] + P1[2] * M[3 + j] + P2[2] * M[6 + j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
    Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
    Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
    Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    eigen_sym4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; i++)
        if (evs[i] > ev_max) {
            i_ev = i;
            ev_max = evs[i];
        }
    const double q0 = U[0 * 4 + i_ev], q1 = U[1 * 4 + i_ev], q2 = U[2 * 4 + i_ev], q3 = U[3 * 4 + i_ev];
    const double q02 = q0 * q0, q12 = q1 * q1, q22 = q2 * q2, q32 = q3 * q3;
    const double q0_1 = q0 * q1, q0_2 = q0 * q2, q0_3 = q0 * q3;
    const double q1_2 = q1 * q2, q1_3 = q1 * q3;
    const double q2_3 = q2 * q3;
    R[0] = q02 + q12 - q22 - q32;
    R[1] = 2. * (q1_2 - q0_3);
    R[2] = 2. * (q1_3 + q0_2);
    R[3] = 2. * (q1_2 + q0_3);
    R[4] = q02 + q22 - q12 - q32;
    R[5] = 2. * (q2_3 - q0_1);
    R[6] = 2. * (q1_3 - q0_2);
    R[7] = 2. * (q2_3 + q0_1);
    R[8] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[3 * i] * C_start[0] + R[3 * i + 1] * C_start[1] + R[3 * i + 2] * C_start[2]);
}

// the camera-to-point distances: lengths[k] = (X, Y, Z) for solution k; returns the number of solutions (<= 4)
PNP_HD int p3p_lengths(double lengths[4][3], const double* distances, const double* cosines) {
    const double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    const double inv_d22 = 1. / (distances[2] * distances[2]);
    const double a = inv_d22 * (distances[0] * distances[0]);
    const double b = inv_d22 * (distances[1] * distances[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double roots[4];
    const int n = poly_deg4(A, B, C, D, E, roots[0], roots[1], roots[2], roots[3]);
    if (n == 0) return 0;
    int nb_solutions = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        const double x = roots[i];
        if (x <= 0) continue;
        const double x2 = x * x;
        const double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        const double y = inv_b0 * b1;
        const double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        const double Z = distances[2] / sqrt(v);
        lengths[nb_solutions][0] = x * Z;
        lengths[nb_solutions][1] = y * Z;
        lengths[nb_solutions][2] = Z;
        nb_solutions++;
    }
    return nb_solutions;
}

// solvePnP(SOLVEPNP_P3P) on four correspondences: mu/mv pixel coordinates (already through undistortPoints), P object points.
// Writes the first pose of OpenCV's list sorted by the fourth point's error.
PNP_HD bool p3p_best_pose(const Camera& cam, const double* mu_in, const double* mv_in, const double* P /*4x3*/, double* Rbest /*9*/, double* tbest /*3*/) {
    const double inv_fx = 1. / cam.fu, inv_fy = 1. / cam.fv, cx_fx = cam.uc / cam.fu, cy_fy = cam.vc / cam.fv;
    double mu[4], mv[4], mk[3];
    for (int i = 0; i < 3; i++) {
        mu[i] = inv_fx * mu_in[i] - cx_fx;
        mv[i] = inv_fy * mv_in[i] - cy_fy;
        const double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm;
        mu[i] *= mk[i];
        mv[i] *= mk[i];
    }
    mu[3] = inv_fx * mu_in[3] - cx_fx;
    mv[3] = inv_fy * mv_in[3] - cy_fy;
    const double* P0 = P;
    const double* P1 = P + 3;
    const double* P2 = P + 6;
    const double* P3 = P + 9;
    double distances[3], cosines[3];
    distances[0] = sqrt((P1[0] - P2[0]) * (P1[0] - P2[0]) + (P1[1] - P2[1]) * (P1[1] - P2[1]) + (P1[2] - P2[2]) * (P1[2] - P2[2]));
    distances[1] = sqrt((P0[0] - P2[0]) * (P0[0] - P2[0]) + (P0[1] - P2[1]) * (P0[1] - P2[1]) + (P0[2] - P2[2]) * (P0[2] - P2[2]));
    distances[2] = sqrt((P0[0] - P1[0]) * (P0[0] - P1[0]) + (P0[1] - P1[1]) * (P0[1] - P1[1]) + (P0[2] - P1[2]) * (P0[2] - P1[2]));
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    double lengths[4][3];
    const int n = p3p_lengths(lengths, distances, cosines);
    double Rs[4][9], ts[4][3], errs[4];
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < n; i++) {
        double M[9];
        for (int j = 0; j < 3; j++) {
            M[3 * j] = lengths[i][j] * mu[j];
            M[3 * j + 1] = lengths[i][j] * mv[j];
            M[3 * j + 2] = lengths[i][j] * mk[j];
        }
        p3p_align(M, P0, P1, P2, Rs[i], ts[i]);
        const double X3p = Rs[i][0] * P3[0] + Rs[i][1] * P3[1] + Rs[i][2] * P3[2] + ts[i][0];
        const double Y3p = Rs[i][3] * P3[0] + Rs[i][4] * P3[1] + Rs[i][5] * P3[2] + ts[i][1];
        const double Z3p = Rs[i][6] * P3[0] + Rs[i][7] * P3[1] + Rs[i][8] * P3[2] + ts[i][2];
        const double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        errs[i] = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
    }
    // OpenCV's insertion sort by the fourth point's error (a NaN neither moves nor lets anything pass it); the head is returned
    for (int i = 1; i < n; i++)
        for (int j = i; j > 0 && errs[j - 1] > errs[j]; j--) {
            const double e = errs[j];
            errs[j] = errs[j - 1];
            errs[j - 1] = e;
            const int o = order[j];
            order[j] = order[j - 1];
            order[j - 1] = o;
        }
    if (n <= 0) return false;
    // (a switch keeps Rs / ts in registers on the device: no dynamically indexed copy)
    const int h = order[0];
    for (int k = 0; k < 9; k++) Rbest[k] = h == 0 ? Rs[0][k] : (h == 1 ? Rs[1][k] : (h == 2 ? Rs[2][k] : Rs[3][k]));
    for (int k = 0; k < 3; k++) tbest[k] = h == 0 ? ts[0][k] : (h == 1 ? ts[1][k] : (h == 2 ? ts[2][k] : ts[3][k]));
    return true;
}

// ---- AP3P (ap3p.cpp: Ke, Roumeliotis 2017) --------------------------------------------------------------------------------------------
// The kernel solvePnPRansac uses for SOLVEPNP_AP3P (mod.rs:327,359 hands `Option<SolvePnPMethod>` straight through): a quartic in
// cos(theta1') from the seven g coefficients, its four roots by Ferrari's closed form in complex arithmetic (the real parts of all four are
// kept, as upstream does), two Newton polishing steps, rotation and translation per root with |cos| <= 1, poses ranked by the fourth
// point. Written once for host and device; elementary functions in fixed order, so the oracle's separate restatement agrees bit for bit.
struct Cx {
    double re, im;
};
PNP_HD Cx cx_sqrt(Cx z) {   // principal square root
    if (z.re == 0.0) {
        const double t = sqrt(fabs(z.im) / 2);
        return Cx{t, z.im < 0.0 ? -t : t};
    }
    const double m = sqrt(z.re * z.re + z.im * z.im);
    const double t = sqrt(2 * (m + fabs(z.re)));
    const double u = t / 2;
    return z.re > 0.0 ? Cx{u, z.im / t} : Cx{fabs(z.im) / t, z.im < 0.0 ? -u : u};
}
PNP_HD double atan2_fixed(double y, double x) {   // y != 0 or x != 0
    const double ax = fabs(x), ay = fabs(y);
    double a = ax >= ay ? atan_fixed(ay / ax) : 1.5707963267948966 - atan_fixed(ax / ay);
    if (x < 0) a = 3.141592653589793 - a;
    return y < 0 ? -a : a;
}
PNP_HD Cx cx_cbrt(Cx z) {   // principal value of pow(z, 1/3)
    const double m = sqrt(z.re * z.re + z.im * z.im);
    const double r = cbrt_fixed(m), th = atan2_fixed(z.im, z.re) / 3.0;
    double sn, cs;
    sincos_fixed(fabs(th), sn, cs);
    return Cx{r * cs, th < 0 ? -(r * sn) : r * sn};
}
PNP_HD void ap3p_quartic(const double* f, double* roots) {
    const double a4 = f[0], a3 = f[1], a2 = f[2], a1 = f[3], a0 = f[4];
    const double a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
    const double p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
    const double q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
    const double r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
    const double p3 = ((p4 * p4) / 12 + r4) / 3;
    const double q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
    double t;
    Cx w = cx_sqrt(Cx{q3 * q3 - p3 * p3 * p3, 0.0});
    if (q3 >= 0) w = Cx{-w.re - q3, -w.im};
    else w = Cx{w.re - q3, w.im};
    if (w.im == 0.0) {
        const double c = w.re < 0 ? -cbrt_fixed(-w.re) : (w.re > 0 ? cbrt_fixed(w.re) : 0.0);
        t = 2.0 * (c + p3 / c);
    } else {
        t = 4.0 * cx_cbrt(w).re;
    }
    const Cx sqrt_2m = cx_sqrt(Cx{-2 * p4 / 3 + t, 0.0});
    const double B_4A = -a3 / (4 * a4);
    const double complex1 = 4 * p4 / 3 + t;
    const double den = sqrt_2m.re * sqrt_2m.re + sqrt_2m.im * sqrt_2m.im;   // 2 q4 / sqrt_2m
    const Cx complex2{2 * q4 * sqrt_2m.re / den, -(2 * q4 * sqrt_2m.im) / den};
    const double sqrt_2m_rh = sqrt_2m.re / 2;
    const double sqrt1 = cx_sqrt(Cx{-(complex1 + complex2.re), -complex2.im}).re / 2;
    roots[0] = B_4A + sqrt_2m_rh + sqrt1;
    roots[1] = B_4A + sqrt_2m_rh - sqrt1;
    const double sqrt_2m_lh = -sqrt_2m_rh;
    const double sqrt2 = cx_sqrt(Cx{-(complex1 - complex2.re), complex2.im}).re / 2;
    roots[2] = B_4A + sqrt_2m_lh + sqrt2;
    roots[3] = B_4A + sqrt_2m_lh - sqrt2;
    for (int it = 0; it < 2; it++)   // polishQuarticRoots
        for (int j = 0; j < 4; j++) {
            const double x = roots[j];
            const double error = (((f[0] * x + f[1]) * x + f[2]) * x + f[3]) * x + f[4];
            const double derivative = ((4 * f[0] * x + 3 * f[1]) * x + 2 * f[2]) * x + f[3];
            roots[j] -= error / derivative;
        }
}
PNP_HD void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
PNP_HD void mult3(const double* a, const double* b, double* r) {   // 3 x 3 row major
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
// solvePnP(SOLVEPNP_AP3P) on four correspondences (as p3p_best_pose): the first pose of the list sorted by the fourth point's error
PNP_HD bool ap3p_best_pose(const Camera& cam, const double* mu_in, const double* mv_in, const double* P /*4x3*/, double* Rbest /*9*/, double* tbest /*3*/) {
    const double inv_fx = 1. / cam.fu, inv_fy = 1. / cam.fv, cx_fx = cam.uc / cam.fu, cy_fy = cam.vc / cam.fv;
    double b0[3], b1[3], b2[3];
    {
        double* bs[3] = {b0, b1, b2};
        for (int i = 0; i < 3; i++) {
            const double mu = inv_fx * mu_in[i] - cx_fx, mv = inv_fy * mv_in[i] - cy_fy;
            const double mk = 1. / sqrt(mu * mu + mv * mv + 1);
            bs[i][0] = mu * mk;
            bs[i][1] = mv * mk;
            bs[i][2] = mk;
        }
    }
    const double mu3 = inv_fx * mu_in[3] - cx_fx, mv3 = inv_fy * mv_in[3] - cy_fy;
    const double *w1 = P, *w2 = P + 3, *w3 = P + 6, *w4 = P + 9;
    const double u0[3] = {w1[0] - w2[0], w1[1] - w2[1], w1[2] - w2[2]};
    const double nu0 = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
    const double k1[3] = {u0[0] / nu0, u0[1] / nu0, u0[2] / nu0};
    double k3[3], tz[3], v1[3], v2[3];
    cross3(b0, b1, k3);
    const double nk3 = sqrt(k3[0] * k3[0] + k3[1] * k3[1] + k3[2] * k3[2]);
    for (int i = 0; i < 3; i++) k3[i] /= nk3;
    cross3(b0, k3, tz);
    cross3(b0, b2, v1);
    cross3(b1, b2, v2);
    const double u1[3] = {w1[0] - w3[0], w1[1] - w3[1], w1[2] - w3[2]};
    const double u1k1 = u1[0] * k1[0] + u1[1] * k1[1] + u1[2] * k1[2];
    const double k3b3 = k3[0] * b2[0] + k3[1] * b2[1] + k3[2] * b2[2];
    double f11 = k3b3;
    double f13 = k3[0] * v1[0] + k3[1] * v1[1] + k3[2] * v1[2];
    const double f15 = -u1k1 * f11;
    double nl[3];
    cross3(u1, k1, nl);
    const double delta = sqrt(nl[0] * nl[0] + nl[1] * nl[1] + nl[2] * nl[2]);
    for (int i = 0; i < 3; i++) nl[i] /= delta;
    f11 *= delta;
    f13 *= delta;
    const double u2k1 = u1k1 - nu0;
    double f21 = tz[0] * v2[0] + tz[1] * v2[1] + tz[2] * v2[2];
    double f22 = nk3 * k3b3;
    double f23 = k3[0] * v2[0] + k3[1] * v2[1] + k3[2] * v2[2];
    const double f24 = u2k1 * f22;
    const double f25 = -u2k1 * f21;
    f21 *= delta;
    f22 *= delta;
    f23 *= delta;
    const double g1 = f13 * f22, g2 = f13 * f25 - f15 * f23, g3 = f11 * f23 - f13 * f21, g4 = -f13 * f24, g5 = f11 * f22, g6 = f11 * f25 - f15 * f21,
                 g7 = -f15 * f24;
    const double coeffs[5] = {g5 * g5 + g1 * g1 + g3 * g3, 2 * (g5 * g6 + g1 * g2 + g3 * g4), g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3,
                              2 * (g6 * g7 - g1 * g2 - g3 * g4), g7 * g7 - g2 * g2 - g4 * g4};
    double s[4];
    ap3p_quartic(coeffs, s);
    double temp[3];
    cross3(k1, nl, temp);
    const double Ck1nl[9] = {k1[0], nl[0], temp[0], k1[1], nl[1], temp[1], k1[2], nl[2], temp[2]};
    const double Cb1k3tzT[9] = {b0[0], b0[1], b0[2], k3[0], k3[1], k3[2], tz[0], tz[1], tz[2]};
    const double sc = delta / k3b3;
    const double b3p[3] = {sc * b2[0], sc * b2[1], sc * b2[2]};
    // the head of OpenCV's insertion sort by the fourth point's error: the first pose whose error no later pose undercuts STRICTLY...
    // a stable insertion sort puts the earliest of the smallest errors first (a NaN neither moves nor lets anything pass it): replayed below
    double Rs[4][9], ts[4][3], errs[4];
    int nb = 0;
    for (int i = 0; i < 4; i++) {
        const double ctheta1p = s[i];
        if (!(fabs(ctheta1p) <= 1)) continue;
        double stheta1p = sqrt(1 - ctheta1p * ctheta1p);
        stheta1p = (k3b3 > 0) ? stheta1p : -stheta1p;
        double ctheta3 = g1 * ctheta1p + g2;
        double stheta3 = g3 * ctheta1p + g4;
        const double ntheta3 = stheta1p / ((g5 * ctheta1p + g6) * ctheta1p + g7);
        ctheta3 *= ntheta3;
        stheta3 *= ntheta3;
        const double C13[9] = {ctheta3, 0, -stheta3, stheta1p * stheta3, ctheta1p, stheta1p * ctheta3, ctheta1p * stheta3, -stheta1p, ctheta1p * ctheta3};
        double tmp[9], Rm[9];
        mult3(Ck1nl, C13, tmp);
        mult3(tmp, Cb1k3tzT, Rm);
        const double rp3[3] = {w3[0] * Rm[0] + w3[1] * Rm[3] + w3[2] * Rm[6], w3[0] * Rm[1] + w3[1] * Rm[4] + w3[2] * Rm[7], w3[0] * Rm[2] + w3[1] * Rm[5] + w3[2] * Rm[8]};
        double Rt[9], tt[3];
        for (int k = 0; k < 3; k++) tt[k] = stheta1p * b3p[k] - rp3[k];
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) Rt[3 * a + c] = Rm[3 * c + a];   // the pose is the transpose
        const double X3p = Rt[0] * w4[0] + Rt[1] * w4[1] + Rt[2] * w4[2] + tt[0];
        const double Y3p = Rt[3] * w4[0] + Rt[4] * w4[1] + Rt[5] * w4[2] + tt[1];
        const double Z3p = Rt[6] * w4[0] + Rt[7] * w4[1] + Rt[8] * w4[2] + tt[2];
        const double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        const double e = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
        // static slots (no dynamically indexed register arrays on the device)
#define APDS_AP3P_PUT(K)                                   \
    if (nb == K) {                                         \
        for (int q = 0; q < 9; q++) Rs[K][q] = Rt[q];      \
        for (int q = 0; q < 3; q++) ts[K][q] = tt[q];      \
        errs[K] = e;                                       \
    }
        APDS_AP3P_PUT(0)
        APDS_AP3P_PUT(1)
        APDS_AP3P_PUT(2)
        APDS_AP3P_PUT(3)
#undef APDS_AP3P_PUT
        nb++;
    }
    if (nb <= 0) return false;
    int order[4] = {0, 1, 2, 3};
    for (int i = 1; i < nb; i++)
        for (int j = i; j > 0 && errs[j - 1] > errs[j]; j--) {
            const double e = errs[j];
            errs[j] = errs[j - 1];
            errs[j - 1] = e;
            const int o = order[j];
            order[j] = order[j - 1];
            order[j - 1] = o;
        }
    const int h = order[0];
    for (int k = 0; k < 9; k++) Rbest[k] = h == 0 ? Rs[0][k] : (h == 1 ? Rs[1][k] : (h == 2 ? Rs[2][k] : Rs[3][k]));
    for (int k = 0; k < 3; k++) tbest[k] = h == 0 ? ts[0][k] : (h == 1 ? ts[1][k] : (h == 2 ? ts[2][k] : ts[3][k]));
    return true;
}

}  // namespace pnp
}  // namespace apds
