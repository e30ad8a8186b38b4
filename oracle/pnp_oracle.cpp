// oracle/pnp_oracle.cpp — solvePnPRansac (EPnP kernel) restated on the CPU. TEST INFRASTRUCTURE ONLY.
//
// Reference call site: homographier/src/homographier/mod.rs:320-369 (pnp_solver_ransac ->
// opencv::calib3d::solve_pnp_ransac(obj, img, K, zeros(4,1), rvec, tvec, false, iters, reproj, conf, inliers,
// method.unwrap_or(SOLVEPNP_EPNP)); note mod.rs:344 shadows dist_coeffs with zeros, so distortion is always zero).
// Arithmetic: OpenCV 4.x calib3d solvepnp.cpp (solvePnPRansac, PnPRansacCallback, solvePnPGeneric EPNP branch),
// epnp.cpp (control points, barycentric coordinates, M^T M null space, three beta approximations, Gauss-Newton,
// absolute orientation), ptsetreg.cpp (RANSAC registrator), calibration.cpp (Rodrigues, projectPoints),
// undistort (undistortPoints with k = 0), core lapack.cpp (one-sided Jacobi SVD, SVBkSb). None of it is in /root/reference.
//
// PARITY UNPINNED: the reference's only live test for this function is the "fewer than 4 points is an error" case
// (mod.rs:627-638), which tests/test_oracle_kat.py holds; its pnp_solver_works test is #[ignore]d and asserts no values.
// Stated deviations from OpenCV (all at the last-ulp level, chosen so that the CPU and the GPU evaluate identical
// IEEE operations): hypot(a,b) is sqrt(a*a+b*b); sin/cos/acos are the fixed polynomials below rather than libm.
// On minimal 5-point samples M^T M has a 2-dimensional null space whose basis OpenCV's SVD leaves to rounding noise,
// so individual hypotheses are not reproducible across SVD implementations even in principle; the inlier set on data
// with a clear inlier/outlier split and the final all-inlier pose (well-posed) are.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct RNG {
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---- deterministic elementary functions (shared specification with csrc/pnp.hip) -------------------------------------
void det_sincos(double a, double& s, double& c) {
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
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
}

// atan on [0, inf): reciprocal above 1, breakpoints k/8, odd Taylor series to v^17 on |v| <= 1/16
double det_atan(double t) {
    static const double tab[9] = {0.0, 0.12435499454676144, 0.24497866312686414, 0.35877067027057225, 0.4636476090008061,
                                  0.5585993153435624, 0.6435011087932844, 0.7188299996216245, 0.7853981633974483};
    const bool inv = t > 1.0;
    const double u = inv ? 1.0 / t : t;
    const int k = (int)(u * 8.0 + 0.5);
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
    const double r = tab[k] + (v + v * (v2 * p));
    return inv ? 1.5707963267948966 - r : r;
}

double det_acos(double c) { return 2.0 * det_atan(std::sqrt((1.0 - c) / (1.0 + c))); }

// ---- core lapack.cpp JacobiSVDImpl_<double>: one-sided (Hestenes) Jacobi on the ROWS of At (n rows of length m) -------
// On return: W = singular values (descending), rows of At = left singular vectors, rows of Vt = right singular vectors.
void jacobi_svd(double* At, int m, int n, double* W, double* Vt) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    const int max_iter = std::max(m, 30);
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const double t = At[i * m + k];
            sd += t * t;
        }
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * n + k] = 0;
        Vt[i * n + i] = 1;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double *Ai = At + i * m, *Aj = At + j * m;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = std::sqrt(p * p + beta * beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * Ai[k] + s * Aj[k];
                    const double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                double *Vi = Vt + i * n, *Vj = Vt + j * n;
                for (int k = 0; k < n; k++) {
                    const double t0 = c * Vi[k] + s * Vj[k];
                    const double t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0;
                    Vj[k] = t1;
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
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < m; k++) std::swap(At[i * m + k], At[j * m + k]);
            for (int k = 0; k < n; k++) std::swap(Vt[i * n + k], Vt[j * n + k]);
        }
    }
    RNG rng(0x12345678);
    for (int i = 0; i < n; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            // zero singular value: random +-1/m vector, orthogonalised against the previous left vectors, normalised
            const double val0 = 1. / m;
            for (int k = 0; k < m; k++) At[i * m + k] = (rng.next() & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; iter++) {
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < m; k++) sd += At[i * m + k] * At[j * m + k];
                    double asum = 0;
                    for (int k = 0; k < m; k++) {
                        const double t = At[i * m + k] - sd * At[j * m + k];
                        At[i * m + k] = t;
                        asum += std::fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < m; k++) At[i * m + k] *= asum;
                }
                sd = 0;
                for (int k = 0; k < m; k++) {
                    const double t = At[i * m + k];
                    sd += t * t;
                }
                sd = std::sqrt(sd);
            }
        }
        const double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

// cv::SVD::compute for m >= n: A (m x n, row-major) -> W[n], Ut (n x m: row i = i-th left vector), Vt (n x n)
void svd(const double* A, int m, int n, double* W, double* Ut, double* Vt) {
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++) Ut[i * m + k] = A[k * n + i];
    jacobi_svd(Ut, m, n, W, Vt);
}

// SVBkSb with one right-hand side: x = sum_i v_i (u_i . b) / w_i over w_i > 2 eps sum(w)
void svd_solve(const double* A, int m, int n, const double* b, double* x) {
    double W[12], Ut[12 * 12], Vt[12 * 12];
    svd(A, m, n, W, Ut, Vt);
    double threshold = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < n; j++) x[j] = 0;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++) s += Ut[i * m + j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] = x[j] + s * Vt[i * n + j];
    }
}

// cv::invert(DECOMP_SVD) for 3x3: inv[j][k] = sum_i v_i[j] * (u_i[k] / w_i)
void svd_invert3(const double* A, double* inv) {
    double W[3], Ut[9], Vt[9];
    svd(A, 3, 3, W, Ut, Vt);
    double threshold = (W[0] + W[1] + W[2]) * (DBL_EPSILON * 2);
    for (int j = 0; j < 9; j++) inv[j] = 0;
    for (int i = 0; i < 3; i++) {
        double wi = W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double buf[3];
        for (int k = 0; k < 3; k++) buf[k] = Ut[i * 3 + k] * wi;
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) inv[j * 3 + k] = inv[j * 3 + k] + Vt[i * 3 + j] * buf[k];
    }
}

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// ---- calibration.cpp Rodrigues --------------------------------------------------------------------------------------
void rodrigues_to_matrix(const double* rv, double* R) {
    const double theta = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        return;
    }
    double s, c;
    det_sincos(theta, s, c);
    const double c1 = 1. - c, itheta = 1. / theta;
    const double rx = rv[0] * itheta, ry = rv[1] * itheta, rz = rv[2] * itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int i = 0; i < 9; i++) R[i] = (c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i]) + s * r_x[i];
}

void rodrigues_to_vector(const double* Rin, double* rv) {
    double W[3], Ut[9], Vt[9], R[9];
    svd(Rin, 3, 3, W, Ut, Vt);
    for (int i = 0; i < 3; i++)      // R = U * Vt, U(i,k) = Ut[k][i]
        for (int j = 0; j < 3; j++) R[i * 3 + j] = Ut[0 * 3 + i] * Vt[0 * 3 + j] + Ut[1 * 3 + i] * Vt[1 * 3 + j] + Ut[2 * 3 + i] * Vt[2 * 3 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = det_acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            rx = std::sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
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

// ---- epnp.cpp -------------------------------------------------------------------------------------------------------
struct Camera { double fu, fv, uc, vc; };

struct Epnp {
    int n;
    Camera cam;
    std::vector<double> pws, us, alphas, pcs;
    double cws[4][3], ccs[4][3];

    void choose_control_points() {
        cws[0][0] = cws[0][1] = cws[0][2] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[0][j] /= n;
        // PW0^T PW0 (cvMulTransposed order 1: dst[i][j] = sum_k a[k][i] a[k][j], k ascending, upper triangle mirrored)
        double pw0tpw0[9] = {}, dc[3], uct[9], vt[9];
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < n; k++) s += (pws[3 * k + i] - cws[0][i]) * (pws[3 * k + j] - cws[0][j]);
                pw0tpw0[i * 3 + j] = pw0tpw0[j * 3 + i] = s;
            }
        svd(pw0tpw0, 3, 3, dc, uct, vt);
        for (int i = 1; i < 4; i++) {
            const double k = std::sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
        }
    }

    void compute_barycentric_coordinates() {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        svd_invert3(cc, ci);
        for (int i = 0; i < n; i++) {
            const double* pi = &pws[3 * i];
            double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }

    void compute_ccs(const double* betas, const double* ut) {
        for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
        for (int i = 0; i < 4; i++) {
            const double* v = ut + 12 * (11 - i);
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
        }
    }

    void compute_pcs() {
        for (int i = 0; i < n; i++) {
            const double* a = &alphas[4 * i];
            double* pc = &pcs[3 * i];
            for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
    }

    void solve_for_sign() {
        if (pcs[2] < 0.0) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
            for (int i = 0; i < n; i++) {
                pcs[3 * i] = -pcs[3 * i];
                pcs[3 * i + 1] = -pcs[3 * i + 1];
                pcs[3 * i + 2] = -pcs[3 * i + 2];
            }
        }
    }

    void estimate_R_and_t(double R[3][3], double t[3]) {
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
        double abt[9] = {}, abt_d[3], abt_ut[9], abt_vt[9];
        for (int i = 0; i < n; i++) {
            const double* pc = &pcs[3 * i];
            const double* pw = &pws[3 * i];
            for (int j = 0; j < 3; j++) {
                abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
                abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
                abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
            }
        }
        // cvSVD(ABt, D, U, V, MODIFY_A): U and V NOT transposed; R[i][j] = dot(row i of U, row j of V)
        svd(abt, 3, 3, abt_d, abt_ut, abt_vt);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R[i][j] = abt_ut[0 * 3 + i] * abt_vt[0 * 3 + j] + abt_ut[1 * 3 + i] * abt_vt[1 * 3 + j] + abt_ut[2 * 3 + i] * abt_vt[2 * 3 + j];
        const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] - R[0][2] * R[1][1] * R[2][0] -
                           R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
        if (det < 0) {
            R[2][0] = -R[2][0];
            R[2][1] = -R[2][1];
            R[2][2] = -R[2][2];
        }
        t[0] = pc0[0] - dot3(R[0], pw0);
        t[1] = pc0[1] - dot3(R[1], pw0);
        t[2] = pc0[2] - dot3(R[2], pw0);
    }

    double reprojection_error(const double R[3][3], const double t[3]) {
        double sum2 = 0.0;
        for (int i = 0; i < n; i++) {
            const double* pw = &pws[3 * i];
            const double Xc = dot3(R[0], pw) + t[0];
            const double Yc = dot3(R[1], pw) + t[1];
            const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
            const double ue = cam.uc + cam.fu * Xc * inv_Zc;
            const double ve = cam.vc + cam.fv * Yc * inv_Zc;
            const double u = us[2 * i], v = us[2 * i + 1];
            sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n;
    }

    double compute_R_and_t(const double* ut, const double* betas, double R[3][3], double t[3]) {
        compute_ccs(betas, ut);
        compute_pcs();
        solve_for_sign();
        estimate_R_and_t(R, t);
        return reprojection_error(R, t);
    }

    static void compute_L_6x10(const double* ut, double* l) {
        const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
                dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
                dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
                b++;
                if (b > 3) {
                    a++;
                    b = a + 1;
                }
            }
        }
        for (int i = 0; i < 6; i++) {
            double* row = l + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
    }

    void compute_rho(double* rho) {
        rho[0] = dist2(cws[0], cws[1]);
        rho[1] = dist2(cws[0], cws[2]);
        rho[2] = dist2(cws[0], cws[3]);
        rho[3] = dist2(cws[1], cws[2]);
        rho[4] = dist2(cws[1], cws[3]);
        rho[5] = dist2(cws[2], cws[3]);
    }

    static void find_betas_approx_1(const double* L, const double* rho, double* betas) {
        double l_6x4[24], b4[4];
        for (int i = 0; i < 6; i++) {
            l_6x4[4 * i] = L[10 * i];
            l_6x4[4 * i + 1] = L[10 * i + 1];
            l_6x4[4 * i + 2] = L[10 * i + 3];
            l_6x4[4 * i + 3] = L[10 * i + 6];
        }
        svd_solve(l_6x4, 6, 4, rho, b4);
        if (b4[0] < 0) {
            betas[0] = std::sqrt(-b4[0]);
            betas[1] = -b4[1] / betas[0];
            betas[2] = -b4[2] / betas[0];
            betas[3] = -b4[3] / betas[0];
        } else {
            betas[0] = std::sqrt(b4[0]);
            betas[1] = b4[1] / betas[0];
            betas[2] = b4[2] / betas[0];
            betas[3] = b4[3] / betas[0];
        }
    }

    static void find_betas_approx_2(const double* L, const double* rho, double* betas) {
        double l_6x3[18], b3[3];
        for (int i = 0; i < 6; i++) {
            l_6x3[3 * i] = L[10 * i];
            l_6x3[3 * i + 1] = L[10 * i + 1];
            l_6x3[3 * i + 2] = L[10 * i + 2];
        }
        svd_solve(l_6x3, 6, 3, rho, b3);
        if (b3[0] < 0) {
            betas[0] = std::sqrt(-b3[0]);
            betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0;
        } else {
            betas[0] = std::sqrt(b3[0]);
            betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0;
        }
        if (b3[1] < 0) betas[0] = -betas[0];
        betas[2] = 0.0;
        betas[3] = 0.0;
    }

    static void find_betas_approx_3(const double* L, const double* rho, double* betas) {
        double l_6x5[30], b5[5];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 5; j++) l_6x5[5 * i + j] = L[10 * i + j];
        svd_solve(l_6x5, 6, 5, rho, b5);
        if (b5[0] < 0) {
            betas[0] = std::sqrt(-b5[0]);
            betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0;
        } else {
            betas[0] = std::sqrt(b5[0]);
            betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0;
        }
        if (b5[1] < 0) betas[0] = -betas[0];
        betas[2] = b5[3] / betas[0];
        betas[3] = 0.0;
    }

    static void compute_A_and_b_gauss_newton(const double* L, const double* rho, const double* betas, double* A, double* b) {
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
    }

    // Householder QR exactly as epnp.cpp writes it, including its pivot scan that stops one row short and the silent
    // return (X left as it was) when a column is all zero
    static void qr_solve(double* pA, double* pb, double* pX) {
        const int nr = 6, nc = 4;
        double A1[6], A2[6];
        double* ppAkk = pA;
        for (int k = 0; k < nc; k++) {
            double* ppAik1 = ppAkk;
            double eta = std::fabs(*ppAik1);
            for (int i = k + 1; i < nr; i++) {
                const double elt = std::fabs(*ppAik1);
                if (eta < elt) eta = elt;
                ppAik1 += nc;
            }
            if (eta == 0) {
                A1[k] = A2[k] = 0.0;
                return;
            }
            double* ppAik2 = ppAkk;
            double sum2 = 0.0;
            const double inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) {
                *ppAik2 *= inv_eta;
                sum2 += *ppAik2 * *ppAik2;
                ppAik2 += nc;
            }
            double sigma = std::sqrt(sum2);
            if (*ppAkk < 0) sigma = -sigma;
            *ppAkk += sigma;
            A1[k] = sigma * *ppAkk;
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double* ppAik = ppAkk;
                double sum = 0;
                for (int i = k; i < nr; i++) {
                    sum += *ppAik * ppAik[j - k];
                    ppAik += nc;
                }
                const double tau = sum / A1[k];
                ppAik = ppAkk;
                for (int i = k; i < nr; i++) {
                    ppAik[j - k] -= tau * *ppAik;
                    ppAik += nc;
                }
            }
            ppAkk += nc + 1;
        }
        double* ppAjj = pA;
        for (int j = 0; j < nc; j++) {
            double* ppAij = ppAjj;
            double tau = 0;
            for (int i = j; i < nr; i++) {
                tau += *ppAij * pb[i];
                ppAij += nc;
            }
            tau /= A1[j];
            ppAij = ppAjj;
            for (int i = j; i < nr; i++) {
                pb[i] -= tau * *ppAij;
                ppAij += nc;
            }
            ppAjj += nc + 1;
        }
        pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
        for (int i = nc - 2; i >= 0; i--) {
            double* ppAij = pA + i * nc + (i + 1);
            double sum = 0;
            for (int j = i + 1; j < nc; j++) {
                sum += *ppAij * pX[j];
                ppAij++;
            }
            pX[i] = (pb[i] - sum) / A2[i];
        }
    }

    static void gauss_newton(const double* L, const double* rho, double* betas) {
        double a[24] = {}, b[6] = {}, x[4] = {};
        for (int k = 0; k < 5; k++) {
            compute_A_and_b_gauss_newton(L, rho, betas, a, b);
            qr_solve(a, b, x);
            for (int i = 0; i < 4; i++) betas[i] += x[i];
        }
    }

    void compute_pose(double Rout[9], double tout[3]) {
        choose_control_points();
        compute_barycentric_coordinates();
        // M (2n x 12) and M^T M (cvMulTransposed order 1)
        std::vector<double> M((size_t)2 * n * 12);
        for (int i = 0; i < n; i++) {
            double* M1 = &M[(size_t)2 * i * 12];
            double* M2 = M1 + 12;
            const double* as = &alphas[4 * i];
            const double u = us[2 * i], v = us[2 * i + 1];
            for (int j = 0; j < 4; j++) {
                M1[3 * j] = as[j] * cam.fu;
                M1[3 * j + 1] = 0.0;
                M1[3 * j + 2] = as[j] * (cam.uc - u);
                M2[3 * j] = 0.0;
                M2[3 * j + 1] = as[j] * cam.fv;
                M2[3 * j + 2] = as[j] * (cam.vc - v);
            }
        }
        double mtm[144], d[12], ut[144], vt[144];
        for (int i = 0; i < 12; i++)
            for (int j = i; j < 12; j++) {
                double s = 0;
                for (int k = 0; k < 2 * n; k++) s += M[(size_t)k * 12 + i] * M[(size_t)k * 12 + j];
                mtm[i * 12 + j] = mtm[j * 12 + i] = s;
            }
        svd(mtm, 12, 12, d, ut, vt);
        double l_6x10[60], rho[6];
        compute_L_6x10(ut, l_6x10);
        compute_rho(rho);
        double Betas[4][4] = {}, rep_errors[4] = {};
        double Rs[4][3][3] = {}, ts[4][3] = {};
        find_betas_approx_1(l_6x10, rho, Betas[1]);
        gauss_newton(l_6x10, rho, Betas[1]);
        rep_errors[1] = compute_R_and_t(ut, Betas[1], Rs[1], ts[1]);
        find_betas_approx_2(l_6x10, rho, Betas[2]);
        gauss_newton(l_6x10, rho, Betas[2]);
        rep_errors[2] = compute_R_and_t(ut, Betas[2], Rs[2], ts[2]);
        find_betas_approx_3(l_6x10, rho, Betas[3]);
        gauss_newton(l_6x10, rho, Betas[3]);
        rep_errors[3] = compute_R_and_t(ut, Betas[3], Rs[3], ts[3]);
        int N = 1;
        if (rep_errors[2] < rep_errors[1]) N = 2;
        if (rep_errors[3] < rep_errors[N]) N = 3;
        for (int i = 0; i < 3; i++) {
            tout[i] = ts[N][i];
            for (int j = 0; j < 3; j++) Rout[3 * i + j] = Rs[N][i][j];
        }
    }
};

// solvePnPGeneric, EPNP branch: undistortPoints(k = 0) -> epnp -> Rodrigues. `as_float` = the points are CV_32F (the
// RANSAC minimal sets), which makes undistortPoints store its output as float.
template <typename T>
void solve_pnp_epnp(const T* obj, const T* img, int n, const double* K, double* rvec, double* tvec) {
    Epnp e;
    e.n = n;
    e.cam = {K[0], K[4], K[2], K[5]};
    e.pws.resize(3 * n);
    e.us.resize(2 * n);
    e.alphas.resize(4 * n);
    e.pcs.resize(3 * n);
    const double ifx = 1. / K[0], ify = 1. / K[4];
    for (int i = 0; i < n; i++) {
        e.pws[3 * i] = obj[3 * i];
        e.pws[3 * i + 1] = obj[3 * i + 1];
        e.pws[3 * i + 2] = obj[3 * i + 2];
        const T xn = (T)(((double)img[2 * i] - K[2]) * ifx);
        const T yn = (T)(((double)img[2 * i + 1] - K[5]) * ify);
        e.us[2 * i] = xn * e.cam.fu + e.cam.uc;
        e.us[2 * i + 1] = yn * e.cam.fv + e.cam.vc;
    }
    double R[9];
    e.compute_pose(R, tvec);
    rodrigues_to_vector(R, rvec);
}

// PnPRansacCallback::computeError: projectPoints (k = 0) to float, squared distance in float
void pnp_errors(const float* obj, const float* img, int n, const double* K, const double* rvec, const double* tvec, float* err) {
    double R[9];
    rodrigues_to_matrix(rvec, R);
    for (int i = 0; i < n; i++) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        const float px = (float)(x * K[0] + K[2]), py = (float)(y * K[4] + K[5]);
        const float dx = img[2 * i] - px, dy = img[2 * i + 1] - py;
        err[i] = dx * dx + dy * dy;
    }
}

// ---- SOLVEPNP_ITERATIVE: the final refinement of solvePnPRansac(..., flags = SOLVEPNP_ITERATIVE) ------------------------------------
// Reference call site: homographier/src/homographier/mod.rs:327,359-360 (`method: Option<SolvePnPMethod>` handed to solve_pnp_ransac).
// calib3d/solvepnp.cpp as recalled (OpenCV 4.8): the RANSAC kernel stays EPnP on 5 points, and the final solvePnP over the inliers runs
// with the CALLER's useExtrinsicGuess - false in the reference (mod.rs:354) - so cvFindExtrinsicCameraParams2 (calibration.cpp) first
// builds its own starting pose (initial_pose_no_guess below: homography for planar sets, DLT otherwise) and then refines it:
// CvLevMarq(6 parameters, 2 n residuals, max 20 iterations, eps FLT_EPSILON, completeSymmFlag) around cvProjectPoints2 with its
// analytic Jacobian (Rodrigues' dR/dr, then d(proj)/d(r, t)). Zero distortion (mod.rs:344). PARITY UNPINNED; what the tests hold it
// to is first-order optimality and an independent numpy optimiser (tests/test_external_anchors.py).
void rodrigues_with_jacobian(const double* rv, double* R, double* J /* 3 x 9: row i = dR/dr_i */) {
    const double theta = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    static const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        for (int i = 0; i < 27; i++) J[i] = d_r_x[i];
        return;
    }
    double s, c;
    det_sincos(theta, s, c);
    const double c1 = 1. - c, itheta = 1. / theta;
    const double r[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
    const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
    const double r_x[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = (c * I[k] + c1 * rrt[k]) + s * r_x[k];
    const double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0, 0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0,
                             0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
    for (int i = 0; i < 3; i++) {
        const double ri = r[i];
        const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri, a4 = s * itheta;
        for (int k = 0; k < 9; k++) J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x[i * 9 + k];
    }
}

// cvProjectPoints2 without distortion: err[2 i .. ] = projection - measurement; Jm (2 n x 6, row major: d/dr then d/dt) if wanted
void project_residuals(const double* obj, const double* img, int n, const Camera& cam, const double* param, double* err, double* Jm) {
    double R[9], dRdr[27];
    rodrigues_with_jacobian(param, R, dRdr);
    const double* t = param + 3;
    for (int i = 0; i < n; i++) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        err[2 * i] = (x * cam.fu + cam.uc) - img[2 * i];
        err[2 * i + 1] = (y * cam.fv + cam.vc) - img[2 * i + 1];
        if (!Jm) continue;
        double* jr0 = Jm + (size_t)(2 * i) * 6;
        double* jr1 = jr0 + 6;
        const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
        for (int j = 0; j < 3; j++) {
            jr0[3 + j] = cam.fu * dxdt[j];
            jr1[3 + j] = cam.fv * dydt[j];
        }
        for (int j = 0; j < 3; j++) {
            const double* d = dRdr + 9 * j;
            const double dx0 = X * d[0] + Y * d[1] + Z * d[2], dy0 = X * d[3] + Y * d[4] + Z * d[5], dz0 = X * d[6] + Y * d[7] + Z * d[8];
            jr0[j] = cam.fu * (z * (dx0 - x * dz0));
            jr1[j] = cam.fv * (z * (dy0 - y * dz0));
        }
    }
}

double l2_norm(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return std::sqrt(s);
}

// cvFindExtrinsicCameraParams2's starting pose when there is NO extrinsic guess - which is how the reference reaches it: mod.rs:354 passes
// use_extrinsic_guess = false and solvePnPRansac hands that flag on to its final solvePnP over the inliers (rounds 2 - 3 started
// the refinement from the best RANSAC model instead). calibration.cpp as recalled: image points normalised by the intrinsics; the object
// points' 3 x 3 scatter decides planar / non-planar (w[2] / w[1] < 1e-3); planar: rotate the points into their plane, homography plane ->
// normalised image (cv::findHomography, method 0, on float copies), columns h1, h2 scaled to unit length, t = h3 * 2 / (|h1| + |h2|), R made
// orthonormal by Rodrigues there and back; non-planar: the 12 x 12 normal matrix of the DLT rows, its last singular vector as the 3 x 4
// projection, R = U V^T of its left block, t rescaled by |R| / |block|. Returns 0 where OpenCV throws "DLT algorithm needs at least 6
// points" (solvePnPRansac catches exactly that, for five inliers, and keeps the RANSAC model).
int initial_pose_no_guess(const double* obj, const double* img, int n, const Camera& cam, double* param) {
    std::vector<double> mn(2 * (size_t)n);
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
    for (int i = 0; i < n; i++) {   // cvUndistortPoints with zero distortion
        mn[2 * (size_t)i] = (img[2 * i] - cam.uc) * ifx;
        mn[2 * (size_t)i + 1] = (img[2 * i + 1] - cam.vc) * ify;
    }
    double Mc[3] = {0, 0, 0}, MM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) Mc[k] += obj[3 * i + k];
    for (int k = 0; k < 3; k++) Mc[k] /= n;
    for (int i = 0; i < n; i++) {
        const double d[3] = {obj[3 * i] - Mc[0], obj[3 * i + 1] - Mc[1], obj[3 * i + 2] - Mc[2]};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += d[a] * d[b];
    }
    double W[3], Ut[9], V[9];   // V holds V^T (CV_SVD_V_T)
    svd(MM, 3, 3, W, Ut, V);
    double R[9], t[3];
    if (W[2] / W[1] < 1e-3) {   // a planar structure: all M's lie in the same plane
        if (V[2] * V[2] + V[5] * V[5] < 1e-10)
            for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1. : 0.;
        const double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
        if (det < 0)
            for (int i = 0; i < 9; i++) V[i] = -V[i];
        double T[3];
        for (int a = 0; a < 3; a++) T[a] = -(V[a * 3] * Mc[0] + V[a * 3 + 1] * Mc[1] + V[a * 3 + 2] * Mc[2]);
        std::vector<float> src(2 * (size_t)n), dst(2 * (size_t)n);   // findHomography converts its points to CV_32F
        for (int i = 0; i < n; i++) {
            const double* M = obj + 3 * i;
            src[2 * (size_t)i] = (float)(V[0] * M[0] + V[1] * M[1] + V[2] * M[2] + T[0]);
            src[2 * (size_t)i + 1] = (float)(V[3] * M[0] + V[4] * M[1] + V[5] * M[2] + T[1]);
            dst[2 * (size_t)i] = (float)mn[2 * (size_t)i];
            dst[2 * (size_t)i + 1] = (float)mn[2 * (size_t)i + 1];
        }
        double h[9];
        const int found = oracle_find_homography(src.data(), dst.data(), n, 0, 3.0, 2000, 0.995, h, nullptr);
        bool finite = found == 1;
        for (int i = 0; i < 9 && finite; i++) finite = std::isfinite(h[i]);
        if (finite) {
            const double h1n = std::sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = std::sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            const double s1 = 1. / std::max(h1n, DBL_EPSILON), s2 = 1. / std::max(h2n, DBL_EPSILON), s3 = 2. / std::max(h1n + h2n, DBL_EPSILON);
            for (int r = 0; r < 3; r++) {
                t[r] = h[r * 3 + 2] * s3;
                h[r * 3] *= s1;
                h[r * 3 + 1] *= s2;
            }
            h[2] = h[3] * h[7] - h[6] * h[4];   // h3 = h1 x h2
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double rv[3], Rh[9];
            rodrigues_to_vector(h, rv);
            rodrigues_to_matrix(rv, Rh);
            for (int a = 0; a < 3; a++) t[a] = (Rh[a * 3] * T[0] + Rh[a * 3 + 1] * T[1] + Rh[a * 3 + 2] * T[2]) + t[a];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) R[a * 3 + b] = Rh[a * 3] * V[b] + Rh[a * 3 + 1] * V[3 + b] + Rh[a * 3 + 2] * V[6 + b];
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            t[0] = t[1] = t[2] = 0;
        }
    } else {   // non-planar structure: DLT
        if (n < 6) return 0;
        double LL[144];
        for (int i = 0; i < 144; i++) LL[i] = 0;
        for (int i = 0; i < n; i++) {
            const double* M = obj + 3 * i;
            const double x = -mn[2 * (size_t)i], y = -mn[2 * (size_t)i + 1];
            const double r1[12] = {M[0], M[1], M[2], 1., 0., 0., 0., 0., x * M[0], x * M[1], x * M[2], x};
            const double r2[12] = {0., 0., 0., 0., M[0], M[1], M[2], 1., y * M[0], y * M[1], y * M[2], y};
            for (int a = 0; a < 12; a++)
                for (int b = 0; b < 12; b++) LL[a * 12 + b] += r1[a] * r1[b] + r2[a] * r2[b];
        }
        double LW[12], LUt[144], LV[144];
        svd(LL, 12, 12, LW, LUt, LV);
        double P[12];
        for (int i = 0; i < 12; i++) P[i] = LV[11 * 12 + i];   // the 3 x 4 projection, row major
        const double det = P[0] * (P[5] * P[10] - P[6] * P[9]) - P[1] * (P[4] * P[10] - P[6] * P[8]) + P[2] * (P[4] * P[9] - P[5] * P[8]);
        if (det < 0)
            for (int i = 0; i < 12; i++) P[i] = -P[i];
        const double RR[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
        double sc = 0;
        for (int i = 0; i < 9; i++) sc += RR[i] * RR[i];
        sc = std::sqrt(sc);
        double w3[3], U3t[9], V3t[9];
        svd(RR, 3, 3, w3, U3t, V3t);
        double rn = 0;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                R[a * 3 + b] = U3t[a] * V3t[b] + U3t[3 + a] * V3t[3 + b] + U3t[6 + a] * V3t[6 + b];   // U V^T
                rn += R[a * 3 + b] * R[a * 3 + b];
            }
        const double k = std::sqrt(rn) / sc;
        t[0] = P[3] * k;
        t[1] = P[7] * k;
        t[2] = P[11] * k;
    }
    rodrigues_to_vector(R, param);
    param[3] = t[0];
    param[4] = t[1];
    param[5] = t[2];
    return 1;
}

// CvLevMarq::update()'s state machine, unrolled into straight code; param: in = the initial pose (rvec, tvec), out = the refined one
void refine_pose_lm(const double* obj, const double* img, int n, const Camera& cam, double* param) {
    const int max_iter = 20, m = 2 * n;
    std::vector<double> J((size_t)m * 6), err(m);
    double JtJ[36], JtErr[6], prev[6], A[36], step[6];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX;
    auto lm_step = [&]() {
        const double lambda = std::exp(lambdaLg10 * std::log(10.));
        std::memcpy(A, JtJ, sizeof(A));
        for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
        svd_solve(A, 6, 6, JtErr, step);
        for (int i = 0; i < 6; i++) param[i] = prev[i] - step[i];
    };
    project_residuals(obj, img, n, cam, param, err.data(), J.data());        // STARTED -> CALC_J
    for (;;) {
        // CALC_J
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) {
                double s = 0;
                for (int k = 0; k < m; k++) s += J[(size_t)k * 6 + a] * J[(size_t)k * 6 + b];
                JtJ[a * 6 + b] = JtJ[b * 6 + a] = s;
            }
        for (int a = 0; a < 6; a++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += J[(size_t)k * 6 + a] * err[k];
            JtErr[a] = s;
        }
        std::memcpy(prev, param, sizeof(prev));
        lm_step();
        if (iters == 0) prevErrNorm = l2_norm(err.data(), m);
        project_residuals(obj, img, n, cam, param, err.data(), nullptr);      // -> CHECK_ERR
        double errNorm;
        for (;;) {
            errNorm = l2_norm(err.data(), m);
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) {
                lm_step();
                project_residuals(obj, img, n, cam, param, err.data(), nullptr);
                continue;
            }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double diff[6];
        for (int i = 0; i < 6; i++) diff[i] = param[i] - prev[i];
        if (++iters >= max_iter || l2_norm(diff, 6) / l2_norm(prev, 6) < FLT_EPSILON) return;   // DONE
        prevErrNorm = errNorm;
        project_residuals(obj, img, n, cam, param, err.data(), J.data());     // -> CALC_J
    }
}

// ---- p3p.cpp (Gao, Hou, Tang, Cheng 2003) + polynom_solver.cpp --------------------------------------------------------
// cube root with a fixed evaluation order (OpenCV calls pow(x, 1/3.)): bit-level first guess + 6 Newton steps, x > 0
double det_cbrt(double x) {
    uint64_t i;
    std::memcpy(&i, &x, 8);
    i = i / 3 + 0x2A9F7893782DA1CEull;
    double y;
    std::memcpy(&y, &i, 8);
    for (int k = 0; k < 6; k++) y = y - (y * y * y - x) / (3.0 * (y * y));
    return y;
}
double det_cos(double a) {   // a in [0, 2 pi]
    double s, c;
    det_sincos(a, s, c);
    return c;
}

int solve_deg2(double a, double b, double c, double& x1, double& x2) {
    const double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    const double inv_2a = 0.5 / a;
    if (delta == 0) {
        x1 = -b * inv_2a;
        x2 = x1;
        return 1;
    }
    const double sqrt_delta = std::sqrt(delta);
    x1 = (-b + sqrt_delta) * inv_2a;
    x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

int solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
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
        x0 = (2 * R > 0 ? det_cbrt(2 * R) : std::nan("")) - b_a_3;   // pow(negative, 1/3.) is NaN
        return 1;
    }
    if (D <= 0) {
        const double theta = det_acos(R / std::sqrt(-Q3));
        const double sqrt_Q = std::sqrt(-Q);
        x0 = 2 * sqrt_Q * det_cos(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * det_cos((theta + 2 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * det_cos((theta + 4 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        return 3;
    }
    const double AD = det_cbrt(std::fabs(R) + std::sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

int solve_deg4(double a, double b, double c, double d, double e, double& x0, double& x1, double& x2, double& x3) {
    if (a == 0) {
        x3 = 0;
        return solve_deg3(b, c, d, e, x0, x1, x2);
    }
    const double inv_a = 1. / a;
    b *= inv_a;
    c *= inv_a;
    d *= inv_a;
    e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    const int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
    if (n == 0) return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double R = std::sqrt(R2);
    const double inv_R = 1. / R;
    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            const double sqrt_temp = std::sqrt(temp);
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
        const double D = std::sqrt(D2);
        nb_real_roots = 2;
        const double D_2 = 0.5 * D;
        x0 = R_2 + D_2 - b_4;
        x1 = x0 - D;
    }
    if (E2 >= 0) {
        const double E = std::sqrt(E2);
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

bool jacobi_4x4(double* A, double* D, double* U) {
    double B[4], Z[4];
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
    B[0] = A[0];
    B[1] = A[5];
    B[2] = A[10];
    B[3] = A[15];
    std::memcpy(D, B, sizeof B);
    std::memset(Z, 0, sizeof Z);
    for (int iter = 0; iter < 50; iter++) {
        const double sum = std::fabs(A[1]) + std::fabs(A[2]) + std::fabs(A[3]) + std::fabs(A[6]) + std::fabs(A[7]) + std::fabs(A[11]);
        if (sum == 0.0) return true;
        const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            double* pAij = A + 5 * i + 1;
            for (int j = i + 1; j < 4; j++) {
                const double Aij = *pAij;
                const double eps_machine = 100.0 * std::fabs(Aij);
                if (iter > 3 && std::fabs(D[i]) + eps_machine == std::fabs(D[i]) && std::fabs(D[j]) + eps_machine == std::fabs(D[j])) {
                    *pAij = 0.0;
                } else if (std::fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (std::fabs(hh) + eps_machine == std::fabs(hh)) t = Aij / hh;
                    else {
                        const double theta = 0.5 * hh / Aij;
                        t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh;
                    Z[j] += hh;
                    D[i] -= hh;
                    D[j] += hh;
                    *pAij = 0.0;
                    const double c = 1.0 / std::sqrt(1 + t * t);
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
                pAij++;
            }
        }
        for (int i = 0; i < 4; i++) B[i] += Z[i];
        std::memcpy(D, B, sizeof B);
        std::memset(Z, 0, sizeof Z);
    }
    return false;
}

bool p3p_align(double M_end[3][3], const double* P0, const double* P1, const double* P2, double R[3][3], double T[3]) {
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (P0[0] + P1[0] + P2[0]) / 3;
    C_start[1] = (P0[1] + P1[1] + P2[1]) / 3;
    C_start[2] = (P0[2] + P1[2] + P2[2]) / 3;
    double s[9];
    for (int j = 0; j < 3; j++) {
        s[0 * 3 + j] = (P0[0] * M_end[0][j] + P1[0] * M_end[1][j] + P2[0] * M_end[2][j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (P0[1] * M_end[0][j] + P1[1] * M_end[1][j] + P2[1] * M_end[2][j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (P0[2] * M_end[0][j] + P1[2] * M_end[1][j] + P2[2] * M_end[2][j]) / 3 - C_end[j] * C_start[2];
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
    jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[i_ev];
    for (int i = 1; i < 4; i++)
        if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    const double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    const double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    const double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3];
    const double q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32;
    R[0][1] = 2. * (q1_2 - q0_3);
    R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3);
    R[1][1] = q02 + q22 - q12 - q32;
    R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2);
    R[2][1] = 2. * (q2_3 + q0_1);
    R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return true;
}

int p3p_solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3]) {
    const double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    const double inv_d22 = 1. / (distances[2] * distances[2]);
    const double a = inv_d22 * (distances[0] * distances[0]);
    const double b = inv_d22 * (distances[1] * distances[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;   // reality condition (the four points are coplanar with the centre)
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
    double real_roots[4];
    const int n = solve_deg4(A, B, C, D, E, real_roots[0], real_roots[1], real_roots[2], real_roots[3]);
    if (n == 0) return 0;
    int nb_solutions = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        const double x = real_roots[i];
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
        const double Z = distances[2] / std::sqrt(v);
        lengths[nb_solutions][0] = x * Z;
        lengths[nb_solutions][1] = y * Z;
        lengths[nb_solutions][2] = Z;
        nb_solutions++;
    }
    return nb_solutions;
}

// p3p::solve for four correspondences (pixel coordinates mu/mv, object X/Y/Z): up to four poses from the first three points,
// sorted by the squared normalised reprojection error of the fourth. Returns the number of poses.
int p3p_solve(const Camera& cam, const double* mu_in, const double* mv_in, const double P[4][3], double R[4][3][3], double t[4][3]) {
    const double inv_fx = 1. / cam.fu, inv_fy = 1. / cam.fv, cx_fx = cam.uc / cam.fu, cy_fy = cam.vc / cam.fv;
    double mu[4], mv[4], mk[3];
    for (int i = 0; i < 3; i++) {
        mu[i] = inv_fx * mu_in[i] - cx_fx;
        mv[i] = inv_fy * mv_in[i] - cy_fy;
        const double norm = std::sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm;
        mu[i] *= mk[i];
        mv[i] *= mk[i];
    }
    mu[3] = inv_fx * mu_in[3] - cx_fx;
    mv[3] = inv_fy * mv_in[3] - cy_fy;
    double distances[3];
    distances[0] = std::sqrt((P[1][0] - P[2][0]) * (P[1][0] - P[2][0]) + (P[1][1] - P[2][1]) * (P[1][1] - P[2][1]) + (P[1][2] - P[2][2]) * (P[1][2] - P[2][2]));
    distances[1] = std::sqrt((P[0][0] - P[2][0]) * (P[0][0] - P[2][0]) + (P[0][1] - P[2][1]) * (P[0][1] - P[2][1]) + (P[0][2] - P[2][2]) * (P[0][2] - P[2][2]));
    distances[2] = std::sqrt((P[0][0] - P[1][0]) * (P[0][0] - P[1][0]) + (P[0][1] - P[1][1]) * (P[0][1] - P[1][1]) + (P[0][2] - P[1][2]) * (P[0][2] - P[1][2]));
    double cosines[3];
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    double lengths[4][3] = {};
    const int n = p3p_solve_for_lengths(lengths, distances, cosines);
    int nb = 0;
    double reproj_errors[4];
    for (int i = 0; i < n; i++) {
        double M_orig[3][3];
        for (int j = 0; j < 3; j++) {
            M_orig[j][0] = lengths[i][j] * mu[j];
            M_orig[j][1] = lengths[i][j] * mv[j];
            M_orig[j][2] = lengths[i][j] * mk[j];
        }
        if (!p3p_align(M_orig, P[0], P[1], P[2], R[nb], t[nb])) continue;
        const double X3p = R[nb][0][0] * P[3][0] + R[nb][0][1] * P[3][1] + R[nb][0][2] * P[3][2] + t[nb][0];
        const double Y3p = R[nb][1][0] * P[3][0] + R[nb][1][1] * P[3][1] + R[nb][1][2] * P[3][2] + t[nb][1];
        const double Z3p = R[nb][2][0] * P[3][0] + R[nb][2][1] * P[3][1] + R[nb][2][2] * P[3][2] + t[nb][2];
        const double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        reproj_errors[nb] = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
        nb++;
    }
    for (int i = 1; i < nb; i++)
        for (int j = i; j > 0 && reproj_errors[j - 1] > reproj_errors[j]; j--) {
            std::swap(reproj_errors[j], reproj_errors[j - 1]);
            for (int k = 0; k < 9; k++) std::swap((&R[j][0][0])[k], (&R[j - 1][0][0])[k]);
            for (int k = 0; k < 3; k++) std::swap(t[j][k], t[j - 1][k]);
        }
    return nb;
}

// solvePnP(4 points, SOLVEPNP_P3P): undistortPoints(k = 0, P = cameraMatrix) -> pixel coordinates (stored as T) -> best pose
template <typename T>
bool solve_pnp_p3p(const T* obj, const T* img, const double* K, double* rvec, double* tvec) {
    const Camera cam{K[0], K[4], K[2], K[5]};
    const double ifx = 1. / K[0], ify = 1. / K[4];
    double mu[4], mv[4], P[4][3];
    for (int i = 0; i < 4; i++) {
        const double xn = ((double)img[2 * i] - K[2]) * ifx, yn = ((double)img[2 * i + 1] - K[5]) * ify;
        mu[i] = (T)(K[0] * xn + K[2]);
        mv[i] = (T)(K[4] * yn + K[5]);
        for (int c = 0; c < 3; c++) P[i][c] = obj[3 * i + c];
    }
    double R[4][3][3], t[4][3];
    if (p3p_solve(cam, mu, mv, P, R, t) <= 0) return false;
    rodrigues_to_vector(&R[0][0][0], rvec);
    for (int i = 0; i < 3; i++) tvec[i] = t[0][i];
    return true;
}

// ---- ap3p.cpp (Ke, Roumeliotis 2017: "An efficient algebraic solution to the perspective-three-point problem") ---------------------------
// SOLVEPNP_AP3P: the kernel solvePnPRansac switches to for flags == SOLVEPNP_AP3P (4 points per sample, the fourth ranks the poses; the
// final pose over the inliers is EPnP, as for P3P). Restated from memory of calib3d/src/ap3p.cpp: a quartic in cos(theta1') from the
// seven g coefficients, its four roots by Ferrari's closed form in COMPLEX arithmetic (the real parts of all four are kept, as upstream
// does), two Newton polishing steps, then rotation and translation per root with |cos| <= 1. Elementary functions are the fixed-order
// ones of this file, so the HIP path (csrc/pnp_core.h) agrees bit for bit. PARITY UNPINNED (oracle.h); the anchors of
// tests/test_external_anchors.py hold it to the construction of the data.
struct Cx {
    double re, im;
};
Cx cx_sqrt(Cx z) {   // principal square root (libstdc++'s formula)
    if (z.re == 0.0) {
        const double t = std::sqrt(std::fabs(z.im) / 2);
        return Cx{t, z.im < 0.0 ? -t : t};
    }
    const double m = std::sqrt(z.re * z.re + z.im * z.im);
    const double t = std::sqrt(2 * (m + std::fabs(z.re)));
    const double u = t / 2;
    return z.re > 0.0 ? Cx{u, z.im / t} : Cx{std::fabs(z.im) / t, z.im < 0.0 ? -u : u};
}
double det_atan2(double y, double x) {   // for y != 0 or x != 0
    const double ax = std::fabs(x), ay = std::fabs(y);
    double a = ax >= ay ? det_atan(ay / ax) : 1.5707963267948966 - det_atan(ax / ay);
    if (x < 0) a = 3.141592653589793 - a;
    return y < 0 ? -a : a;
}
Cx cx_cbrt(Cx z) {   // principal value of pow(z, 1/3)
    const double m = std::sqrt(z.re * z.re + z.im * z.im);
    const double r = det_cbrt(m), th = det_atan2(z.im, z.re) / 3.0;
    double sn, cs;
    det_sincos(std::fabs(th), sn, cs);
    return Cx{r * cs, th < 0 ? -(r * sn) : r * sn};
}
void ap3p_solve_quartic(const double* f, double* roots) {
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
        const double c = w.re < 0 ? -det_cbrt(-w.re) : (w.re > 0 ? det_cbrt(w.re) : 0.0);
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
inline void v3_cross(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
inline double v3_norm(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline void m3_mult(const double a[3][3], const double b[3][3], double r[3][3]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}
// ap3p::solve + computePoses for four correspondences: up to four poses from the first three, sorted by the fourth's error
int ap3p_solve(const Camera& cam, const double* mu_in, const double* mv_in, const double P[4][3], double R[4][3][3], double t[4][3]) {
    const double inv_fx = 1. / cam.fu, inv_fy = 1. / cam.fv, cx_fx = cam.uc / cam.fu, cy_fy = cam.vc / cam.fv;
    double b[3][3];   // unit bearing vectors of the first three points
    for (int i = 0; i < 3; i++) {
        const double mu = inv_fx * mu_in[i] - cx_fx, mv = inv_fy * mv_in[i] - cy_fy;
        const double mk = 1. / std::sqrt(mu * mu + mv * mv + 1);
        b[i][0] = mu * mk;
        b[i][1] = mv * mk;
        b[i][2] = mk;
    }
    const double mu3 = inv_fx * mu_in[3] - cx_fx, mv3 = inv_fy * mv_in[3] - cy_fy;
    const double *w1 = P[0], *w2 = P[1], *w3 = P[2];
    double u0[3] = {w1[0] - w2[0], w1[1] - w2[1], w1[2] - w2[2]};
    const double nu0 = v3_norm(u0);
    const double k1[3] = {u0[0] / nu0, u0[1] / nu0, u0[2] / nu0};
    double k3[3], tz[3], v1[3], v2[3];
    v3_cross(b[0], b[1], k3);
    const double nk3 = v3_norm(k3);
    for (int i = 0; i < 3; i++) k3[i] /= nk3;
    v3_cross(b[0], k3, tz);
    v3_cross(b[0], b[2], v1);
    v3_cross(b[1], b[2], v2);
    const double u1[3] = {w1[0] - w3[0], w1[1] - w3[1], w1[2] - w3[2]};
    const double u1k1 = u1[0] * k1[0] + u1[1] * k1[1] + u1[2] * k1[2];
    const double k3b3 = k3[0] * b[2][0] + k3[1] * b[2][1] + k3[2] * b[2][2];
    double f11 = k3b3;
    double f13 = k3[0] * v1[0] + k3[1] * v1[1] + k3[2] * v1[2];
    const double f15 = -u1k1 * f11;
    double nl[3];
    v3_cross(u1, k1, nl);
    const double delta = v3_norm(nl);
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
    ap3p_solve_quartic(coeffs, s);
    double temp[3];
    v3_cross(k1, nl, temp);
    const double Ck1nl[3][3] = {{k1[0], nl[0], temp[0]}, {k1[1], nl[1], temp[1]}, {k1[2], nl[2], temp[2]}};
    const double Cb1k3tzT[3][3] = {{b[0][0], b[0][1], b[0][2]}, {k3[0], k3[1], k3[2]}, {tz[0], tz[1], tz[2]}};
    const double sc = delta / k3b3;
    const double b3p[3] = {sc * b[2][0], sc * b[2][1], sc * b[2][2]};
    double reproj_errors[4];
    int nb = 0;
    for (int i = 0; i < 4; i++) {
        const double ctheta1p = s[i];
        if (!(std::fabs(ctheta1p) <= 1)) continue;   // (upstream: abs(c) > 1 -> skip; a NaN root is skipped here)
        double stheta1p = std::sqrt(1 - ctheta1p * ctheta1p);
        stheta1p = (k3b3 > 0) ? stheta1p : -stheta1p;
        double ctheta3 = g1 * ctheta1p + g2;
        double stheta3 = g3 * ctheta1p + g4;
        const double ntheta3 = stheta1p / ((g5 * ctheta1p + g6) * ctheta1p + g7);
        ctheta3 *= ntheta3;
        stheta3 *= ntheta3;
        const double C13[3][3] = {{ctheta3, 0, -stheta3}, {stheta1p * stheta3, ctheta1p, stheta1p * ctheta3}, {ctheta1p * stheta3, -stheta1p, ctheta1p * ctheta3}};
        double tmp[3][3], Rm[3][3];
        m3_mult(Ck1nl, C13, tmp);
        m3_mult(tmp, Cb1k3tzT, Rm);
        const double rp3[3] = {w3[0] * Rm[0][0] + w3[1] * Rm[1][0] + w3[2] * Rm[2][0], w3[0] * Rm[0][1] + w3[1] * Rm[1][1] + w3[2] * Rm[2][1],
                               w3[0] * Rm[0][2] + w3[1] * Rm[1][2] + w3[2] * Rm[2][2]};
        for (int k = 0; k < 3; k++) t[nb][k] = stheta1p * b3p[k] - rp3[k];
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) R[nb][a][c] = Rm[c][a];   // the pose is the transpose
        const double X3p = R[nb][0][0] * P[3][0] + R[nb][0][1] * P[3][1] + R[nb][0][2] * P[3][2] + t[nb][0];
        const double Y3p = R[nb][1][0] * P[3][0] + R[nb][1][1] * P[3][1] + R[nb][1][2] * P[3][2] + t[nb][1];
        const double Z3p = R[nb][2][0] * P[3][0] + R[nb][2][1] * P[3][1] + R[nb][2][2] * P[3][2] + t[nb][2];
        const double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        reproj_errors[nb] = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
        nb++;
    }
    for (int i = 1; i < nb; i++)
        for (int j = i; j > 0 && reproj_errors[j - 1] > reproj_errors[j]; j--) {
            std::swap(reproj_errors[j], reproj_errors[j - 1]);
            for (int k = 0; k < 9; k++) std::swap((&R[j][0][0])[k], (&R[j - 1][0][0])[k]);
            for (int k = 0; k < 3; k++) std::swap(t[j][k], t[j - 1][k]);
        }
    return nb;
}

template <typename T>
bool solve_pnp_ap3p(const T* obj, const T* img, const double* K, double* rvec, double* tvec) {
    const Camera cam{K[0], K[4], K[2], K[5]};
    const double ifx = 1. / K[0], ify = 1. / K[4];
    double mu[4], mv[4], P[4][3];
    for (int i = 0; i < 4; i++) {
        const double xn = ((double)img[2 * i] - K[2]) * ifx, yn = ((double)img[2 * i + 1] - K[5]) * ify;
        mu[i] = (T)(K[0] * xn + K[2]);
        mv[i] = (T)(K[4] * yn + K[5]);
        for (int c = 0; c < 3; c++) P[i][c] = obj[3 * i + c];
    }
    double R[4][3][3], t[4][3];
    if (ap3p_solve(cam, mu, mv, P, R, t) <= 0) return false;
    rodrigues_to_vector(&R[0][0][0], rvec);
    for (int i = 0; i < 3; i++) tvec[i] = t[0][i];
    return true;
}

// ---- sqpnp.cpp (Terzakis & Lourakis, "A consistently fast and globally optimal solution to the PnP problem", ECCV 2020) ----
// solvePnP(SOLVEPNP_SQPNP) over the inliers, as recalled for OpenCV >= 4.7 (the revision with the FOAM nearest rotation and the
// majority cheirality test): undistortPoints with zero distortion, PoseSolver::solve, the solution with the smallest
// reprojection error. PARITY UNPINNED (no OpenCV in this image). Chosen where memory does not decide: sums in index order.
struct SqpnpSolution {
    double r[9], r_hat[9], t[3], sq_error;
};

struct SqpnpSolver {
    static constexpr double RANK_TOLERANCE = 1e-7, SQP_SQUARED_TOLERANCE = 1e-10, SQP_DET_THRESHOLD = 1.001;
    static constexpr double ORTHOGONALITY_SQUARED_ERROR_THRESHOLD = 1e-8, EQUAL_VECTORS_SQUARED_DIFF = 1e-10;
    static constexpr double EQUAL_SQUARED_ERRORS_DIFF = 1e-6, POINT_VARIANCE_THRESHOLD = 1e-5;
    static constexpr int SQP_MAX_ITERATION = 15;

    double omega_[9][9], s_[9], u_[9][9] /* u_[k][i]: component k of eigenvector i */, p_[3][9], point_mean_[3];
    int num_null_vectors_ = -1, num_solutions_ = 0;
    SqpnpSolution solutions_[18];
    const double* obj_ = nullptr;
    int n_ = 0;

    static double det3x3(const double* e) {
        return e[0] * e[4] * e[8] + e[1] * e[5] * e[6] + e[2] * e[3] * e[7] - e[6] * e[4] * e[2] - e[7] * e[5] * e[0] - e[8] * e[3] * e[1];
    }

    static double orthogonalityError(const double* a) {
        const double sq_norm_a1 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], sq_norm_a2 = a[3] * a[3] + a[4] * a[4] + a[5] * a[5],
                     sq_norm_a3 = a[6] * a[6] + a[7] * a[7] + a[8] * a[8];
        const double dot_a1a2 = a[0] * a[3] + a[1] * a[4] + a[2] * a[5], dot_a1a3 = a[0] * a[6] + a[1] * a[7] + a[2] * a[8],
                     dot_a2a3 = a[3] * a[6] + a[4] * a[7] + a[5] * a[8];
        return (sq_norm_a1 - 1) * (sq_norm_a1 - 1) + (sq_norm_a2 - 1) * (sq_norm_a2 - 1) + (sq_norm_a3 - 1) * (sq_norm_a3 - 1) +
               2 * (dot_a1a2 * dot_a1a2 + dot_a1a3 * dot_a1a3 + dot_a2a3 * dot_a2a3);
    }

    // analyticalInverse3x3Symm: closed form, cv::invert(DECOMP_SVD) when the determinant is below 1e-8
    static void analyticalInverse3x3Symm(const double Q[3][3], double Qinv[3][3]) {
        const double a = Q[0][0], b = Q[1][0], d = Q[1][1], c = Q[2][0], e = Q[2][1], f = Q[2][2];
        const double t2 = e * e, t4 = a * d, t7 = b * b, t9 = b * c, t12 = c * c;
        const double det = -t4 * f + a * t2 + t7 * f - 2.0 * t9 * e + t12 * d;
        if (std::fabs(det) < 1e-8) {
            svd_invert3(&Q[0][0], &Qinv[0][0]);
            return;
        }
        const double t15 = 1.0 / det, t20 = (-b * f + c * e) * t15, t24 = (b * e - c * d) * t15, t30 = (a * e - t9) * t15;
        Qinv[0][0] = (-d * f + t2) * t15;
        Qinv[0][1] = Qinv[1][0] = -t20;
        Qinv[0][2] = Qinv[2][0] = -t24;
        Qinv[1][1] = -(a * f - t12) * t15;
        Qinv[1][2] = Qinv[2][1] = t30;
        Qinv[2][2] = -(t4 - t7) * t15;
    }

    // nearestRotationMatrixFOAM: the orthogonal factor of e's polar decomposition from the largest root of FOAM's quartic
    static void nearestRotationMatrix(const double* e, double* r) {
        double adj_e[9];
        adj_e[0] = e[4] * e[8] - e[5] * e[7];
        adj_e[1] = e[2] * e[7] - e[1] * e[8];
        adj_e[2] = e[1] * e[5] - e[2] * e[4];
        adj_e[3] = e[5] * e[6] - e[3] * e[8];
        adj_e[4] = e[0] * e[8] - e[2] * e[6];
        adj_e[5] = e[2] * e[3] - e[0] * e[5];
        adj_e[6] = e[3] * e[7] - e[4] * e[6];
        adj_e[7] = e[1] * e[6] - e[0] * e[7];
        adj_e[8] = e[0] * e[4] - e[1] * e[3];
        const double det_e = e[0] * e[4] * e[8] - e[0] * e[5] * e[7] - e[1] * e[3] * e[8] + e[2] * e[3] * e[7] + e[1] * e[6] * e[5] - e[2] * e[6] * e[4];
        double e_sq = 0, adj_e_sq = 0;
        for (int i = 0; i < 9; i++) e_sq += e[i] * e[i];
        for (int i = 0; i < 9; i++) adj_e_sq += adj_e[i] * adj_e[i];
        double l = 0.5 * (e_sq + 3.0), lprev = 0.0;
        if (det_e < 0.0) l = -l;
        for (int i = 15; std::fabs(l - lprev) > 1E-12 * std::fabs(lprev) && i > 0; --i) {
            const double tmp = l * l - e_sq;
            const double p = tmp * tmp - 8.0 * l * det_e - 4.0 * adj_e_sq;
            const double pp = 8.0 * (0.5 * tmp * l - det_e);
            lprev = l;
            l -= p / pp;
        }
        const double a = l * l + e_sq;
        double e_et[9], tmp[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) e_et[3 * i + j] = e[3 * i] * e[3 * j] + e[3 * i + 1] * e[3 * j + 1] + e[3 * i + 2] * e[3 * j + 2];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) tmp[3 * i + j] = e_et[3 * i] * e[j] + e_et[3 * i + 1] * e[3 + j] + e_et[3 * i + 2] * e[6 + j];
        double denom = l * (l * l - e_sq) - 2.0 * det_e;
        // DOCUMENTED DEVIATION: denom = (s1 + s2)(s1 + s3)(s2 + s3) of e's singular values. A rank-one e - every null vector of Omega when the
        // object points are coplanar is one, w n' - makes it vanish and FOAM returns e / l, no rotation at all (the SQP that starts there divides
        // by zero). Such an e is completed to a rotation the way nearestRotationMatrixSVD (OpenCV <= 4.6) does it: U diag(1, 1, det U det V') V'.
        if (!(std::fabs(denom) >= 1e-3 * (e_sq * std::sqrt(e_sq)))) {
            double W[3], Ut[9], Vt[9];
            svd(e, 3, 3, W, Ut, Vt);
            const double detuv = det3x3(Ut) * det3x3(Vt);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) r[3 * i + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + (Ut[6 + i] * detuv) * Vt[6 + j];
            return;
        }
        denom = 1.0 / denom;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r[3 * i + j] = (a * e[3 * i + j] + 2.0 * (l * adj_e[3 * j + i] - tmp[3 * i + j])) * denom;
    }

    void computeOmega(const double* obj, const double* nimg, int n) {
        obj_ = obj;
        n_ = n;
        std::memset(omega_, 0, sizeof omega_);
        double qa_sum[3][9];
        std::memset(qa_sum, 0, sizeof qa_sum);
        double sum_img_x = 0, sum_img_y = 0, sum_obj[3] = {0, 0, 0}, sum_sq_norm = 0;
        for (int i = 0; i < n; i++) {
            const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2], x = nimg[2 * i], y = nimg[2 * i + 1];
            const double sq_norm = x * x + y * y;
            sum_sq_norm += sq_norm;
            sum_img_x += x;
            sum_img_y += y;
            sum_obj[0] += X;
            sum_obj[1] += Y;
            sum_obj[2] += Z;
            const double X2 = X * X, XY = X * Y, XZ = X * Z, Y2 = Y * Y, YZ = Y * Z, Z2 = Z * Z;
            omega_[0][0] += X2; omega_[0][1] += XY; omega_[0][2] += XZ; omega_[1][1] += Y2; omega_[1][2] += YZ; omega_[2][2] += Z2;
            omega_[0][6] += -x * X2; omega_[0][7] += -x * XY; omega_[0][8] += -x * XZ;
            omega_[1][7] += -x * Y2; omega_[1][8] += -x * YZ;
            omega_[2][8] += -x * Z2;
            omega_[3][6] += -y * X2; omega_[3][7] += -y * XY; omega_[3][8] += -y * XZ;
            omega_[4][7] += -y * Y2; omega_[4][8] += -y * YZ;
            omega_[5][8] += -y * Z2;
            omega_[6][6] += sq_norm * X2; omega_[6][7] += sq_norm * XY; omega_[6][8] += sq_norm * XZ;
            omega_[7][7] += sq_norm * Y2; omega_[7][8] += sq_norm * YZ;
            omega_[8][8] += sq_norm * Z2;
            qa_sum[0][0] += X; qa_sum[0][1] += Y; qa_sum[0][2] += Z;
            qa_sum[1][3] += X; qa_sum[1][4] += Y; qa_sum[1][5] += Z;
            qa_sum[0][6] += -x * X; qa_sum[0][7] += -x * Y; qa_sum[0][8] += -x * Z;
            qa_sum[1][6] += -y * X; qa_sum[1][7] += -y * Y; qa_sum[1][8] += -y * Z;
            qa_sum[2][0] += -x * X; qa_sum[2][1] += -x * Y; qa_sum[2][2] += -x * Z;
            qa_sum[2][3] += -y * X; qa_sum[2][4] += -y * Y; qa_sum[2][5] += -y * Z;
            qa_sum[2][6] += sq_norm * X; qa_sum[2][7] += sq_norm * Y; qa_sum[2][8] += sq_norm * Z;
        }
        omega_[1][6] = omega_[0][7]; omega_[2][6] = omega_[0][8]; omega_[2][7] = omega_[1][8];
        omega_[4][6] = omega_[3][7]; omega_[5][6] = omega_[3][8]; omega_[5][7] = omega_[4][8];
        omega_[7][6] = omega_[6][7]; omega_[8][6] = omega_[6][8]; omega_[8][7] = omega_[7][8];
        omega_[3][3] = omega_[0][0]; omega_[3][4] = omega_[0][1]; omega_[3][5] = omega_[0][2];
        omega_[4][4] = omega_[1][1]; omega_[4][5] = omega_[1][2];
        omega_[5][5] = omega_[2][2];
        for (int r = 0; r < 9; r++)
            for (int c = 0; c < r; c++) omega_[r][c] = omega_[c][r];
        double q[3][3], q_inv[3][3];
        q[0][0] = n; q[0][1] = 0; q[0][2] = -sum_img_x;
        q[1][0] = 0; q[1][1] = n; q[1][2] = -sum_img_y;
        q[2][0] = -sum_img_x; q[2][1] = -sum_img_y; q[2][2] = sum_sq_norm;
        analyticalInverse3x3Symm(q, q_inv);
        for (int i = 0; i < 3; i++)          // p_ = -q_inv * qa_sum
            for (int j = 0; j < 9; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += -q_inv[i][k] * qa_sum[k][j];
                p_[i][j] = s;
            }
        for (int i = 0; i < 9; i++)          // omega_ += qa_sum.t() * p_
            for (int j = 0; j < 9; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += qa_sum[k][i] * p_[k][j];
                omega_[i][j] += s;
            }
        // cv::SVD(omega_, FULL_UV): u_ = vt.t()
        double Ut[81], Vt[81];
        svd(&omega_[0][0], 9, 9, s_, Ut, Vt);
        for (int i = 0; i < 9; i++)
            for (int k = 0; k < 9; k++) u_[k][i] = Vt[i * 9 + k];
        while (7 - num_null_vectors_ >= 0 && s_[7 - num_null_vectors_] < RANK_TOLERANCE) num_null_vectors_++;
        ++num_null_vectors_;
        const double inv_n = 1.0 / n;
        for (int k = 0; k < 3; k++) point_mean_[k] = sum_obj[k] * inv_n;
    }

    // an orthonormal basis H of the row space of the constraints' Jacobian at r, K = J H (lower triangular), N = a basis of its null space
    static void computeRowAndNullspace(const double* r, double H[9][6], double N[9][3], double K[6][6]) {
        const double norm_threshold = 0.1;
        std::memset(H, 0, sizeof(double) * 54);
        std::memset(K, 0, sizeof(double) * 36);
        // 1. q1
        const double norm_r1 = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const double inv_norm_r1 = norm_r1 > 1e-5 ? 1.0 / norm_r1 : 0.0;
        H[0][0] = r[0] * inv_norm_r1; H[1][0] = r[1] * inv_norm_r1; H[2][0] = r[2] * inv_norm_r1;
        K[0][0] = 2 * norm_r1;
        // 2. q2
        const double norm_r2 = std::sqrt(r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
        const double inv_norm_r2 = 1.0 / norm_r2;
        H[3][1] = r[3] * inv_norm_r2; H[4][1] = r[4] * inv_norm_r2; H[5][1] = r[5] * inv_norm_r2;
        K[1][0] = 0; K[1][1] = 2 * norm_r2;
        // 3. q3
        const double norm_r3 = std::sqrt(r[6] * r[6] + r[7] * r[7] + r[8] * r[8]);
        const double inv_norm_r3 = 1.0 / norm_r3;
        H[6][2] = r[6] * inv_norm_r3; H[7][2] = r[7] * inv_norm_r3; H[8][2] = r[8] * inv_norm_r3;
        K[2][0] = K[2][1] = 0; K[2][2] = 2 * norm_r3;
        // 4. q4
        const double dot_j4q1 = r[3] * H[0][0] + r[4] * H[1][0] + r[5] * H[2][0], dot_j4q2 = r[0] * H[3][1] + r[1] * H[4][1] + r[2] * H[5][1];
        H[0][3] = r[3] - dot_j4q1 * H[0][0]; H[1][3] = r[4] - dot_j4q1 * H[1][0]; H[2][3] = r[5] - dot_j4q1 * H[2][0];
        H[3][3] = r[0] - dot_j4q2 * H[3][1]; H[4][3] = r[1] - dot_j4q2 * H[4][1]; H[5][3] = r[2] - dot_j4q2 * H[5][1];
        const double inv_norm_j4 = 1.0 / std::sqrt(H[0][3] * H[0][3] + H[1][3] * H[1][3] + H[2][3] * H[2][3] + H[3][3] * H[3][3] + H[4][3] * H[4][3] + H[5][3] * H[5][3]);
        for (int i = 0; i < 6; i++) H[i][3] *= inv_norm_j4;
        K[3][0] = r[3] * H[0][0] + r[4] * H[1][0] + r[5] * H[2][0];
        K[3][1] = r[0] * H[3][1] + r[1] * H[4][1] + r[2] * H[5][1];
        K[3][2] = 0;
        K[3][3] = r[3] * H[0][3] + r[4] * H[1][3] + r[5] * H[2][3] + r[0] * H[3][3] + r[1] * H[4][3] + r[2] * H[5][3];
        // 5. q5
        const double dot_j5q2 = r[6] * H[3][1] + r[7] * H[4][1] + r[8] * H[5][1];
        const double dot_j5q3 = r[3] * H[6][2] + r[4] * H[7][2] + r[5] * H[8][2];
        const double dot_j5q4 = r[6] * H[3][3] + r[7] * H[4][3] + r[8] * H[5][3];
        H[0][4] = -dot_j5q4 * H[0][3]; H[1][4] = -dot_j5q4 * H[1][3]; H[2][4] = -dot_j5q4 * H[2][3];
        H[3][4] = r[6] - dot_j5q2 * H[3][1] - dot_j5q4 * H[3][3];
        H[4][4] = r[7] - dot_j5q2 * H[4][1] - dot_j5q4 * H[4][3];
        H[5][4] = r[8] - dot_j5q2 * H[5][1] - dot_j5q4 * H[5][3];
        H[6][4] = r[3] - dot_j5q3 * H[6][2]; H[7][4] = r[4] - dot_j5q3 * H[7][2]; H[8][4] = r[5] - dot_j5q3 * H[8][2];
        {
            double sq = 0;
            for (int i = 0; i < 9; i++) sq += H[i][4] * H[i][4];
            const double inv = 1.0 / std::sqrt(sq);
            for (int i = 0; i < 9; i++) H[i][4] *= inv;
        }
        K[4][0] = 0;
        K[4][1] = r[6] * H[3][1] + r[7] * H[4][1] + r[8] * H[5][1];
        K[4][2] = r[3] * H[6][2] + r[4] * H[7][2] + r[5] * H[8][2];
        K[4][3] = r[6] * H[3][3] + r[7] * H[4][3] + r[8] * H[5][3];
        K[4][4] = r[6] * H[3][4] + r[7] * H[4][4] + r[8] * H[5][4] + r[3] * H[6][4] + r[4] * H[7][4] + r[5] * H[8][4];
        // 6. q6
        const double dot_j6q1 = r[6] * H[0][0] + r[7] * H[1][0] + r[8] * H[2][0];
        const double dot_j6q3 = r[0] * H[6][2] + r[1] * H[7][2] + r[2] * H[8][2];
        const double dot_j6q4 = r[6] * H[0][3] + r[7] * H[1][3] + r[8] * H[2][3];
        const double dot_j6q5 = r[0] * H[6][4] + r[1] * H[7][4] + r[2] * H[8][4] + r[6] * H[0][4] + r[7] * H[1][4] + r[8] * H[2][4];
        H[0][5] = r[6] - dot_j6q1 * H[0][0] - dot_j6q4 * H[0][3] - dot_j6q5 * H[0][4];
        H[1][5] = r[7] - dot_j6q1 * H[1][0] - dot_j6q4 * H[1][3] - dot_j6q5 * H[1][4];
        H[2][5] = r[8] - dot_j6q1 * H[2][0] - dot_j6q4 * H[2][3] - dot_j6q5 * H[2][4];
        H[3][5] = -dot_j6q5 * H[3][4] - dot_j6q4 * H[3][3];
        H[4][5] = -dot_j6q5 * H[4][4] - dot_j6q4 * H[4][3];
        H[5][5] = -dot_j6q5 * H[5][4] - dot_j6q4 * H[5][3];
        H[6][5] = r[0] - dot_j6q3 * H[6][2] - dot_j6q5 * H[6][4];
        H[7][5] = r[1] - dot_j6q3 * H[7][2] - dot_j6q5 * H[7][4];
        H[8][5] = r[2] - dot_j6q3 * H[8][2] - dot_j6q5 * H[8][4];
        {
            double sq = 0;
            for (int i = 0; i < 9; i++) sq += H[i][5] * H[i][5];
            const double inv = 1.0 / std::sqrt(sq);
            for (int i = 0; i < 9; i++) H[i][5] *= inv;
        }
        K[5][0] = r[6] * H[0][0] + r[7] * H[1][0] + r[8] * H[2][0];
        K[5][1] = 0;
        K[5][2] = r[0] * H[6][2] + r[1] * H[7][2] + r[2] * H[8][2];
        K[5][3] = r[6] * H[0][3] + r[7] * H[1][3] + r[8] * H[2][3];
        K[5][4] = r[6] * H[0][4] + r[7] * H[1][4] + r[8] * H[2][4] + r[0] * H[6][4] + r[1] * H[7][4] + r[2] * H[8][4];
        K[5][5] = r[6] * H[0][5] + r[7] * H[1][5] + r[8] * H[2][5] + r[0] * H[6][5] + r[1] * H[7][5] + r[2] * H[8][5];
        // the projector onto the null space of H, and three of its columns: the longest, the one most orthogonal to it, then to both
        double Pn[9][9];
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += H[i][k] * H[j][k];
                Pn[i][j] = (i == j ? 1.0 : 0.0) - s;
            }
        auto col_dot = [&](int a, int b) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += Pn[k][a] * Pn[k][b];
            return s;
        };
        int index1 = 0, index2 = 0, index3 = 0;
        double max_norm1 = DBL_MIN, min_dot12 = DBL_MAX, min_dot1323 = DBL_MAX, col_norms[9];
        for (int i = 0; i < 9; i++) {
            col_norms[i] = std::sqrt(col_dot(i, i));
            if (col_norms[i] >= norm_threshold && max_norm1 < col_norms[i]) {
                max_norm1 = col_norms[i];
                index1 = i;
            }
        }
        for (int k = 0; k < 9; k++) N[k][0] = Pn[k][index1] * (1.0 / max_norm1);
        for (int i = 0; i < 9; i++) {
            if (i == index1) continue;
            if (col_norms[i] >= norm_threshold) {
                const double cos_v1_x_col = std::fabs(col_dot(i, index1) / col_norms[i]);
                if (cos_v1_x_col <= min_dot12) {
                    index2 = i;
                    min_dot12 = cos_v1_x_col;
                }
            }
        }
        {
            double d = 0;
            for (int k = 0; k < 9; k++) d += Pn[k][index2] * N[k][0];
            for (int k = 0; k < 9; k++) N[k][1] = Pn[k][index2] - d * N[k][0];
            double sq = 0;
            for (int k = 0; k < 9; k++) sq += N[k][1] * N[k][1];
            const double inv = 1.0 / std::sqrt(sq);
            for (int k = 0; k < 9; k++) N[k][1] *= inv;
        }
        for (int i = 0; i < 9; i++) {
            if (i == index2 || i == index1) continue;
            if (col_norms[i] >= norm_threshold) {
                const double cos_v1_x_col = std::fabs(col_dot(i, index1) / col_norms[i]);
                const double cos_v2_x_col = std::fabs(col_dot(i, index2) / col_norms[i]);
                if (cos_v1_x_col + cos_v2_x_col <= min_dot1323) {
                    index3 = i;
                    min_dot1323 = cos_v2_x_col + cos_v2_x_col;   // (sic: sqpnp.cpp stores twice the second cosine)
                }
            }
        }
        {
            double d1 = 0, d0 = 0;
            for (int k = 0; k < 9; k++) d1 += Pn[k][index3] * N[k][1];
            for (int k = 0; k < 9; k++) d0 += Pn[k][index3] * N[k][0];
            for (int k = 0; k < 9; k++) N[k][2] = Pn[k][index3] - d1 * N[k][1] - d0 * N[k][0];
            double sq = 0;
            for (int k = 0; k < 9; k++) sq += N[k][2] * N[k][2];
            const double inv = 1.0 / std::sqrt(sq);
            for (int k = 0; k < 9; k++) N[k][2] *= inv;
        }
    }

    void solveSQPSystem(const double* r, double* delta) const {
        const double sqnorm_r1 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2], sqnorm_r2 = r[3] * r[3] + r[4] * r[4] + r[5] * r[5],
                     sqnorm_r3 = r[6] * r[6] + r[7] * r[7] + r[8] * r[8];
        const double dot_r1r2 = r[0] * r[3] + r[1] * r[4] + r[2] * r[5], dot_r1r3 = r[0] * r[6] + r[1] * r[7] + r[2] * r[8],
                     dot_r2r3 = r[3] * r[6] + r[4] * r[7] + r[5] * r[8];
        double N[9][3], H[9][6], JH[6][6];
        computeRowAndNullspace(r, H, N, JH);
        const double g[6] = {1 - sqnorm_r1, 1 - sqnorm_r2, 1 - sqnorm_r3, -dot_r1r2, -dot_r2r3, -dot_r1r3};
        double x[6];
        x[0] = g[0] / JH[0][0];
        x[1] = g[1] / JH[1][1];
        x[2] = g[2] / JH[2][2];
        x[3] = (g[3] - JH[3][0] * x[0] - JH[3][1] * x[1]) / JH[3][3];
        x[4] = (g[4] - JH[4][1] * x[1] - JH[4][2] * x[2] - JH[4][3] * x[3]) / JH[4][4];
        x[5] = (g[5] - JH[5][0] * x[0] - JH[5][2] * x[2] - JH[5][3] * x[3] - JH[5][4] * x[4]) / JH[5][5];
        for (int i = 0; i < 9; i++) {   // delta = H x
            double s = 0;
            for (int k = 0; k < 6; k++) s += H[i][k] * x[k];
            delta[i] = s;
        }
        double NtOmega[3][9], W[3][3], Winv[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 9; j++) {
                double s = 0;
                for (int k = 0; k < 9; k++) s += N[k][i] * omega_[k][j];
                NtOmega[i][j] = s;
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 9; k++) s += NtOmega[i][k] * N[k][j];
                W[i][j] = s;
            }
        analyticalInverse3x3Symm(W, Winv);
        // y = -Winv * NtOmega * (delta + r), evaluated left to right: (-Winv * NtOmega) first
        double WN[3][9], y[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 9; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += -Winv[i][k] * NtOmega[k][j];
                WN[i][j] = s;
            }
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += WN[i][k] * (delta[k] + r[k]);
            y[i] = s;
        }
        for (int i = 0; i < 9; i++) {   // delta += N y
            double s = 0;
            for (int k = 0; k < 3; k++) s += N[i][k] * y[k];
            delta[i] += s;
        }
    }

    SqpnpSolution runSQP(const double* r0) const {
        double r[9], delta[9];
        std::memcpy(r, r0, sizeof r);
        double delta_squared_norm = DBL_MAX;
        int step = 0;
        while (delta_squared_norm > SQP_SQUARED_TOLERANCE && step++ < SQP_MAX_ITERATION) {
            solveSQPSystem(r, delta);
            for (int i = 0; i < 9; i++) r[i] += delta[i];
            delta_squared_norm = 0;
            for (int i = 0; i < 9; i++) delta_squared_norm += delta[i] * delta[i];
        }
        SqpnpSolution solution{};
        double det_r = det3x3(r);
        if (det_r < 0) {
            for (int i = 0; i < 9; i++) r[i] = -r[i];
            det_r = -det_r;
        }
        std::memcpy(solution.r, r, sizeof r);
        if (det_r > SQP_DET_THRESHOLD)
            nearestRotationMatrix(r, solution.r_hat);
        else
            std::memcpy(solution.r_hat, r, sizeof r);
        return solution;
    }

    void translationOf(SqpnpSolution& s) const {   // t = p_ * r_hat
        for (int i = 0; i < 3; i++) {
            double a = 0;
            for (int k = 0; k < 9; k++) a += p_[i][k] * s.r_hat[k];
            s.t[i] = a;
        }
    }

    bool positiveDepth(const SqpnpSolution& s) const {
        const double* r = s.r_hat;
        return r[6] * point_mean_[0] + r[7] * point_mean_[1] + r[8] * point_mean_[2] + s.t[2] > 0;
    }

    bool positiveMajorityDepths(const SqpnpSolution& s) const {
        const double* r = s.r_hat;
        int npos = 0, nneg = 0;
        for (int i = 0; i < n_; i++) {
            if (r[6] * obj_[3 * i] + r[7] * obj_[3 * i + 1] + r[8] * obj_[3 * i + 2] + s.t[2] > 0)
                ++npos;
            else
                ++nneg;
        }
        return npos >= nneg;
    }

    void checkSolution(SqpnpSolution& solution, double& min_error) {
        // DOCUMENTED DEVIATION (a guard sqpnp.cpp does not have): runSQP hands back r unprojected when det r <= 1.001, so a start inside Omega's
        // null space (coplanar object points: the rank-one matrices w n') can come back as a near-zero, non-orthogonal "solution" of cost 0
        // whose depth test passes or fails by rounding; accepted, it ends the search. Candidates that are not rotations are dropped here.
        if (!(orthogonalityError(solution.r_hat) <= 0.1)) return;
        if (!(positiveDepth(solution) || positiveMajorityDepths(solution))) return;
        double e = 0;   // (omega_ * r_hat) . r_hat
        for (int i = 0; i < 9; i++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += omega_[i][k] * solution.r_hat[k];
            e += s * solution.r_hat[i];
        }
        solution.sq_error = e;
        if (std::fabs(min_error - solution.sq_error) > EQUAL_SQUARED_ERRORS_DIFF) {
            if (min_error > solution.sq_error) {
                min_error = solution.sq_error;
                solutions_[0] = solution;
                num_solutions_ = 1;
            }
        } else {
            bool found = false;
            for (int i = 0; i < num_solutions_; i++) {
                double d = 0;
                for (int k = 0; k < 9; k++) d += (solutions_[i].r_hat[k] - solution.r_hat[k]) * (solutions_[i].r_hat[k] - solution.r_hat[k]);
                if (d < EQUAL_VECTORS_SQUARED_DIFF) {
                    if (solutions_[i].sq_error > solution.sq_error) solutions_[i] = solution;
                    found = true;
                    break;
                }
            }
            if (!found) solutions_[num_solutions_++] = solution;
            if (min_error > solution.sq_error) min_error = solution.sq_error;
        }
    }

    void tryBothSigns(const double* e, double& min_sq_err) {
        double neg[9], start[9];
        for (int k = 0; k < 9; k++) neg[k] = -e[k];
        nearestRotationMatrix(e, start);
        SqpnpSolution a = runSQP(start);
        translationOf(a);
        checkSolution(a, min_sq_err);
        nearestRotationMatrix(neg, start);
        SqpnpSolution b = runSQP(start);
        translationOf(b);
        checkSolution(b, min_sq_err);
    }

    void solveInternal() {
        double min_sq_err = DBL_MAX;
        const int num_eigen_points = num_null_vectors_ > 0 ? num_null_vectors_ : 1;
        const double SQRT3 = std::sqrt(3.0);
        for (int i = 9 - num_eigen_points; i < 9; i++) {
            double e[9];
            for (int k = 0; k < 9; k++) e[k] = SQRT3 * u_[k][i];
            if (orthogonalityError(e) < ORTHOGONALITY_SQUARED_ERROR_THRESHOLD) {   // e is a rotation up to its sign: no SQP needed
                SqpnpSolution s{};
                const double d = det3x3(e);
                for (int k = 0; k < 9; k++) s.r_hat[k] = d * e[k];
                translationOf(s);
                checkSolution(s, min_sq_err);
            } else {
                tryBothSigns(e, min_sq_err);
            }
        }
        int index, c = 1;
        while ((index = 9 - num_eigen_points - c) > 0 && min_sq_err > 3 * s_[index]) {
            double e[9];
            for (int k = 0; k < 9; k++) e[k] = u_[k][index];
            tryBothSigns(e, min_sq_err);
            c++;
        }
    }
};

// solvePnP(flags = SOLVEPNP_SQPNP) on double points: returns false where OpenCV's asserts would throw (degenerate point sets)
bool solve_pnp_sqpnp(const double* obj, const double* img, int n, const double* K, double* rvec, double* tvec) {
    if (n < 3) return false;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    std::vector<double> nimg(2 * (size_t)n);
    for (int i = 0; i < n; i++) {   // undistortPoints with zero distortion
        nimg[2 * i] = (img[2 * i] - cx) * ifx;
        nimg[2 * i + 1] = (img[2 * i + 1] - cy) * ify;
    }
    SqpnpSolver solver;
    solver.computeOmega(obj, nimg.data(), n);
    if (!(solver.s_[0] >= 1e-7) || solver.num_null_vectors_ > 6) return false;
    solver.solveInternal();
    if (solver.num_solutions_ <= 0) return false;
    // solvePnPGeneric orders the solutions by their reprojection error; solvePnP takes the first
    int best = 0;
    double best_err = DBL_MAX;
    for (int si = 0; si < solver.num_solutions_; si++) {
        const SqpnpSolution& s = solver.solutions_[si];
        double err = 0;
        for (int i = 0; i < n; i++) {
            const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
            const double xc = s.r_hat[0] * X + s.r_hat[1] * Y + s.r_hat[2] * Z + s.t[0], yc = s.r_hat[3] * X + s.r_hat[4] * Y + s.r_hat[5] * Z + s.t[1],
                         zc = s.r_hat[6] * X + s.r_hat[7] * Y + s.r_hat[8] * Z + s.t[2];
            const double iz = 1. / zc, du = xc * iz * fx + cx - img[2 * i], dv = yc * iz * fy + cy - img[2 * i + 1];
            err += du * du + dv * dv;
        }
        if (err < best_err) {
            best_err = err;
            best = si;
        }
    }
    rodrigues_to_vector(solver.solutions_[best].r_hat, rvec);
    for (int k = 0; k < 3; k++) tvec[k] = solver.solutions_[best].t[k];
    return true;
}

// ---- ippe.cpp (Collins & Bartoli, "Infinitesimal plane-based pose estimation", IJCV 2014) -----------------------------------------
// solvePnP(flags = SOLVEPNP_IPPE) over the inliers, as recalled for OpenCV 4.8: undistortPoints with zero distortion,
// IPPE::PoseSolver::solveGeneric (object points moved to the plane z = 0 about their centroid, homography by Harker & O'Leary's
// method, the two rotations from the homography's Jacobian at the origin, a translation for each by linear least squares), the pose
// that reprojects better first. Object points that are not coplanar within IPPE_SMALL = 1e-3 make OpenCV's solver throw inside
// solvePnPGeneric's try block: no solution. PARITY UNPINNED. DOCUMENTED DEVIATIONS: the eigenvector of the smallest eigenvalue of the
// 3 x 3 matrix D'D is taken from this file's one-sided Jacobi SVD, not cv::eigen (the same vector to rounding; H is divided by H(2,2), so
// its sign is immaterial); the square roots in computeRotations are taken of max(0, .) (rounding can push their arguments below zero);
// rotateVec2ZAxis turns -z onto +z by half a turn about x (see there).
struct Ippe {
    static constexpr double IPPE_SMALL = 1e-3;

    static void rotateVec2ZAxis(const double a[3], double Ra[9]) {
        double ax = a[0], ay = a[1], az = a[2];
        const double nrm = std::sqrt(ax * ax + ay * ay + az * az);
        ax = ax / nrm;
        ay = ay / nrm;
        az = az / nrm;
        const double c = az;
        if (std::fabs(1.0 + c) < (double)FLT_EPSILON) {
            // a = -z: half a turn about x. (Memory of ippe.cpp says diag(1, 1, -1) here, which is a reflection: the pose built on it has
            // det -1 and Rodrigues makes nonsense of it - tests/test_external_anchors.py, a plane z = const seen from its back. GUESSED.)
            for (int i = 0; i < 9; i++) Ra[i] = 0;
            Ra[0] = 1.0;
            Ra[4] = -1.0;
            Ra[8] = -1.0;
        } else {
            const double d = 1.0 / (1.0 + c), ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
            Ra[0] = -ax2 * d + 1.0;
            Ra[1] = -axay * d;
            Ra[2] = -ax;
            Ra[3] = -axay * d;
            Ra[4] = -ay2 * d + 1.0;
            Ra[5] = -ay;
            Ra[6] = ax;
            Ra[7] = ay;
            Ra[8] = 1.0 - (ax2 + ay2) * d;
        }
    }

    // zero-centred object points on the plane z = 0 (canon: n x 2) and the 4 x 4 transform that takes model points there; false = not coplanar
    static bool makeCanonicalObjectPoints(const double* obj, int n, std::vector<double>& canon, double M2C[16]) {
        std::vector<double> UZero(3 * (size_t)n);
        double xBar = 0, yBar = 0, zBar = 0;
        bool isOnZPlane = true;
        for (int i = 0; i < n; i++) {
            const double x = obj[3 * i], y = obj[3 * i + 1], z = obj[3 * i + 2];
            xBar += x;
            yBar += y;
            zBar += z;
            if (std::fabs(z) > IPPE_SMALL) isOnZPlane = false;
        }
        xBar = xBar / n;
        yBar = yBar / n;
        zBar = zBar / n;
        for (int i = 0; i < n; i++) {
            UZero[3 * i] = obj[3 * i] - xBar;
            UZero[3 * i + 1] = obj[3 * i + 1] - yBar;
            UZero[3 * i + 2] = obj[3 * i + 2] - zBar;
        }
        canon.resize(2 * (size_t)n);
        for (int i = 0; i < 16; i++) M2C[i] = (i % 5 == 0) ? 1.0 : 0.0;
        if (isOnZPlane) {
            M2C[3] = -xBar;
            M2C[7] = -yBar;
            M2C[11] = -zBar;
            for (int i = 0; i < n; i++) {
                canon[2 * i] = UZero[3 * i];
                canon[2 * i + 1] = UZero[3 * i + 1];
            }
            return true;
        }
        double R[9];
        // computeObjextSpaceR3Pts: the plane's normal from the first three points
        const double *p1 = obj, *p2 = obj + 3, *p3 = obj + 6;
        double nx = (p1[1] - p2[1]) * (p1[2] - p3[2]) - (p1[1] - p3[1]) * (p1[2] - p2[2]);
        double ny = (p1[0] - p3[0]) * (p1[2] - p2[2]) - (p1[0] - p2[0]) * (p1[2] - p3[2]);
        double nz = (p1[0] - p2[0]) * (p1[1] - p3[1]) - (p1[0] - p3[0]) * (p1[1] - p2[1]);
        const double nrm = std::sqrt(nx * nx + ny * ny + nz * nz);
        if (nrm > IPPE_SMALL) {
            const double v[3] = {nx / nrm, ny / nrm, nz / nrm};
            rotateVec2ZAxis(v, R);
        } else {
            // computeObjextSpaceRSvD: R = U' of the SVD of UZero UZero'
            double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, W[3], Ut[9], Vt[9];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int i = 0; i < n; i++) s += UZero[3 * i + a] * UZero[3 * i + b];
                    S[3 * a + b] = s;
                }
            svd(S, 3, 3, W, Ut, Vt);
            if (!(W[2] / W[1] < IPPE_SMALL)) return false;   // CV_Assert(s3 / s2 < IPPE_SMALL)
            std::memcpy(R, Ut, sizeof R);
            const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
            if (det < 0) {
                R[6] = -R[6];
                R[7] = -R[7];
                R[8] = -R[8];
            }
        }
        for (int i = 0; i < n; i++) {
            const double* u = &UZero[3 * i];
            const double ax = R[0] * u[0] + R[1] * u[1] + R[2] * u[2], ay = R[3] * u[0] + R[4] * u[1] + R[5] * u[2], az = R[6] * u[0] + R[7] * u[1] + R[8] * u[2];
            canon[2 * i] = ax;
            canon[2 * i + 1] = ay;
            if (std::fabs(az) > IPPE_SMALL) return false;   // "Cannot transform object points to the plane z=0!"
        }
        // MRot * MCenter
        const double c[3] = {-xBar, -yBar, -zBar};
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) M2C[4 * r + k] = R[3 * r + k];
            M2C[4 * r + 3] = R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
        }
        return true;
    }

    // zero mean, mean squared distance 2: DataN (2 x n as two rows), T (back) and Ti (forth)
    static void normalizeDataIsotropic(const double* pts, int n, std::vector<double>& DataN, double T[9], double Ti[9]) {
        double xm = 0, ym = 0;
        for (int i = 0; i < n; i++) {
            xm += pts[2 * i];
            ym += pts[2 * i + 1];
        }
        xm = xm / (double)n;
        ym = ym / (double)n;
        double kappa = 0;
        DataN.resize(2 * (size_t)n);
        for (int i = 0; i < n; i++) {
            const double xh = pts[2 * i] - xm, yh = pts[2 * i + 1] - ym;
            DataN[i] = xh;
            DataN[n + i] = yh;
            kappa = kappa + xh * xh + yh * yh;
        }
        const double beta = std::sqrt(2 * n / kappa);
        for (size_t i = 0; i < DataN.size(); i++) DataN[i] = DataN[i] * beta;
        for (int i = 0; i < 9; i++) T[i] = Ti[i] = 0;
        T[0] = 1.0 / beta;
        T[4] = 1.0 / beta;
        T[2] = xm;
        T[5] = ym;
        T[8] = 1;
        Ti[0] = beta;
        Ti[4] = beta;
        Ti[2] = -beta * xm;
        Ti[5] = -beta * ym;
        Ti[8] = 1;
    }

    static void mul33(const double* A, const double* B, double* C) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
                C[3 * i + j] = s;
            }
    }

    // HomographyHO::homographyHO: src (n x 2, canonical object points) -> targ (n x 2, normalised image points)
    static void homographyHO(const double* src, const double* targ, int n, double H[9]) {
        std::vector<double> A, B;
        double TA[9], TAi[9], TB[9], TBi[9];
        normalizeDataIsotropic(src, n, A, TA, TAi);
        normalizeDataIsotropic(targ, n, B, TB, TBi);
        const double *Ax = &A[0], *Ay = &A[n], *Bx_ = &B[0], *By_ = &B[n];
        std::vector<double> C1(n), C2(n), C3(n), C4(n), Mx(3 * (size_t)n), My(3 * (size_t)n);
        double mC1 = 0, mC2 = 0, mC3 = 0, mC4 = 0;
        for (int i = 0; i < n; i++) {
            C1[i] = -Bx_[i] * Ax[i];
            C2[i] = -Bx_[i] * Ay[i];
            C3[i] = -By_[i] * Ax[i];
            C4[i] = -By_[i] * Ay[i];
            mC1 += C1[i];
            mC2 += C2[i];
            mC3 += C3[i];
            mC4 += C4[i];
        }
        mC1 /= n;
        mC2 /= n;
        mC3 /= n;
        mC4 /= n;
        for (int i = 0; i < n; i++) {
            Mx[3 * i] = C1[i] - mC1;
            Mx[3 * i + 1] = C2[i] - mC2;
            Mx[3 * i + 2] = -Bx_[i];
            My[3 * i] = C3[i] - mC3;
            My[3 * i + 1] = C4[i] - mC4;
            My[3 * i + 2] = -By_[i];
        }
        double g00 = 0, g01 = 0, g11 = 0;   // DataA * DataA'
        for (int i = 0; i < n; i++) {
            g00 += Ax[i] * Ax[i];
            g01 += Ax[i] * Ay[i];
            g11 += Ay[i] * Ay[i];
        }
        const double dt = g00 * g11 - g01 * g01;
        const double gi00 = g11 / dt, gi01 = -g01 / dt, gi10 = -g01 / dt, gi11 = g00 / dt;
        std::vector<double> Pp(2 * (size_t)n);   // DataADataATi * DataA
        for (int i = 0; i < n; i++) {
            Pp[i] = gi00 * Ax[i] + gi01 * Ay[i];
            Pp[n + i] = gi10 * Ax[i] + gi11 * Ay[i];
        }
        double Bx[6], By[6];   // 2 x 3: Pp * Mx, Pp * My
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) {
                double sx = 0, sy = 0;
                for (int i = 0; i < n; i++) {
                    sx += Pp[r * n + i] * Mx[3 * i + c];
                    sy += Pp[r * n + i] * My[3 * i + c];
                }
                Bx[3 * r + c] = sx;
                By[3 * r + c] = sy;
            }
        double DDT[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<double> D(6 * (size_t)n);   // rows 0..n-1: Mx - DataA' Bx, rows n..2n-1: My - DataA' By
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) {
                D[3 * i + c] = Mx[3 * i + c] - (Ax[i] * Bx[c] + Ay[i] * Bx[3 + c]);
                D[3 * (n + i) + c] = My[3 * i + c] - (Ax[i] * By[c] + Ay[i] * By[3 + c]);
            }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double s = 0;
                for (int i = 0; i < 2 * n; i++) s += D[3 * i + a] * D[3 * i + b];
                DDT[3 * a + b] = s;
            }
        double W[3], Ut[9], Vt[9];
        svd(DDT, 3, 3, W, Ut, Vt);
        const double h789[3] = {Vt[6], Vt[7], Vt[8]};
        double Hn[9];
        for (int r = 0; r < 2; r++) {
            Hn[r] = -(Bx[3 * r] * h789[0] + Bx[3 * r + 1] * h789[1] + Bx[3 * r + 2] * h789[2]);
            Hn[3 + r] = -(By[3 * r] * h789[0] + By[3 * r + 1] * h789[1] + By[3 * r + 2] * h789[2]);
        }
        Hn[2] = -(mC1 * h789[0] + mC2 * h789[1]);
        Hn[5] = -(mC3 * h789[0] + mC4 * h789[1]);
        Hn[6] = h789[0];
        Hn[7] = h789[1];
        Hn[8] = h789[2];
        double tmp[9];
        mul33(TB, Hn, tmp);
        mul33(tmp, TAi, H);
        const double h22_inv = 1 / H[8];
        for (int i = 0; i < 9; i++) H[i] = H[i] * h22_inv;
    }

    static void computeRotations(double j00, double j01, double j10, double j11, double p, double q, double R1[9], double R2[9]) {
        double Rv[9], RvT[9];
        const double v[3] = {p, q, 1};
        rotateVec2ZAxis(v, RvT);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rv[3 * i + j] = RvT[3 * j + i];
        const double rv00 = Rv[0], rv01 = Rv[1], rv02 = Rv[2], rv10 = Rv[3], rv11 = Rv[4], rv12 = Rv[5], rv20 = Rv[6], rv21 = Rv[7], rv22 = Rv[8];
        const double b00 = rv00 - p * rv20, b01 = rv01 - p * rv21, b10 = rv10 - q * rv20, b11 = rv11 - q * rv21;
        const double dtinv = 1.0 / ((b00 * b11 - b01 * b10));
        const double binv00 = dtinv * b11, binv01 = -dtinv * b01, binv10 = -dtinv * b10, binv11 = dtinv * b00;
        const double a00 = binv00 * j00 + binv01 * j10, a01 = binv00 * j01 + binv01 * j11, a10 = binv10 * j00 + binv11 * j10, a11 = binv10 * j01 + binv11 * j11;
        const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
        const double gamma2 = 0.5 * (ata00 + ata11 + std::sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
        const double gamma = std::sqrt(gamma2);
        const double rtilde00 = a00 / gamma, rtilde01 = a01 / gamma, rtilde10 = a10 / gamma, rtilde11 = a11 / gamma;
        const double rtilde00_2 = rtilde00 * rtilde00, rtilde01_2 = rtilde01 * rtilde01, rtilde10_2 = rtilde10 * rtilde10, rtilde11_2 = rtilde11 * rtilde11;
        double b0 = std::sqrt(std::max(0.0, -rtilde00_2 - rtilde10_2 + 1));
        double b1 = std::sqrt(std::max(0.0, -rtilde01_2 - rtilde11_2 + 1));
        const double sp = (-rtilde00 * rtilde01 - rtilde10 * rtilde11);
        if (sp < 0) b1 = -b1;
        const double c0 = b1 * rtilde10 - b0 * rtilde11, c1 = b0 * rtilde01 - b1 * rtilde00, c2 = rtilde00 * rtilde11 - rtilde01 * rtilde10;
        const double rvr[3][3] = {{rv00, rv01, rv02}, {rv10, rv11, rv12}, {rv20, rv21, rv22}};
        for (int r = 0; r < 3; r++) {
            R1[3 * r] = (rtilde00)*rvr[r][0] + (rtilde10)*rvr[r][1] + (b0)*rvr[r][2];
            R1[3 * r + 1] = (rtilde01)*rvr[r][0] + (rtilde11)*rvr[r][1] + (b1)*rvr[r][2];
            R1[3 * r + 2] = c0 * rvr[r][0] + c1 * rvr[r][1] + c2 * rvr[r][2];
            R2[3 * r] = (rtilde00)*rvr[r][0] + (rtilde10)*rvr[r][1] + (-b0) * rvr[r][2];
            R2[3 * r + 1] = (rtilde01)*rvr[r][0] + (rtilde11)*rvr[r][1] + (-b1) * rvr[r][2];
            R2[3 * r + 2] = (-c0) * rvr[r][0] + (-c1) * rvr[r][1] + c2 * rvr[r][2];
        }
    }

    // the translation that goes with R: normal equations of the 2n x 3 system
    static void computeTranslation(const double* canon, const double* nimg, int n, const double R[9], double t[3]) {
        const double ATA00 = (double)n, ATA11 = (double)n;
        double ATA02 = 0, ATA12 = 0, ATA22 = 0, ATb0 = 0, ATb1 = 0, ATb2 = 0;
        for (int i = 0; i < n; i++) {
            const double X = canon[2 * i], Y = canon[2 * i + 1];
            const double rx = R[0] * X + R[1] * Y, ry = R[3] * X + R[4] * Y, rz = R[6] * X + R[7] * Y;
            const double a2 = -nimg[2 * i], b2 = -nimg[2 * i + 1];
            ATA02 = ATA02 + a2;
            ATA12 = ATA12 + b2;
            ATA22 = ATA22 + (a2 * a2) + (b2 * b2);
            const double bx = -a2 * rz - rx, by = -b2 * rz - ry;
            ATb0 = ATb0 + bx;
            ATb1 = ATb1 + by;
            ATb2 = ATb2 + a2 * bx + b2 * by;
        }
        const double detAInv = 1.0 / (ATA00 * ATA11 * ATA22 - ATA00 * ATA12 * ATA12 - ATA02 * ATA02 * ATA11);
        const double S00 = ATA11 * ATA22 - ATA12 * ATA12, S01 = ATA02 * ATA12, S02 = -ATA02 * ATA11, S11 = ATA00 * ATA22 - ATA02 * ATA02, S12 = -ATA00 * ATA12,
                     S22 = ATA00 * ATA11;
        t[0] = detAInv * (S00 * ATb0 + S01 * ATb1 + S02 * ATb2);
        t[1] = detAInv * (S01 * ATb0 + S11 * ATb1 + S12 * ATb2);
        t[2] = detAInv * (S02 * ATb0 + S12 * ATb1 + S22 * ATb2);
    }
};

bool solve_pnp_ippe(const double* obj, const double* img, int n, const double* K, double* rvec, double* tvec) {
    if (n < 4) return false;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    std::vector<double> nimg(2 * (size_t)n), canon;
    for (int i = 0; i < n; i++) {
        nimg[2 * i] = (img[2 * i] - cx) * ifx;
        nimg[2 * i + 1] = (img[2 * i + 1] - cy) * ify;
    }
    double M2C[16], H[9];
    if (!Ippe::makeCanonicalObjectPoints(obj, n, canon, M2C)) return false;
    Ippe::homographyHO(canon.data(), nimg.data(), n, H);
    const double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double Rc[2][9], tc[2][3];
    Ippe::computeRotations(j00, j01, j10, j11, H[2], H[5], Rc[0], Rc[1]);
    double pose_R[2][9], pose_t[2][3], perr[2];
    for (int s = 0; s < 2; s++) {
        Ippe::computeTranslation(canon.data(), nimg.data(), n, Rc[s], tc[s]);
        // M = [Rc tc] * MmodelPoints2Canonical
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) pose_R[s][3 * r + c] = Rc[s][3 * r] * M2C[c] + Rc[s][3 * r + 1] * M2C[4 + c] + Rc[s][3 * r + 2] * M2C[8 + c];
            pose_t[s][r] = Rc[s][3 * r] * M2C[3] + Rc[s][3 * r + 1] * M2C[7] + Rc[s][3 * r + 2] * M2C[11] + tc[s][r];
        }
        double err = 0;   // the order solvePnPGeneric gives its solutions: reprojection error in pixels
        for (int i = 0; i < n; i++) {
            const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
            const double* R = pose_R[s];
            const double xc = R[0] * X + R[1] * Y + R[2] * Z + pose_t[s][0], yc = R[3] * X + R[4] * Y + R[5] * Z + pose_t[s][1],
                         zc = R[6] * X + R[7] * Y + R[8] * Z + pose_t[s][2];
            const double iz = 1. / zc, du = xc * iz * fx + cx - img[2 * i], dv = yc * iz * fy + cy - img[2 * i + 1];
            err += du * du + dv * dv;
        }
        perr[s] = err;
    }
    const int best = perr[1] < perr[0] ? 1 : 0;
    if (!(perr[best] == perr[best])) return false;   // NaN: a degenerate homography
    rodrigues_to_vector(pose_R[best], rvec);
    for (int k = 0; k < 3; k++) tvec[k] = pose_t[best][k];
    return true;
}

int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
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

bool next_subset(int count, int* idx, RNG& rng, int modelPoints = 5) {   // getSubset with the default checkSubset (always true)
    for (int i = 0; i < modelPoints; ++i) {
        int idx_i;
        for (idx_i = rng.uniform(0, count); std::find(idx, idx + i, idx_i) != idx + i; idx_i = rng.uniform(0, count)) {
        }
        idx[i] = idx_i;
    }
    return true;
}

}  // namespace

extern "C" {

void oracle_rodrigues(const double* in, int in_is_matrix, double* out) {
    if (in_is_matrix) rodrigues_to_vector(in, out);
    else rodrigues_to_matrix(in, out);
}

void oracle_det_acos(const double* c, int n, double* out) {
    for (int i = 0; i < n; i++) out[i] = det_acos(c[i]);
}

int oracle_solve_pnp_epnp(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec) {
    if (n < 4) return -215;
    solve_pnp_epnp<double>(obj_xyz, img_xy, n, K, rvec, tvec);
    return 1;
}

int oracle_pnp_ransac_samples(int n, int iters, int32_t* idx5) {
    RNG rng((uint64_t)-1);
    for (int it = 0; it < iters; it++) {
        int idx[5];
        next_subset(n, idx, rng);
        for (int j = 0; j < 5; j++) idx5[it * 5 + j] = idx[j];
    }
    return iters;
}

int oracle_pnp_hypothesis(const double* obj_xyz, const double* img_xy, const int32_t* idx5, const double* K, double* rvec, double* tvec) {
    float o[15], m[10];
    for (int j = 0; j < 5; j++) {
        for (int c = 0; c < 3; c++) o[3 * j + c] = (float)obj_xyz[3 * idx5[j] + c];
        for (int c = 0; c < 2; c++) m[2 * j + c] = (float)img_xy[2 * idx5[j] + c];
    }
    solve_pnp_epnp<float>(o, m, 5, K, rvec, tvec);
    return 1;
}

int oracle_pnp_p3p_hypothesis(const double* obj_xyz, const double* img_xy, const int32_t* idx4, const double* K, double* rvec, double* tvec) {
    float o[12], m[8];
    for (int j = 0; j < 4; j++) {
        for (int c = 0; c < 3; c++) o[3 * j + c] = (float)obj_xyz[3 * idx4[j] + c];
        for (int c = 0; c < 2; c++) m[2 * j + c] = (float)img_xy[2 * idx4[j] + c];
    }
    return solve_pnp_p3p<float>(o, m, K, rvec, tvec) ? 1 : 0;
}

int oracle_pnp_ap3p_hypothesis(const double* obj_xyz, const double* img_xy, const int32_t* idx4, const double* K, double* rvec, double* tvec) {
    float o[12], m[8];
    for (int j = 0; j < 4; j++) {
        for (int c = 0; c < 3; c++) o[3 * j + c] = (float)obj_xyz[3 * idx4[j] + c];
        for (int c = 0; c < 2; c++) m[2 * j + c] = (float)img_xy[2 * idx4[j] + c];
    }
    return solve_pnp_ap3p<float>(o, m, K, rvec, tvec) ? 1 : 0;
}

int oracle_pnp_ransac_samples4(int n, int iters, int32_t* idx4) {
    RNG rng((uint64_t)-1);
    for (int it = 0; it < iters; it++) {
        int idx[4];
        next_subset(n, idx, rng, 4);
        for (int j = 0; j < 4; j++) idx4[it * 4 + j] = idx[j];
    }
    return iters;
}

int oracle_solve_pnp_ransac(const double* obj_xyz, const double* img_xy, int n, const double* K, int iterations, float reproj_thr,
                            double confidence, int method, double* rvec, double* tvec, int32_t* inliers, int* n_inliers) {
    *n_inliers = 0;
    if (n < 4 || !obj_xyz || !img_xy || !K) return -215;          // CV_Assert(npoints >= 4 && ...)
    if (method == 3 /* SOLVEPNP_DLS */ || method == 4 /* SOLVEPNP_UPNP */) method = 1;   // solvePnPGeneric: "broken implementation", both run EPnP
    if (method != 0 /* SOLVEPNP_ITERATIVE */ && method != 1 /* SOLVEPNP_EPNP */ && method != 2 /* SOLVEPNP_P3P */ && method != 5 /* SOLVEPNP_AP3P */ &&
        method != 8 /* SOLVEPNP_SQPNP */ && method != 7 /* SOLVEPNP_IPPE_SQUARE */ && method != 6 /* SOLVEPNP_IPPE */)
        return -213;   // past the last member of cv::SolvePnPMethod
    // kernel choice of solvePnPRansac: P3P / AP3P on 4 points when asked for, P3P when there are only 4 points; EPnP on 5 otherwise
    const bool ap3p = method == 5;
    const bool p3p = method == 2 || method == 5 || n == 4;
    const int modelPoints = p3p ? 4 : 5;
    // Point3d / Point2d -> CV_32F (solvePnPRansac converts CV_64F inputs to float)
    std::vector<float> op(3 * (size_t)n), ip(2 * (size_t)n);
    for (size_t i = 0; i < op.size(); i++) op[i] = (float)obj_xyz[i];
    for (size_t i = 0; i < ip.size(); i++) ip[i] = (float)img_xy[i];
    std::vector<uint8_t> mask(n), bestMask(n);
    std::vector<float> err(n);
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    int maxGoodCount = 0;
    if (n == modelPoints) {   // one direct solve, every point an inlier
        if (p3p) {
            if (!(ap3p ? solve_pnp_ap3p<float>(op.data(), ip.data(), K, rvec, tvec) : solve_pnp_p3p<float>(op.data(), ip.data(), K, rvec, tvec))) return 0;
        } else {
            solve_pnp_epnp<float>(op.data(), ip.data(), n, K, rvec, tvec);
        }
        for (int i = 0; i < n; i++) inliers[i] = i;
        *n_inliers = n;
        return 1;
    }
    RNG rng((uint64_t)-1);
    int niters = std::max(iterations, 1);
    const float t = (float)((double)reproj_thr * (double)reproj_thr);
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        next_subset(n, idx, rng, modelPoints);
        float o[15], m[10];
        for (int j = 0; j < modelPoints; j++) {
            std::memcpy(&o[3 * j], &op[3 * (size_t)idx[j]], 12);
            std::memcpy(&m[2 * j], &ip[2 * (size_t)idx[j]], 8);
        }
        double r[3], tv[3];
        if (p3p) {
            if (!(ap3p ? solve_pnp_ap3p<float>(o, m, K, r, tv) : solve_pnp_p3p<float>(o, m, K, r, tv))) continue;   // runKernel returned 0 models
        } else {
            solve_pnp_epnp<float>(o, m, 5, K, r, tv);
        }
        pnp_errors(op.data(), ip.data(), n, K, r, tv, err.data());
        int goodCount = 0;
        for (int i = 0; i < n; i++) {
            const int f = err[i] <= t;
            mask[i] = (uint8_t)f;
            goodCount += f;
        }
        if (goodCount > std::max(maxGoodCount, modelPoints - 1)) {
            std::swap(mask, bestMask);
            std::memcpy(best_r, r, sizeof r);
            std::memcpy(best_t, tv, sizeof tv);
            maxGoodCount = goodCount;
            niters = update_num_iters(confidence, (double)(n - goodCount) / n, modelPoints, niters);
        }
    }
    if (maxGoodCount <= 0) return 0;
    // final pose: EPnP over the inliers (also when the RANSAC kernel was P3P), as doubles converted back from the float copies
    std::vector<double> oi, ii;
    int cnt = 0;
    for (int i = 0; i < n; i++)
        if (bestMask[i]) {
            for (int c = 0; c < 3; c++) oi.push_back((double)op[3 * (size_t)i + c]);
            for (int c = 0; c < 2; c++) ii.push_back((double)ip[2 * (size_t)i + c]);
            inliers[cnt++] = i;
        }
    if (method == 7) return -215;   // SOLVEPNP_IPPE_SQUARE: solvePnP's CV_Assert(npoints == 4) on >= 5 inliers, rethrown by solvePnPRansac
    if (method == 0) {
        // SOLVEPNP_ITERATIVE: solvePnP over the inliers WITHOUT an extrinsic guess (mod.rs:354): homography / DLT start, then the
        // Levenberg-Marquardt refinement. Five non-planar inliers cannot start the DLT: solvePnPRansac keeps the RANSAC model then.
        double param[6] = {best_r[0], best_r[1], best_r[2], best_t[0], best_t[1], best_t[2]};
        const Camera camK{K[0], K[4], K[2], K[5]};
        if (initial_pose_no_guess(oi.data(), ii.data(), cnt, camK, param)) refine_pose_lm(oi.data(), ii.data(), cnt, camK, param);
        std::memcpy(rvec, param, 3 * sizeof(double));
        std::memcpy(tvec, param + 3, 3 * sizeof(double));
    } else if (method == 6) {
        // SOLVEPNP_IPPE: EPnP stays the RANSAC kernel; the final solvePnP over the inliers is IPPE. Inliers that are not coplanar: its solver
        // throws inside solvePnPGeneric's try block, solvePnP finds nothing, solvePnPRansac hands back the RANSAC model and returns false.
        if (!solve_pnp_ippe(oi.data(), ii.data(), cnt, K, rvec, tvec)) {
            std::memcpy(rvec, best_r, sizeof best_r);
            std::memcpy(tvec, best_t, sizeof best_t);
            return 0;
        }
    } else if (method == 8) {
        // SOLVEPNP_SQPNP: EPnP stays the RANSAC kernel (solvepnp.cpp: only P3P / AP3P replace it), the final solvePnP over the inliers is SQPnP.
        // No solution: solvePnPRansac hands back the RANSAC model and returns false.
        if (!solve_pnp_sqpnp(oi.data(), ii.data(), cnt, K, rvec, tvec)) {
            std::memcpy(rvec, best_r, sizeof best_r);
            std::memcpy(tvec, best_t, sizeof best_t);
            return 0;
        }
    } else {
        solve_pnp_epnp<double>(oi.data(), ii.data(), cnt, K, rvec, tvec);
    }
    *n_inliers = cnt;
    return 1;
}

int oracle_solve_pnp_ippe(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec) {
    return solve_pnp_ippe(obj_xyz, img_xy, n, K, rvec, tvec) ? 1 : 0;
}

int oracle_solve_pnp_sqpnp(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec) {
    return solve_pnp_sqpnp(obj_xyz, img_xy, n, K, rvec, tvec) ? 1 : 0;
}

}  // extern "C"
