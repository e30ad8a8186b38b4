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

bool next_subset(int count, int* idx, RNG& rng) {   // getSubset with the default checkSubset (always true)
    const int modelPoints = 5;
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

int oracle_solve_pnp_ransac(const double* obj_xyz, const double* img_xy, int n, const double* K, int iterations, float reproj_thr,
                            double confidence, int method, double* rvec, double* tvec, int32_t* inliers, int* n_inliers) {
    *n_inliers = 0;
    if (n < 4 || !obj_xyz || !img_xy || !K) return -215;          // CV_Assert(npoints >= 4 && ...)
    if (method != 1 /* SOLVEPNP_EPNP */ || n == 4) return -213;    // P3P / AP3P kernels (and the n == 4 shortcut through P3P): not restated
    // Point3d / Point2d -> CV_32F (solvePnPRansac converts CV_64F inputs to float)
    std::vector<float> op(3 * (size_t)n), ip(2 * (size_t)n);
    for (size_t i = 0; i < op.size(); i++) op[i] = (float)obj_xyz[i];
    for (size_t i = 0; i < ip.size(); i++) ip[i] = (float)img_xy[i];
    const int modelPoints = 5;
    std::vector<uint8_t> mask(n), bestMask(n);
    std::vector<float> err(n);
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    int maxGoodCount = 0;
    if (n == modelPoints) {
        solve_pnp_epnp<float>(op.data(), ip.data(), n, K, rvec, tvec);
        for (int i = 0; i < n; i++) inliers[i] = i;
        *n_inliers = n;
        return 1;
    }
    RNG rng((uint64_t)-1);
    int niters = std::max(iterations, 1);
    const float t = (float)((double)reproj_thr * (double)reproj_thr);
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        next_subset(n, idx, rng);
        float o[15], m[10];
        for (int j = 0; j < 5; j++) {
            std::memcpy(&o[3 * j], &op[3 * (size_t)idx[j]], 12);
            std::memcpy(&m[2 * j], &ip[2 * (size_t)idx[j]], 8);
        }
        double r[3], tv[3];
        solve_pnp_epnp<float>(o, m, 5, K, r, tv);
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
    // final EPnP over the inliers, as doubles converted back from the float copies
    std::vector<double> oi, ii;
    int cnt = 0;
    for (int i = 0; i < n; i++)
        if (bestMask[i]) {
            for (int c = 0; c < 3; c++) oi.push_back((double)op[3 * (size_t)i + c]);
            for (int c = 0; c < 2; c++) ii.push_back((double)ip[2 * (size_t)i + c]);
            inliers[cnt++] = i;
        }
    solve_pnp_epnp<double>(oi.data(), ii.data(), cnt, K, rvec, tvec);
    *n_inliers = cnt;
    return 1;
}

}  // extern "C"
