// csrc/sqpnp_core.h — the final pose of pnp_solver_ransac when the caller names SOLVEPNP_SQPNP.
//
// Reference call site: homographier/src/homographier/mod.rs:327,359 hand Option<SolvePnPMethod> to cv::solvePnPRansac; with
// SOLVEPNP_SQPNP the RANSAC kernel stays EPnP on 5 points and the last solvePnP over the inliers is calib3d/sqpnp.cpp
// (Terzakis & Lourakis, ECCV 2020; OpenCV >= 4.7 form: FOAM nearest rotation, majority cheirality test). One 9 x 9 problem per
// call whatever the number of inliers (the points only enter through 51 sums), so it runs on the host like the other final
// refits (pnp.hip). Arithmetic contract shared with oracle/pnp_oracle.cpp (a separate text): IEEE double, -ffp-contract=off,
// every sum in the order written here. PARITY UNPINNED against OpenCV.
//
// Layout: a 9-vector r is a row-major 3 x 3 rotation; matrices are flat row-major arrays.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "pnp_core.h"

namespace apds {
namespace sqpnp {

struct Candidate {
    double rot[9];    // r_hat
    double trans[3];
    double cost;      // r_hat' Omega r_hat
};

struct Problem {
    double omega[81];   // 9 x 9
    double sv[9];       // its singular values, descending
    double vec[81];     // vec[i * 9 + k]: component k of the i-th singular vector
    double P[27];       // 3 x 9: t = P r
    double centroid[3];
    const double* obj;
    int n;
    int null_dim;
    Candidate kept[18];
    int n_kept;
};

inline double sum3(double a, double b, double c) { return a + b + c; }
inline double sq3(const double* v) { return sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2]); }
inline double dot3at(const double* a, const double* b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline double det9(const double* e) {
    return e[0] * e[4] * e[8] + e[1] * e[5] * e[6] + e[2] * e[3] * e[7] - e[6] * e[4] * e[2] - e[7] * e[5] * e[0] - e[8] * e[3] * e[1];
}

// how far the rows of e are from orthonormal (squared)
inline double rows_orthonormality_defect(const double* e) {
    const double n1 = sq3(e), n2 = sq3(e + 3), n3 = sq3(e + 6);
    const double d12 = dot3at(e, e + 3), d13 = dot3at(e, e + 6), d23 = dot3at(e + 3, e + 6);
    return (n1 - 1) * (n1 - 1) + (n2 - 1) * (n2 - 1) + (n3 - 1) * (n3 - 1) + 2 * (d12 * d12 + d13 * d13 + d23 * d23);
}

// inverse of a symmetric 3 x 3 (row-major S): cofactors over the determinant; through the SVD when the determinant is tiny
inline void inv_sym3(const double* S, double* out) {
    const double a = S[0], b = S[3], d = S[4], c = S[6], e = S[7], f = S[8];
    const double ee = e * e, ad = a * d, bb = b * b, bc = b * c, cc = c * c;
    const double det = -ad * f + a * ee + bb * f - 2.0 * bc * e + cc * d;
    if (std::fabs(det) < 1e-8) {   // cv::invert(DECOMP_SVD): sum over w_i above 2 eps sum(w)
        double W[3], Ut[9], Vt[9];
        pnp::svd3(S, W, Ut, Vt);
        const double cut = (W[0] + W[1] + W[2]) * (DBL_EPSILON * 2);
        for (int j = 0; j < 9; j++) out[j] = 0;
        for (int i = 0; i < 3; i++) {
            if (std::fabs(W[i]) <= cut) continue;
            const double iw = 1 / W[i];
            const double scaled[3] = {Ut[i * 3] * iw, Ut[i * 3 + 1] * iw, Ut[i * 3 + 2] * iw};
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) out[j * 3 + k] = out[j * 3 + k] + Vt[i * 3 + j] * scaled[k];
        }
        return;
    }
    const double id = 1.0 / det, m01 = (-b * f + c * e) * id, m02 = (b * e - c * d) * id, m12 = (a * e - bc) * id;
    out[0] = (-d * f + ee) * id;
    out[1] = out[3] = -m01;
    out[2] = out[6] = -m02;
    out[4] = -(a * f - cc) * id;
    out[5] = out[7] = m12;
    out[8] = -(ad - bb) * id;
}

// the rotation nearest to e (FOAM: Newton on the quartic whose largest root is the trace of the polar factor's stretch)
inline void nearest_rotation(const double* e, double* r) {
    const double adj[9] = {e[4] * e[8] - e[5] * e[7], e[2] * e[7] - e[1] * e[8], e[1] * e[5] - e[2] * e[4],
                           e[5] * e[6] - e[3] * e[8], e[0] * e[8] - e[2] * e[6], e[2] * e[3] - e[0] * e[5],
                           e[3] * e[7] - e[4] * e[6], e[1] * e[6] - e[0] * e[7], e[0] * e[4] - e[1] * e[3]};
    const double det = e[0] * e[4] * e[8] - e[0] * e[5] * e[7] - e[1] * e[3] * e[8] + e[2] * e[3] * e[7] + e[1] * e[6] * e[5] - e[2] * e[6] * e[4];
    double ee = 0, aa = 0;
    for (int i = 0; i < 9; i++) ee += e[i] * e[i];
    for (int i = 0; i < 9; i++) aa += adj[i] * adj[i];
    double lam = 0.5 * (ee + 3.0), before = 0.0;
    if (det < 0.0) lam = -lam;
    for (int left = 15; std::fabs(lam - before) > 1E-12 * std::fabs(before) && left > 0; --left) {
        const double q = lam * lam - ee;
        const double poly = q * q - 8.0 * lam * det - 4.0 * aa;
        const double slope = 8.0 * (0.5 * q * lam - det);
        before = lam;
        lam -= poly / slope;
    }
    const double scale_e = lam * lam + ee;
    double gram[9], cubic[9];   // e e', then (e e') e
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) gram[3 * i + j] = dot3at(e + 3 * i, e + 3 * j);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) cubic[3 * i + j] = sum3(gram[3 * i] * e[j], gram[3 * i + 1] * e[3 + j], gram[3 * i + 2] * e[6 + j]);
    const double den = lam * (lam * lam - ee) - 2.0 * det;
    // DOCUMENTED DEVIATION: den is (s1 + s2)(s1 + s3)(s2 + s3) of e's singular values; for a rank-one e (what every null vector of Omega is
    // when the object points are coplanar: w n') it vanishes and the formula below returns e / lam, which is no rotation - the descent that
    // starts there divides by zero. Such a matrix is completed to a rotation through its SVD, as OpenCV <= 4.6 did for every e.
    if (!(std::fabs(den) >= 1e-3 * (ee * std::sqrt(ee)))) {
        double W[3], Ut[9], Vt[9];
        pnp::svd3(e, W, Ut, Vt);
        const double flip = det9(Ut) * det9(Vt);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r[3 * i + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + (Ut[6 + i] * flip) * Vt[6 + j];
        return;
    }
    const double inv_den = 1.0 / den;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[3 * i + j] = (scale_e * e[3 * i + j] + 2.0 * (lam * adj[3 * j + i] - cubic[3 * i + j])) * inv_den;
}

// Omega, P, the singular vectors of Omega and the rank test, from object points and NORMALISED image points
inline void build(Problem& pb, const double* obj, const double* nimg, int n) {
    pb.obj = obj;
    pb.n = n;
    pb.n_kept = 0;
    double* om = pb.omega;
    std::memset(om, 0, sizeof pb.omega);
    double QA[27];   // 3 x 9: sum of Q_i A_i
    std::memset(QA, 0, sizeof QA);
    double sx = 0, sy = 0, sq = 0, so[3] = {0, 0, 0};
    // the blocks of sum A_i' Q_i A_i are multiples of X X': (0,0) = (1,1) = +1, (0,2) = -x, (1,2) = -y, (2,2) = x^2 + y^2
    for (int i = 0; i < n; i++) {
        const double* Xp = obj + 3 * (size_t)i;
        const double x = nimg[2 * (size_t)i], y = nimg[2 * (size_t)i + 1], w = x * x + y * y;
        sq += w;
        sx += x;
        sy += y;
        for (int k = 0; k < 3; k++) so[k] += Xp[k];
        const double outer[6] = {Xp[0] * Xp[0], Xp[0] * Xp[1], Xp[0] * Xp[2], Xp[1] * Xp[1], Xp[1] * Xp[2], Xp[2] * Xp[2]};   // upper triangle of X X'
        static const int ur[6] = {0, 0, 0, 1, 1, 2}, uc[6] = {0, 1, 2, 1, 2, 2};
        for (int u = 0; u < 6; u++) {
            om[ur[u] * 9 + uc[u]] += outer[u];
            om[ur[u] * 9 + 6 + uc[u]] += -x * outer[u];
            om[(3 + ur[u]) * 9 + 6 + uc[u]] += -y * outer[u];
            om[(6 + ur[u]) * 9 + 6 + uc[u]] += w * outer[u];
        }
        for (int k = 0; k < 3; k++) {
            QA[k] += Xp[k];
            QA[9 + 3 + k] += Xp[k];
            QA[6 + k] += -x * Xp[k];
            QA[9 + 6 + k] += -y * Xp[k];
            QA[18 + k] += -x * Xp[k];
            QA[18 + 3 + k] += -y * Xp[k];
            QA[18 + 6 + k] += w * Xp[k];
        }
    }
    // the strictly lower parts of the three off-diagonal / last blocks, block (1,1) = block (0,0), then the lower triangle
    for (int blk = 0; blk < 3; blk++) {
        double* B = om + blk * 3 * 9 + 6;
        B[9] = B[1];
        B[18] = B[2];
        B[18 + 1] = B[9 + 2];
    }
    for (int u = 0; u < 3; u++)
        for (int v = u; v < 3; v++) om[(3 + u) * 9 + 3 + v] = om[u * 9 + v];
    for (int rr = 0; rr < 9; rr++)
        for (int cc = 0; cc < rr; cc++) om[rr * 9 + cc] = om[cc * 9 + rr];
    const double Q[9] = {(double)n, 0, -sx, 0, (double)n, -sy, -sx, -sy, sq};
    double Qi[9];
    inv_sym3(Q, Qi);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += -Qi[i * 3 + k] * QA[k * 9 + j];
            pb.P[i * 9 + j] = s;
        }
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += QA[k * 9 + i] * pb.P[k * 9 + j];
            om[i * 9 + j] += s;
        }
    double At[81];
    for (int i = 0; i < 9; i++)
        for (int k = 0; k < 9; k++) At[i * 9 + k] = om[k * 9 + i];
    pnp::svd_rows<true>(pnp::Plain<double>{At}, 9, 9, pnp::Plain<double>{pb.sv}, pnp::Plain<double>{pb.vec});
    int small = 0;   // singular values below the rank tolerance, counted from the smallest
    while (small < 9 && pb.sv[8 - small] < 1e-7) small++;
    pb.null_dim = small;
    const double inv_n = 1.0 / n;
    for (int k = 0; k < 3; k++) pb.centroid[k] = so[k] * inv_n;
}

// H (9 x 6, flat, row-major): orthonormal basis of the row space of the six constraints' Jacobian at r; L = J H (6 x 6 lower
// triangular); Nb (9 x 3): an orthonormal basis of the Jacobian's null space
inline void constraint_spaces(const double* r, double* H, double* Nb, double* L) {
    std::memset(H, 0, 54 * sizeof(double));
    std::memset(L, 0, 36 * sizeof(double));
    auto h = [&](int row, int col) -> double& { return H[row * 6 + col]; };
    auto hdot = [&](int r0, int h0, int col) { return sum3(r[r0] * h(h0, col), r[r0 + 1] * h(h0 + 1, col), r[r0 + 2] * h(h0 + 2, col)); };
    auto normalise_col = [&](int col, int rows) {
        double s = 0;
        for (int i = 0; i < rows; i++) s += h(i, col) * h(i, col);
        const double inv = 1.0 / std::sqrt(s);
        for (int i = 0; i < rows; i++) h(i, col) *= inv;
    };
    // columns 0..2: the rows of r, each in its own block
    const double len1 = std::sqrt(sq3(r)), len2 = std::sqrt(sq3(r + 3)), len3 = std::sqrt(sq3(r + 6));
    const double il1 = len1 > 1e-5 ? 1.0 / len1 : 0.0, il2 = 1.0 / len2, il3 = 1.0 / len3;
    for (int k = 0; k < 3; k++) {
        h(k, 0) = r[k] * il1;
        h(3 + k, 1) = r[3 + k] * il2;
        h(6 + k, 2) = r[6 + k] * il3;
    }
    L[0] = 2 * len1;
    L[6 + 1] = 2 * len2;
    L[12 + 2] = 2 * len3;
    // column 3: the gradient of r1.r2 = (r2, r1, 0), made orthogonal to columns 0 and 1
    const double c30 = hdot(3, 0, 0), c31 = hdot(0, 3, 1);
    for (int k = 0; k < 3; k++) {
        h(k, 3) = r[3 + k] - c30 * h(k, 0);
        h(3 + k, 3) = r[k] - c31 * h(3 + k, 1);
    }
    {
        const double inv = 1.0 / std::sqrt(h(0, 3) * h(0, 3) + h(1, 3) * h(1, 3) + h(2, 3) * h(2, 3) + h(3, 3) * h(3, 3) + h(4, 3) * h(4, 3) + h(5, 3) * h(5, 3));
        for (int i = 0; i < 6; i++) h(i, 3) *= inv;
    }
    L[18 + 0] = hdot(3, 0, 0);
    L[18 + 1] = hdot(0, 3, 1);
    L[18 + 3] = h(0, 3) * r[3] + h(1, 3) * r[4] + h(2, 3) * r[5] + h(3, 3) * r[0] + h(4, 3) * r[1] + h(5, 3) * r[2];
    // column 4: the gradient of r2.r3 = (0, r3, r2), against columns 1, 2, 3
    const double c41 = hdot(6, 3, 1), c42 = hdot(3, 6, 2), c43 = hdot(6, 3, 3);
    for (int k = 0; k < 3; k++) {
        h(k, 4) = -c43 * h(k, 3);
        h(3 + k, 4) = r[6 + k] - c41 * h(3 + k, 1) - c43 * h(3 + k, 3);
        h(6 + k, 4) = r[3 + k] - c42 * h(6 + k, 2);
    }
    normalise_col(4, 9);
    L[24 + 1] = hdot(6, 3, 1);
    L[24 + 2] = hdot(3, 6, 2);
    L[24 + 3] = hdot(6, 3, 3);
    L[24 + 4] = h(3, 4) * r[6] + h(4, 4) * r[7] + h(5, 4) * r[8] + h(6, 4) * r[3] + h(7, 4) * r[4] + h(8, 4) * r[5];
    // column 5: the gradient of r1.r3 = (r3, 0, r1), against columns 0, 2, 3, 4
    const double c50 = hdot(6, 0, 0), c52 = hdot(0, 6, 2), c53 = hdot(6, 0, 3);
    const double c54 = h(6, 4) * r[0] + h(7, 4) * r[1] + h(8, 4) * r[2] + h(0, 4) * r[6] + h(1, 4) * r[7] + h(2, 4) * r[8];
    for (int k = 0; k < 3; k++) {
        h(k, 5) = r[6 + k] - c50 * h(k, 0) - c53 * h(k, 3) - c54 * h(k, 4);
        h(3 + k, 5) = -c54 * h(3 + k, 4) - c53 * h(3 + k, 3);
        h(6 + k, 5) = r[k] - c52 * h(6 + k, 2) - c54 * h(6 + k, 4);
    }
    normalise_col(5, 9);
    L[30 + 0] = hdot(6, 0, 0);
    L[30 + 2] = hdot(0, 6, 2);
    L[30 + 3] = hdot(6, 0, 3);
    L[30 + 4] = h(0, 4) * r[6] + h(1, 4) * r[7] + h(2, 4) * r[8] + h(6, 4) * r[0] + h(7, 4) * r[1] + h(8, 4) * r[2];
    L[30 + 5] = h(0, 5) * r[6] + h(1, 5) * r[7] + h(2, 5) * r[8] + h(6, 5) * r[0] + h(7, 5) * r[1] + h(8, 5) * r[2];
    // I - H H': three of its columns span the null space (the longest, the one most orthogonal to it, then to both)
    double proj[81], len[9];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += h(i, k) * h(j, k);
            proj[i * 9 + j] = (i == j ? 1.0 : 0.0) - s;
        }
    auto cdot = [&](int a, int b) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += proj[k * 9 + a] * proj[k * 9 + b];
        return s;
    };
    const double floor_len = 0.1;
    int first = 0, second = 0, third = 0;
    double longest = DBL_MIN, least12 = DBL_MAX, least3 = DBL_MAX;
    for (int i = 0; i < 9; i++) {
        len[i] = std::sqrt(cdot(i, i));
        if (len[i] >= floor_len && longest < len[i]) {
            longest = len[i];
            first = i;
        }
    }
    for (int k = 0; k < 9; k++) Nb[k * 3] = proj[k * 9 + first] * (1.0 / longest);
    for (int i = 0; i < 9; i++) {
        if (i == first || len[i] < floor_len) continue;
        const double c1 = std::fabs(cdot(i, first) / len[i]);
        if (c1 <= least12) {
            second = i;
            least12 = c1;
        }
    }
    auto orthonormalise = [&](int col, int src) {   // column `col` of Nb from column `src` of proj, against the columns before it (last first)
        double coef[2];
        for (int p = col - 1; p >= 0; p--) {
            double d = 0;
            for (int k = 0; k < 9; k++) d += proj[k * 9 + src] * Nb[k * 3 + p];
            coef[p] = d;
        }
        for (int k = 0; k < 9; k++) {
            double v = proj[k * 9 + src];
            for (int p = col - 1; p >= 0; p--) v = v - coef[p] * Nb[k * 3 + p];
            Nb[k * 3 + col] = v;
        }
        double s = 0;
        for (int k = 0; k < 9; k++) s += Nb[k * 3 + col] * Nb[k * 3 + col];
        const double inv = 1.0 / std::sqrt(s);
        for (int k = 0; k < 9; k++) Nb[k * 3 + col] *= inv;
    };
    orthonormalise(1, second);
    for (int i = 0; i < 9; i++) {
        if (i == second || i == first || len[i] < floor_len) continue;
        const double c1 = std::fabs(cdot(i, first) / len[i]), c2 = std::fabs(cdot(i, second) / len[i]);
        if (c1 + c2 <= least3) {
            third = i;
            least3 = c2 + c2;   // sqpnp.cpp keeps twice the second cosine here, not the sum it compared
        }
    }
    orthonormalise(2, third);
}

// one SQP step: step = H x + Nb y with (J H) x = g (the constraint residuals) and y minimising the quadratic in the tangent space
inline void sqp_step(const Problem& pb, const double* r, double* step) {
    double H[54], Nb[27], L[36];
    const double g[6] = {1 - sq3(r), 1 - sq3(r + 3), 1 - sq3(r + 6), -dot3at(r, r + 3), -dot3at(r + 3, r + 6), -dot3at(r, r + 6)};
    constraint_spaces(r, H, Nb, L);
    double x[6];
    x[0] = g[0] / L[0];
    x[1] = g[1] / L[7];
    x[2] = g[2] / L[14];
    x[3] = (g[3] - L[18] * x[0] - L[19] * x[1]) / L[21];
    x[4] = (g[4] - L[25] * x[1] - L[26] * x[2] - L[27] * x[3]) / L[28];
    x[5] = (g[5] - L[30] * x[0] - L[32] * x[2] - L[33] * x[3] - L[34] * x[4]) / L[35];
    for (int i = 0; i < 9; i++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += H[i * 6 + k] * x[k];
        step[i] = s;
    }
    double NO[27], Wm[9], Wi[9], M[27];   // Nb' Omega, Nb' Omega Nb, its inverse, -Wi * NO
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += Nb[k * 3 + i] * pb.omega[k * 9 + j];
            NO[i * 9 + j] = s;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += NO[i * 9 + k] * Nb[k * 3 + j];
            Wm[i * 3 + j] = s;
        }
    inv_sym3(Wm, Wi);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += -Wi[i * 3 + k] * NO[k * 9 + j];
            M[i * 9 + j] = s;
        }
    double y[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += M[i * 9 + k] * (step[k] + r[k]);
        y[i] = s;
    }
    for (int i = 0; i < 9; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += Nb[i * 3 + k] * y[k];
        step[i] += s;
    }
}

// SQP from a rotation; the result's sign makes det > 0, and a result that drifted off SO(3) is pulled back
inline void descend(const Problem& pb, const double* start, double* r_hat) {
    double r[9], step[9];
    std::memcpy(r, start, sizeof r);
    double moved = DBL_MAX;
    for (int it = 0; moved > 1e-10 && it < 15; it++) {
        sqp_step(pb, r, step);
        moved = 0;
        for (int i = 0; i < 9; i++) r[i] += step[i];
        for (int i = 0; i < 9; i++) moved += step[i] * step[i];
    }
    double d = det9(r);
    if (d < 0) {
        for (int i = 0; i < 9; i++) r[i] = -r[i];
        d = -d;
    }
    if (d > 1.001)
        nearest_rotation(r, r_hat);
    else
        std::memcpy(r_hat, r, sizeof r);
}

// translation, cheirality (centroid, else the majority of the points), cost, and the bookkeeping of equal-cost minima
inline void consider(Problem& pb, Candidate& cand, double& least) {
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += pb.P[i * 9 + k] * cand.rot[k];
        cand.trans[i] = s;
    }
    const double* r = cand.rot;
    // DOCUMENTED DEVIATION (a guard sqpnp.cpp does not have): descend() returns its iterate unprojected when det <= 1.001, and a start inside
    // Omega's null space (coplanar object points) can come back near zero with cost 0 and a depth test decided by rounding; accepted, it
    // ends the search. What is not a rotation is not a candidate.
    if (!(rows_orthonormality_defect(r) <= 0.1)) return;
    bool in_front = r[6] * pb.centroid[0] + r[7] * pb.centroid[1] + r[8] * pb.centroid[2] + cand.trans[2] > 0;
    if (!in_front) {
        int front = 0;
        for (int i = 0; i < pb.n; i++) front += r[6] * pb.obj[3 * (size_t)i] + r[7] * pb.obj[3 * (size_t)i + 1] + r[8] * pb.obj[3 * (size_t)i + 2] + cand.trans[2] > 0;
        in_front = front >= pb.n - front;
    }
    if (!in_front) return;
    double cost = 0;
    for (int i = 0; i < 9; i++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += pb.omega[i * 9 + k] * r[k];
        cost += s * r[i];
    }
    cand.cost = cost;
    if (std::fabs(least - cost) > 1e-6) {
        if (least > cost) {
            least = cost;
            pb.kept[0] = cand;
            pb.n_kept = 1;
        }
        return;
    }
    int same = -1;
    for (int i = 0; i < pb.n_kept && same < 0; i++) {
        double d = 0;
        for (int k = 0; k < 9; k++) d += (pb.kept[i].rot[k] - r[k]) * (pb.kept[i].rot[k] - r[k]);
        if (d < 1e-10) same = i;
    }
    if (same < 0)
        pb.kept[pb.n_kept++] = cand;
    else if (pb.kept[same].cost > cost)
        pb.kept[same] = cand;
    if (least > cost) least = cost;
}

inline void search(Problem& pb) {
    double least = DBL_MAX;
    const int starts = pb.null_dim > 0 ? pb.null_dim : 1;
    const double root3 = std::sqrt(3.0);
    auto both_signs = [&](const double* e) {
        double flipped[9], start[9];
        Candidate c{};
        for (int k = 0; k < 9; k++) flipped[k] = -e[k];
        nearest_rotation(e, start);
        descend(pb, start, c.rot);
        consider(pb, c, least);
        nearest_rotation(flipped, start);
        descend(pb, start, c.rot);
        consider(pb, c, least);
    };
    for (int i = 9 - starts; i < 9; i++) {
        double e[9];
        for (int k = 0; k < 9; k++) e[k] = root3 * pb.vec[i * 9 + k];
        if (rows_orthonormality_defect(e) < 1e-8) {
            Candidate c{};
            const double d = det9(e);
            for (int k = 0; k < 9; k++) c.rot[k] = d * e[k];
            consider(pb, c, least);
        } else {
            both_signs(e);
        }
    }
    for (int back = 1, i; (i = 9 - starts - back) > 0 && least > 3 * pb.sv[i]; back++) both_signs(pb.vec + i * 9);
}

// solvePnP(SOLVEPNP_SQPNP) on double points; false where OpenCV throws (degenerate sets) or finds no pose in front of the camera
inline bool solve(const double* obj, const double* img, int n, const pnp::Camera& cam, double* rvec, double* tvec) {
    if (n < 3) return false;
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
    std::vector<double> nimg(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        nimg[2 * (size_t)i] = (img[2 * (size_t)i] - cam.uc) * ifx;
        nimg[2 * (size_t)i + 1] = (img[2 * (size_t)i + 1] - cam.vc) * ify;
    }
    Problem pb;
    build(pb, obj, nimg.data(), n);
    if (!(pb.sv[0] >= 1e-7) || pb.null_dim > 6) return false;
    search(pb);
    if (pb.n_kept <= 0) return false;
    int pick = 0;
    double pick_err = DBL_MAX;
    for (int c = 0; c < pb.n_kept; c++) {   // several minima: the one that reprojects best
        const double* R = pb.kept[c].rot;
        const double* t = pb.kept[c].trans;
        double err = 0;
        for (int i = 0; i < n; i++) {
            const double* X = obj + 3 * (size_t)i;
            const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                         zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
            const double iz = 1. / zc, du = xc * iz * cam.fu + cam.uc - img[2 * (size_t)i], dv = yc * iz * cam.fv + cam.vc - img[2 * (size_t)i + 1];
            err += du * du + dv * dv;
        }
        if (err < pick_err) {
            pick_err = err;
            pick = c;
        }
    }
    pnp::rvec_from_rotation(pb.kept[pick].rot, rvec);
    for (int k = 0; k < 3; k++) tvec[k] = pb.kept[pick].trans[k];
    return true;
}

}  // namespace sqpnp
}  // namespace apds
