// csrc/ippe_core.h — the final pose of pnp_solver_ransac when the caller names SOLVEPNP_IPPE.
//
// Reference call site: homographier/src/homographier/mod.rs:327,359 (Option<SolvePnPMethod> handed to cv::solvePnPRansac). With
// SOLVEPNP_IPPE the RANSAC kernel stays EPnP on 5 points and the last solvePnP over the inliers is calib3d/ippe.cpp (Collins & Bartoli's
// infinitesimal plane-based pose estimation): a pose from a PLANAR target. Object points that are not coplanar within 1e-3 (in their own
// unit) make OpenCV's solver throw inside solvePnPGeneric's try block - no solution, solvePnPRansac returns false; solve() below returns
// false there. Host arithmetic (a handful of sums over the inliers, then 3 x 3 algebra), like the other final refits in pnp.hip.
// Arithmetic contract shared with oracle/pnp_oracle.cpp (a separate text): IEEE double, -ffp-contract=off, sums in the order written.
// PARITY UNPINNED against OpenCV; the two documented deviations (SVD instead of cv::eigen for a 3 x 3 null vector, clamped square roots)
// are listed in the oracle's header and DESIGN.md section 2.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "pnp_core.h"

namespace apds {
namespace ippe {

constexpr double kFlat = 1e-3;   // IPPE_SMALL

// the rotation that turns direction a onto +z (a need not be a unit vector)
inline void turn_to_z(double a0, double a1, double a2, double* M) {
    const double len = std::sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    const double x = a0 / len, y = a1 / len, z = a2 / len;
    if (std::fabs(1.0 + z) < (double)FLT_EPSILON) {
        const double flip[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1};   // half a turn about x (diag(1, 1, -1), a reflection, is what memory of ippe.cpp says: GUESSED)
        std::memcpy(M, flip, sizeof flip);
        return;
    }
    const double k = 1.0 / (1.0 + z), xx = x * x, yy = y * y, xy = x * y;
    M[0] = -xx * k + 1.0;
    M[1] = -xy * k;
    M[2] = -x;
    M[3] = -xy * k;
    M[4] = -yy * k + 1.0;
    M[5] = -y;
    M[6] = x;
    M[7] = y;
    M[8] = 1.0 - (xx + yy) * k;
}

struct PlaneFrame {
    double rot[9];      // model -> plane frame
    double shift[3];    // rot * (-centroid)
    std::vector<double> uv;   // n x 2: the points in the plane, about their centroid
};

// move the object points to the plane z = 0 about their centroid; false when they are not coplanar
inline bool to_plane(const double* obj, int n, PlaneFrame& f) {
    double sum[3] = {0, 0, 0};
    bool already_flat = true;
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) sum[k] += obj[3 * (size_t)i + k];
        if (std::fabs(obj[3 * (size_t)i + 2]) > kFlat) already_flat = false;
    }
    const double mean[3] = {sum[0] / n, sum[1] / n, sum[2] / n};
    std::vector<double> centred(3 * (size_t)n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) centred[3 * (size_t)i + k] = obj[3 * (size_t)i + k] - mean[k];
    f.uv.resize(2 * (size_t)n);
    const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(f.rot, ident, sizeof ident);
    if (already_flat) {
        for (int k = 0; k < 3; k++) f.shift[k] = -mean[k];
        for (int i = 0; i < n; i++) {
            f.uv[2 * (size_t)i] = centred[3 * (size_t)i];
            f.uv[2 * (size_t)i + 1] = centred[3 * (size_t)i + 1];
        }
        return true;
    }
    // the plane's normal: the first three points, or - when they are (nearly) collinear - the weakest direction of the scatter matrix
    const double *a = obj, *b = obj + 3, *c = obj + 6;
    const double nx = (a[1] - b[1]) * (a[2] - c[2]) - (a[1] - c[1]) * (a[2] - b[2]);
    const double ny = (a[0] - c[0]) * (a[2] - b[2]) - (a[0] - b[0]) * (a[2] - c[2]);
    const double nz = (a[0] - b[0]) * (a[1] - c[1]) - (a[0] - c[0]) * (a[1] - b[1]);
    const double nlen = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (nlen > kFlat) {
        turn_to_z(nx / nlen, ny / nlen, nz / nlen, f.rot);
    } else {
        double scatter[9], W[3], Ut[9], Vt[9];
        for (int p = 0; p < 3; p++)
            for (int q = 0; q < 3; q++) {
                double s = 0;
                for (int i = 0; i < n; i++) s += centred[3 * (size_t)i + p] * centred[3 * (size_t)i + q];
                scatter[3 * p + q] = s;
            }
        pnp::svd3(scatter, W, Ut, Vt);
        if (!(W[2] / W[1] < kFlat)) return false;
        std::memcpy(f.rot, Ut, sizeof f.rot);   // rows = left singular vectors: U'
        const double* R = f.rot;
        const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (det < 0)
            for (int k = 6; k < 9; k++) f.rot[k] = -f.rot[k];
    }
    for (int i = 0; i < n; i++) {
        const double* u = &centred[3 * (size_t)i];
        const double* R = f.rot;
        const double px = R[0] * u[0] + R[1] * u[1] + R[2] * u[2], py = R[3] * u[0] + R[4] * u[1] + R[5] * u[2], pz = R[6] * u[0] + R[7] * u[1] + R[8] * u[2];
        f.uv[2 * (size_t)i] = px;
        f.uv[2 * (size_t)i + 1] = py;
        if (std::fabs(pz) > kFlat) return false;
    }
    const double back[3] = {-mean[0], -mean[1], -mean[2]};
    for (int r = 0; r < 3; r++) f.shift[r] = f.rot[3 * r] * back[0] + f.rot[3 * r + 1] * back[1] + f.rot[3 * r + 2] * back[2];
    return true;
}

// isotropic normalisation of n 2-D points: zero mean, mean squared distance 2. xs / ys: the normalised coordinates; undo / apply: 3 x 3
inline void isotropic(const double* pts, int n, std::vector<double>& xs, std::vector<double>& ys, double* undo, double* apply) {
    double mx = 0, my = 0;
    for (int i = 0; i < n; i++) {
        mx += pts[2 * (size_t)i];
        my += pts[2 * (size_t)i + 1];
    }
    mx = mx / (double)n;
    my = my / (double)n;
    xs.resize(n);
    ys.resize(n);
    double spread = 0;
    for (int i = 0; i < n; i++) {
        const double dx = pts[2 * (size_t)i] - mx, dy = pts[2 * (size_t)i + 1] - my;
        xs[i] = dx;
        ys[i] = dy;
        spread = spread + dx * dx + dy * dy;
    }
    const double gain = std::sqrt(2 * n / spread);
    for (int i = 0; i < n; i++) xs[i] = xs[i] * gain;
    for (int i = 0; i < n; i++) ys[i] = ys[i] * gain;
    const double u[9] = {1.0 / gain, 0, mx, 0, 1.0 / gain, my, 0, 0, 1};
    const double a[9] = {gain, 0, -gain * mx, 0, gain, -gain * my, 0, 0, 1};
    std::memcpy(undo, u, sizeof u);
    std::memcpy(apply, a, sizeof a);
}

inline void mat3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

// Harker & O'Leary's homography from plane points to normalised image points, H(2,2) = 1
inline void homography_ho(const double* plane_uv, const double* image_xy, int n, double* H) {
    std::vector<double> ax, ay, bx, by;
    double undoA[9], applyA[9], undoB[9], applyB[9];
    isotropic(plane_uv, n, ax, ay, undoA, applyA);
    isotropic(image_xy, n, bx, by, undoB, applyB);
    // the four product columns and their means
    std::vector<double> prod(4 * (size_t)n);
    double mean[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double v[4] = {-bx[i] * ax[i], -bx[i] * ay[i], -by[i] * ax[i], -by[i] * ay[i]};
        for (int k = 0; k < 4; k++) {
            prod[4 * (size_t)i + k] = v[k];
            mean[k] += v[k];
        }
    }
    for (int k = 0; k < 4; k++) mean[k] /= n;
    std::vector<double> mx(3 * (size_t)n), my(3 * (size_t)n);
    for (int i = 0; i < n; i++) {
        mx[3 * (size_t)i] = prod[4 * (size_t)i] - mean[0];
        mx[3 * (size_t)i + 1] = prod[4 * (size_t)i + 1] - mean[1];
        mx[3 * (size_t)i + 2] = -bx[i];
        my[3 * (size_t)i] = prod[4 * (size_t)i + 2] - mean[2];
        my[3 * (size_t)i + 1] = prod[4 * (size_t)i + 3] - mean[3];
        my[3 * (size_t)i + 2] = -by[i];
    }
    // projector onto the plane coordinates: (A A')^-1 A
    double sxx = 0, sxy = 0, syy = 0;
    for (int i = 0; i < n; i++) {
        sxx += ax[i] * ax[i];
        sxy += ax[i] * ay[i];
        syy += ay[i] * ay[i];
    }
    const double det = sxx * syy - sxy * sxy;
    const double i00 = syy / det, i01 = -sxy / det, i10 = -sxy / det, i11 = sxx / det;
    std::vector<double> px(n), py(n);
    for (int i = 0; i < n; i++) {
        px[i] = i00 * ax[i] + i01 * ay[i];
        py[i] = i10 * ax[i] + i11 * ay[i];
    }
    double Bx[6], By[6];
    for (int r = 0; r < 2; r++) {
        const std::vector<double>& row = r == 0 ? px : py;
        for (int c = 0; c < 3; c++) {
            double s0 = 0, s1 = 0;
            for (int i = 0; i < n; i++) {
                s0 += row[i] * mx[3 * (size_t)i + c];
                s1 += row[i] * my[3 * (size_t)i + c];
            }
            Bx[3 * r + c] = s0;
            By[3 * r + c] = s1;
        }
    }
    // residual rows (the x block first, then the y block) and their 3 x 3 Gram matrix
    std::vector<double> res(6 * (size_t)n);
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            res[3 * (size_t)i + c] = mx[3 * (size_t)i + c] - (ax[i] * Bx[c] + ay[i] * Bx[3 + c]);
            res[3 * ((size_t)n + i) + c] = my[3 * (size_t)i + c] - (ax[i] * By[c] + ay[i] * By[3 + c]);
        }
    double gram[9];
    for (int p = 0; p < 3; p++)
        for (int q = 0; q < 3; q++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += res[3 * (size_t)i + p] * res[3 * (size_t)i + q];
            gram[3 * p + q] = s;
        }
    double W[3], Ut[9], Vt[9];
    pnp::svd3(gram, W, Ut, Vt);
    const double* last = Vt + 6;   // the direction of the smallest singular value
    double Hn[9];
    for (int r = 0; r < 2; r++) {
        Hn[r] = -(Bx[3 * r] * last[0] + Bx[3 * r + 1] * last[1] + Bx[3 * r + 2] * last[2]);
        Hn[3 + r] = -(By[3 * r] * last[0] + By[3 * r + 1] * last[1] + By[3 * r + 2] * last[2]);
    }
    Hn[2] = -(mean[0] * last[0] + mean[1] * last[1]);
    Hn[5] = -(mean[2] * last[0] + mean[3] * last[1]);
    Hn[6] = last[0];
    Hn[7] = last[1];
    Hn[8] = last[2];
    double left[9];
    mat3(undoB, Hn, left);
    mat3(left, applyA, H);
    const double inv = 1 / H[8];
    for (int i = 0; i < 9; i++) H[i] = H[i] * inv;
}

// the two rotations compatible with the homography's Jacobian J at the plane's origin, which maps to (p, q)
inline void two_rotations(const double* J, double p, double q, double* Ra, double* Rb) {
    double T[9], V[9];
    turn_to_z(p, q, 1, T);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) V[3 * i + j] = T[3 * j + i];
    const double b00 = V[0] - p * V[6], b01 = V[1] - p * V[7], b10 = V[3] - q * V[6], b11 = V[4] - q * V[7];
    const double idet = 1.0 / ((b00 * b11 - b01 * b10));
    const double n00 = idet * b11, n01 = -idet * b01, n10 = -idet * b10, n11 = idet * b00;
    const double a00 = n00 * J[0] + n01 * J[2], a01 = n00 * J[1] + n01 * J[3], a10 = n10 * J[0] + n11 * J[2], a11 = n10 * J[1] + n11 * J[3];
    const double g00 = a00 * a00 + a01 * a01, g01 = a00 * a10 + a01 * a11, g11 = a10 * a10 + a11 * a11;
    const double top2 = 0.5 * (g00 + g11 + std::sqrt((g00 - g11) * (g00 - g11) + 4.0 * g01 * g01));   // largest singular value, squared
    const double top = std::sqrt(top2);
    const double r00 = a00 / top, r01 = a01 / top, r10 = a10 / top, r11 = a11 / top;
    const double q00 = r00 * r00, q01 = r01 * r01, q10 = r10 * r10, q11 = r11 * r11;
    const double h0 = std::sqrt(std::fmax(0.0, -q00 - q10 + 1));
    double h1 = std::sqrt(std::fmax(0.0, -q01 - q11 + 1));
    if ((-r00 * r01 - r10 * r11) < 0) h1 = -h1;
    const double k0 = h1 * r10 - h0 * r11, k1 = h0 * r01 - h1 * r00, k2 = r00 * r11 - r01 * r10;
    for (int r = 0; r < 3; r++) {
        const double* v = V + 3 * r;
        Ra[3 * r] = (r00)*v[0] + (r10)*v[1] + (h0)*v[2];
        Ra[3 * r + 1] = (r01)*v[0] + (r11)*v[1] + (h1)*v[2];
        Ra[3 * r + 2] = k0 * v[0] + k1 * v[1] + k2 * v[2];
        Rb[3 * r] = (r00)*v[0] + (r10)*v[1] + (-h0) * v[2];
        Rb[3 * r + 1] = (r01)*v[0] + (r11)*v[1] + (-h1) * v[2];
        Rb[3 * r + 2] = (-k0) * v[0] + (-k1) * v[1] + k2 * v[2];
    }
}

// least-squares translation for rotation R: normal equations of x (R P + t)_z = (R P + t)_x, y (...)_z = (...)_y
inline void translation_for(const double* uv, const double* xy, int n, const double* R, double* t) {
    const double m00 = (double)n, m11 = (double)n;
    double m02 = 0, m12 = 0, m22 = 0, v0 = 0, v1 = 0, v2 = 0;
    for (int i = 0; i < n; i++) {
        const double U = uv[2 * (size_t)i], Vv = uv[2 * (size_t)i + 1];
        const double rx = R[0] * U + R[1] * Vv, ry = R[3] * U + R[4] * Vv, rz = R[6] * U + R[7] * Vv;
        const double ca = -xy[2 * (size_t)i], cb = -xy[2 * (size_t)i + 1];
        m02 = m02 + ca;
        m12 = m12 + cb;
        m22 = m22 + (ca * ca) + (cb * cb);
        const double ex = -ca * rz - rx, ey = -cb * rz - ry;
        v0 = v0 + ex;
        v1 = v1 + ey;
        v2 = v2 + ca * ex + cb * ey;
    }
    const double idet = 1.0 / (m00 * m11 * m22 - m00 * m12 * m12 - m02 * m02 * m11);
    const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12, c02 = -m02 * m11, c11 = m00 * m22 - m02 * m02, c12 = -m00 * m12, c22 = m00 * m11;
    t[0] = idet * (c00 * v0 + c01 * v1 + c02 * v2);
    t[1] = idet * (c01 * v0 + c11 * v1 + c12 * v2);
    t[2] = idet * (c02 * v0 + c12 * v1 + c22 * v2);
}

// solvePnP(SOLVEPNP_IPPE) on double points; false for object points that are not coplanar (OpenCV's solver throws, solvePnP finds nothing)
inline bool solve(const double* obj, const double* img, int n, const pnp::Camera& cam, double* rvec, double* tvec) {
    if (n < 4) return false;
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
    std::vector<double> xy(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        xy[2 * (size_t)i] = (img[2 * (size_t)i] - cam.uc) * ifx;
        xy[2 * (size_t)i + 1] = (img[2 * (size_t)i + 1] - cam.vc) * ify;
    }
    PlaneFrame f;
    if (!to_plane(obj, n, f)) return false;
    double H[9];
    homography_ho(f.uv.data(), xy.data(), n, H);
    const double J[4] = {H[0] - H[6] * H[2], H[1] - H[7] * H[2], H[3] - H[6] * H[5], H[4] - H[7] * H[5]};
    double Rp[2][9], R[2][9], t[2][3], err[2];
    two_rotations(J, H[2], H[5], Rp[0], Rp[1]);
    for (int s = 0; s < 2; s++) {
        double tp[3];
        translation_for(f.uv.data(), xy.data(), n, Rp[s], tp);
        // back to the model frame: [Rp tp] * [rot shift; 0 1]
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) R[s][3 * r + c] = Rp[s][3 * r] * f.rot[c] + Rp[s][3 * r + 1] * f.rot[3 + c] + Rp[s][3 * r + 2] * f.rot[6 + c];
            t[s][r] = Rp[s][3 * r] * f.shift[0] + Rp[s][3 * r + 1] * f.shift[1] + Rp[s][3 * r + 2] * f.shift[2] + tp[r];
        }
        double e = 0;
        for (int i = 0; i < n; i++) {
            const double* X = obj + 3 * (size_t)i;
            const double xc = R[s][0] * X[0] + R[s][1] * X[1] + R[s][2] * X[2] + t[s][0], yc = R[s][3] * X[0] + R[s][4] * X[1] + R[s][5] * X[2] + t[s][1],
                         zc = R[s][6] * X[0] + R[s][7] * X[1] + R[s][8] * X[2] + t[s][2];
            const double iz = 1. / zc, du = xc * iz * cam.fu + cam.uc - img[2 * (size_t)i], dv = yc * iz * cam.fv + cam.vc - img[2 * (size_t)i + 1];
            e += du * du + dv * dv;
        }
        err[s] = e;
    }
    const int pick = err[1] < err[0] ? 1 : 0;
    if (!(err[pick] == err[pick])) return false;
    pnp::rvec_from_rotation(R[pick], rvec);
    for (int k = 0; k < 3; k++) tvec[k] = t[pick][k];
    return true;
}

}  // namespace ippe
}  // namespace apds
