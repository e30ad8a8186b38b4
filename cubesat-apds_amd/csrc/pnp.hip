// csrc/pnp.hip — homographier::pnp_solver_ransac (homographier/src/homographier/mod.rs:320-369) on the device.
//
// The reference forwards to cv::solvePnPRansac with distCoeffs = zeros(4,1) (mod.rs:344 shadows the argument),
// useExtrinsicGuess = false and SOLVEPNP_EPNP unless the caller names a method. OpenCV's loop is sequential: draw 5
// correspondences, EPnP on them, project every point, count inliers, shorten the iteration budget when a better model
// appears. Here the cv::RNG sample stream is generated ahead on the host (it does not depend on the scores), a batch of
// samples is solved one thread per sample (12x12 system in LDS), every hypothesis of the batch is scored against all
// points in one launch, and the host replays the sequential accept/shorten logic over the batch's counts; the result is
// the model and inlier set the sequential loop would have produced. The final EPnP over the inliers runs on the host in
// the same code (pnp_core.h), as OpenCV does it once, in index order. Every member of cv::SolvePnPMethod is served: P3P / AP3P swap
// the RANSAC kernel (4-point samples); ITERATIVE, SQPNP (sqpnp_core.h) and IPPE (ippe_core.h) swap the final solve; DLS / UPNP are
// EPnP in OpenCV 4; IPPE_SQUARE ends in solvePnP's npoints == 4 assertion unless there are exactly four correspondences (P3P).
#include "config.h"
#include "kernels.h"
#include "pnp_core.h"
#include "sqpnp_core.h"
#include "ippe_core.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace apds {

using pnp::Camera;

namespace {

// One thread per sample; per-thread LDS: 156 doubles (system / singular vectors / values) + 64 of solver workspace.
constexpr int PNP_THREADS = 32;
constexpr int PNP_LDS_DOUBLES = 220;
constexpr size_t PNP_LDS_BYTES = (size_t)PNP_LDS_DOUBLES * PNP_THREADS * sizeof(double);

__global__ __launch_bounds__(PNP_THREADS) void pnp_hypothesis_kernel(const float* __restrict__ obj, const float* __restrict__ img, const int* __restrict__ idx5,
                                                                     int B, Camera cam, double* __restrict__ models) {
    APDS_RAISE_WAVE_PRIORITY();
    extern __shared__ double pnp_lds[];
    const int h = blockIdx.x * PNP_THREADS + threadIdx.x;
    if (h >= B) return;
    pnp::Strided<double, PNP_THREADS> big{pnp_lds + threadIdx.x}, wrk{pnp_lds + 156 * PNP_THREADS + threadIdx.x};
    double pws[15], us[10], alphas[20], pcs[15];
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int id = idx5[h * 5 + j];
        pws[3 * j] = obj[3 * (size_t)id];
        pws[3 * j + 1] = obj[3 * (size_t)id + 1];
        pws[3 * j + 2] = obj[3 * (size_t)id + 2];
        // undistortPoints with k = 0 stores (u - cx)/fx as float; epnp maps it back to pixels in double
        const float xn = (float)(((double)img[2 * (size_t)id] - cam.uc) * ifx);
        const float yn = (float)(((double)img[2 * (size_t)id + 1] - cam.vc) * ify);
        us[2 * j] = xn * cam.fu + cam.uc;
        us[2 * j + 1] = yn * cam.fv + cam.vc;
    }
    double R[9], t[3], rv[3];
    pnp::epnp_pose(5, pnp::Plain<double>{pws}, pnp::Plain<double>{us}, pnp::Plain<double>{alphas}, pnp::Plain<double>{pcs}, cam, big, wrk, R, t);
    pnp::rvec_from_rotation(R, rv);
    double* m = models + (size_t)h * 6;
    m[0] = rv[0];
    m[1] = rv[1];
    m[2] = rv[2];
    m[3] = t[0];
    m[4] = t[1];
    m[5] = t[2];
}

// P3P kernel of the RANSAC loop: one thread per 4-point sample, everything in registers (closed form). valid[h] = 0 when the
// three-point system has no admissible solution (runKernel then returns no model and the iteration is skipped).
// AP3P = true: the same sample through ap3p.cpp's algebraic solver (SOLVEPNP_AP3P).
template <bool AP3P>
__global__ __launch_bounds__(64) void p3p_hypothesis_kernel(const float* __restrict__ obj, const float* __restrict__ img, const int* __restrict__ idx4, int B,
                                                            Camera cam, double* __restrict__ models, uint8_t* __restrict__ valid) {
    APDS_RAISE_WAVE_PRIORITY();
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= B) return;
    double mu[4], mv[4], P[12];
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int id = idx4[h * 4 + j];
        P[3 * j] = obj[3 * (size_t)id];
        P[3 * j + 1] = obj[3 * (size_t)id + 1];
        P[3 * j + 2] = obj[3 * (size_t)id + 2];
        // undistortPoints(k = 0, P = cameraMatrix): back to pixels, stored as float
        const double xn = ((double)img[2 * (size_t)id] - cam.uc) * ifx, yn = ((double)img[2 * (size_t)id + 1] - cam.vc) * ify;
        mu[j] = (float)(cam.fu * xn + cam.uc);
        mv[j] = (float)(cam.fv * yn + cam.vc);
    }
    double R[9], t[3], rv[3] = {0, 0, 0};
    const bool ok = AP3P ? pnp::ap3p_best_pose(cam, mu, mv, P, R, t) : pnp::p3p_best_pose(cam, mu, mv, P, R, t);
    if (ok) pnp::rvec_from_rotation(R, rv);
    double* m = models + (size_t)h * 6;
    m[0] = rv[0];
    m[1] = rv[1];
    m[2] = rv[2];
    m[3] = ok ? t[0] : 0.0;
    m[4] = ok ? t[1] : 0.0;
    m[5] = ok ? t[2] : 0.0;
    valid[h] = ok ? 1 : 0;
}

constexpr int PNP_HT = 4;   // hypotheses per block of the scoring kernel

// grid: x = ceil(B / PNP_HT), y = point parts. good[h] += #points whose float squared reprojection error is <= t
__global__ __launch_bounds__(256) void pnp_score_kernel(const float* __restrict__ obj, const float* __restrict__ img, int n, const double* __restrict__ models,
                                                        int B, Camera cam, float t, int* __restrict__ good) {
    APDS_RAISE_WAVE_PRIORITY();
    const int h0 = blockIdx.x * PNP_HT;
    double R[PNP_HT][9], tv[PNP_HT][3];
#pragma unroll
    for (int h = 0; h < PNP_HT; h++) {
        const double* m = models + (size_t)min(h0 + h, B - 1) * 6;
        const double rv[3] = {m[0], m[1], m[2]};
        pnp::rotation_from_rvec(rv, R[h]);
        tv[h][0] = m[3];
        tv[h][1] = m[4];
        tv[h][2] = m[5];
    }
    int cnt[PNP_HT];
#pragma unroll
    for (int h = 0; h < PNP_HT; h++) cnt[h] = 0;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = min(n, i0 + per);
    for (int base = i0; base < i1; base += 256) {
        const int i = base + threadIdx.x;
        const bool in = i < i1;
        const size_t q = in ? i : i0;
        const float X = obj[3 * q], Y = obj[3 * q + 1], Z = obj[3 * q + 2], u = img[2 * q], v = img[2 * q + 1];
#pragma unroll
        for (int h = 0; h < PNP_HT; h++) {
            const float e = pnp::reprojection_sqerr(R[h], tv[h], cam, X, Y, Z, u, v);
            cnt[h] += __popcll(__ballot(in && e <= t));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int h = 0; h < PNP_HT; h++)
            if (h0 + h < B && cnt[h]) atomicAdd(&good[h0 + h], cnt[h]);
    }
}

__global__ void pnp_mask_kernel(const float* __restrict__ obj, const float* __restrict__ img, int n, const double* __restrict__ model, Camera cam, float t,
                                uint8_t* __restrict__ mask) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double R[9];
    const double rv[3] = {model[0], model[1], model[2]}, tv[3] = {model[3], model[4], model[5]};
    pnp::rotation_from_rvec(rv, R);
    mask[i] = pnp::reprojection_sqerr(R, tv, cam, obj[3 * (size_t)i], obj[3 * (size_t)i + 1], obj[3 * (size_t)i + 2], img[2 * (size_t)i], img[2 * (size_t)i + 1]) <= t;
}

// getSubset with the default checkSubset: `model_points` distinct indices from the cv::RNG stream
void next_sample(int count, int* idx, pnp::MwcRng& rng, int model_points) {
    for (int i = 0; i < model_points; ++i) {
        int v;
        for (;;) {
            v = (int)(rng.next() % (unsigned)count);
            bool dup = false;
            for (int j = 0; j < i; j++) dup |= idx[j] == v;
            if (!dup) break;
        }
        idx[i] = v;
    }
}

int update_num_iters(double p, double ep, int modelPoints, int maxIters) {   // ptsetreg.cpp RANSACUpdateNumIters
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

// EPnP on host arrays (float for the n == 5 shortcut, double for the all-inlier solve), then Rodrigues
template <typename T>
void host_epnp(const T* obj, const T* img, int n, const Camera& cam, double* rvec, double* tvec) {
    std::vector<double> pws(3 * (size_t)n), us(2 * (size_t)n), alphas(4 * (size_t)n), pcs(3 * (size_t)n), big(156), wrk(64);
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) pws[3 * (size_t)i + c] = obj[3 * (size_t)i + c];
        const T xn = (T)(((double)img[2 * (size_t)i] - cam.uc) * ifx);
        const T yn = (T)(((double)img[2 * (size_t)i + 1] - cam.vc) * ify);
        us[2 * (size_t)i] = xn * cam.fu + cam.uc;
        us[2 * (size_t)i + 1] = yn * cam.fv + cam.vc;
    }
    double R[9];
    pnp::epnp_pose(n, pnp::Plain<double>{pws.data()}, pnp::Plain<double>{us.data()}, pnp::Plain<double>{alphas.data()}, pnp::Plain<double>{pcs.data()}, cam,
                   pnp::Plain<double>{big.data()}, pnp::Plain<double>{wrk.data()}, R, tvec);
    pnp::rvec_from_rotation(R, rvec);
}

// solvePnP(4 points, SOLVEPNP_P3P) on host arrays
template <typename T>
bool host_p3p(const T* obj, const T* img, const Camera& cam, double* rvec, double* tvec, bool ap3p = false) {
    const double ifx = 1. / cam.fu, ify = 1. / cam.fv;
    double mu[4], mv[4], P[12];
    for (int i = 0; i < 4; i++) {
        const double xn = ((double)img[2 * i] - cam.uc) * ifx, yn = ((double)img[2 * i + 1] - cam.vc) * ify;
        mu[i] = (T)(cam.fu * xn + cam.uc);
        mv[i] = (T)(cam.fv * yn + cam.vc);
        for (int c = 0; c < 3; c++) P[3 * i + c] = obj[3 * i + c];
    }
    double R[9];
    if (!(ap3p ? pnp::ap3p_best_pose(cam, mu, mv, P, R, tvec) : pnp::p3p_best_pose(cam, mu, mv, P, R, tvec))) return false;
    pnp::rvec_from_rotation(R, rvec);
    return true;
}

// ---- SOLVEPNP_ITERATIVE: the final solvePnP over the inliers (host; six parameters, a few thousand residuals at most) ---------------------
// solvePnPRansac(flags = SOLVEPNP_ITERATIVE) as recalled for OpenCV 4.8: EPnP stays the RANSAC kernel, and the final solvePnP runs with the
// CALLER's useExtrinsicGuess, which the reference sets to false (mod.rs:354). cvFindExtrinsicCameraParams2 therefore builds its own
// starting pose first - a homography for planar object points, the DLT otherwise (iterative_start_pose) - and then refines it with
// CvLevMarq(6, 2 n, 20 iterations, FLT_EPSILON) around cvProjectPoints2's analytic Jacobian, zero distortion (mod.rs:344). With five
// non-planar inliers the DLT cannot start ("needs at least 6 points"): solvePnPRansac catches that and keeps the RANSAC model.
// (Rounds 2 - 3 started the refinement from the best RANSAC model for every input: ADVICE r3.)
// obj: n x 3, img: n x 2 doubles. Returns false in the five-point case above.
static bool iterative_start_pose(const double* obj, const double* img, int n, const Camera& cam, double* pose, hipStream_t s) {
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
    double W[3], Ut[9], V[9];   // V = V^T (CV_SVD_V_T): its rows are the right singular vectors
    pnp::svd3(MM, W, Ut, V);
    double R[9], t[3];
    if (W[2] / W[1] < 1e-3) {   // all object points in one plane
        if (V[2] * V[2] + V[5] * V[5] < 1e-10)
            for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1. : 0.;
        const double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
        if (det < 0)
            for (int i = 0; i < 9; i++) V[i] = -V[i];
        double T[3];
        for (int a = 0; a < 3; a++) T[a] = -(V[a * 3] * Mc[0] + V[a * 3 + 1] * Mc[1] + V[a * 3 + 2] * Mc[2]);
        // the points in their plane -> the normalised image points: cv::findHomography (method 0) on float copies; this library's own
        // find_homography_device does it (the least-squares path: host refit up to 256 points, device reductions above)
        std::vector<float> src(2 * (size_t)n), dst(2 * (size_t)n);
        for (int i = 0; i < n; i++) {
            const double* M = obj + 3 * i;
            src[2 * (size_t)i] = (float)(V[0] * M[0] + V[1] * M[1] + V[2] * M[2] + T[0]);
            src[2 * (size_t)i + 1] = (float)(V[3] * M[0] + V[4] * M[1] + V[5] * M[2] + T[1]);
            dst[2 * (size_t)i] = (float)mn[2 * (size_t)i];
            dst[2 * (size_t)i + 1] = (float)mn[2 * (size_t)i + 1];
        }
        ThreadCtx& c = ctx();
        float* src_dev = c.alloc_n<float>(src.size());
        float* dst_dev = c.alloc_n<float>(dst.size());
        HIP_CHECK(hipMemcpyAsync(src_dev, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(dst_dev, dst.data(), dst.size() * sizeof(float), hipMemcpyHostToDevice, s));
        double h[9];
        const int found = find_homography_device(src_dev, dst_dev, n, 0, 3.0, 2000, 0.995, h, nullptr, s);
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
            pnp::rvec_from_rotation(h, rv);     // (orthonormalises: cvRodrigues2 takes U V^T of its input first)
            pnp::rotation_from_rvec(rv, Rh);
            for (int a = 0; a < 3; a++) t[a] = (Rh[a * 3] * T[0] + Rh[a * 3 + 1] * T[1] + Rh[a * 3 + 2] * T[2]) + t[a];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) R[a * 3 + b] = Rh[a * 3] * V[b] + Rh[a * 3 + 1] * V[3 + b] + Rh[a * 3 + 2] * V[6 + b];
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            t[0] = t[1] = t[2] = 0;
        }
    } else {   // non-planar: DLT
        if (n < 6) return false;
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
        double LW[12], LAt[144], LV[144];
        for (int i = 0; i < 12; i++)
            for (int k = 0; k < 12; k++) LAt[i * 12 + k] = LL[k * 12 + i];
        pnp::svd_rows<true>(pnp::Plain<double>{LAt}, 12, 12, pnp::Plain<double>{LW}, pnp::Plain<double>{LV});
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
        pnp::svd3(RR, w3, U3t, V3t);
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
    pnp::rvec_from_rotation(R, pose);
    pose[3] = t[0];
    pose[4] = t[1];
    pose[5] = t[2];
    return true;
}


struct PoseRefiner {
    const double* obj;   // n x 3
    const double* img;   // n x 2
    int n;
    Camera cam;
    std::vector<double> jac, res;   // 2 n x 6, 2 n

    // R and dR/dr (3 x 9) of a rotation vector: cvRodrigues2 with its Jacobian
    static void rotation_and_derivative(const double* rv, double* R, double* D) {
        static const double skew_d[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        static const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double theta = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
        if (theta < DBL_EPSILON) {
            std::memcpy(R, eye, sizeof(eye));
            std::memcpy(D, skew_d, sizeof(skew_d));
            return;
        }
        double sn, cs;
        pnp::sincos_fixed(theta, sn, cs);
        const double c1 = 1. - cs, itheta = 1. / theta;
        const double u[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
        const double outer[9] = {u[0] * u[0], u[0] * u[1], u[0] * u[2], u[0] * u[1], u[1] * u[1], u[1] * u[2], u[0] * u[2], u[1] * u[2], u[2] * u[2]};
        const double skew[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
        for (int k = 0; k < 9; k++) R[k] = (cs * eye[k] + c1 * outer[k]) + sn * skew[k];
        const double outer_d[27] = {u[0] + u[0], u[1], u[2], u[1], 0, 0, u[2], 0, 0, 0, u[0], 0, u[0], u[1] + u[1], u[2], 0, u[2], 0,
                                    0, 0, u[0], 0, 0, u[1], u[0], u[1], u[2] + u[2]};
        for (int i = 0; i < 3; i++) {
            const double ui = u[i];
            const double a0 = -sn * ui, a1 = (sn - 2 * c1 * itheta) * ui, a2 = c1 * itheta, a3 = (cs - sn * itheta) * ui, a4 = sn * itheta;
            for (int k = 0; k < 9; k++) D[i * 9 + k] = a0 * eye[k] + a1 * outer[k] + a2 * outer_d[i * 9 + k] + a3 * skew[k] + a4 * skew_d[i * 9 + k];
        }
    }
    // residuals (projection - measurement) of pose p and, if asked, their Jacobian
    void evaluate(const double* p, bool with_jacobian) {
        double R[9], D[27];
        rotation_and_derivative(p, R, D);
        for (int i = 0; i < n; i++) {
            const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
            double x = R[0] * X + R[1] * Y + R[2] * Z + p[3];
            double y = R[3] * X + R[4] * Y + R[5] * Z + p[4];
            double z = R[6] * X + R[7] * Y + R[8] * Z + p[5];
            z = z ? 1. / z : 1;
            x *= z;
            y *= z;
            res[2 * (size_t)i] = (x * cam.fu + cam.uc) - img[2 * i];
            res[2 * (size_t)i + 1] = (y * cam.fv + cam.vc) - img[2 * i + 1];
            if (!with_jacobian) continue;
            double* ju = &jac[(size_t)(2 * i) * 6];
            double* jv = ju + 6;
            const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                ju[3 + j] = cam.fu * dxdt[j];
                jv[3 + j] = cam.fv * dydt[j];
            }
            for (int j = 0; j < 3; j++) {
                const double* d = D + 9 * j;
                const double dx0 = X * d[0] + Y * d[1] + Z * d[2], dy0 = X * d[3] + Y * d[4] + Z * d[5], dz0 = X * d[6] + Y * d[7] + Z * d[8];
                ju[j] = cam.fu * (z * (dx0 - x * dz0));
                jv[j] = cam.fv * (z * (dy0 - y * dz0));
            }
        }
    }
    static double norm2(const double* v, size_t k) {
        double s = 0;
        for (size_t i = 0; i < k; i++) s += v[i] * v[i];
        return std::sqrt(s);
    }
    // cv::solve(A, b, x, DECOMP_SVD) for the 6 x 6 normal equations
    static void solve6(const double* A, const double* b, double* x) {
        double At[36], Vt[36], W[6];
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 6; k++) At[i * 6 + k] = A[k * 6 + i];
        pnp::svd_rows<true>(pnp::Plain<double>{At}, 6, 6, pnp::Plain<double>{W}, pnp::Plain<double>{Vt});
        double threshold = 0;
        for (int i = 0; i < 6; i++) threshold += W[i];
        threshold *= DBL_EPSILON * 2;
        for (int j = 0; j < 6; j++) x[j] = 0;
        for (int i = 0; i < 6; i++) {
            double wi = W[i];
            if (std::fabs(wi) <= threshold) continue;
            wi = 1 / wi;
            double s = 0;
            for (int j = 0; j < 6; j++) s += At[i * 6 + j] * b[j];
            s *= wi;
            for (int j = 0; j < 6; j++) x[j] = x[j] + s * Vt[i * 6 + j];
        }
    }
    // CvLevMarq::update / step; pose: the starting pose in, the refined pose out
    void run(double* pose) {
        const size_t m = 2 * (size_t)n;
        jac.resize(m * 6);
        res.resize(m);
        double N[36], g[6], before[6], damped[36], delta[6];
        int lambda_lg10 = -3, iters = 0;
        double prev_norm = DBL_MAX;
        auto take_step = [&]() {
            const double lambda = std::exp(lambda_lg10 * std::log(10.));
            std::memcpy(damped, N, sizeof(N));
            for (int i = 0; i < 6; i++) damped[i * 6 + i] *= 1. + lambda;
            solve6(damped, g, delta);
            for (int i = 0; i < 6; i++) pose[i] = before[i] - delta[i];
        };
        evaluate(pose, true);
        for (;;) {
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++) {
                    double s = 0;
                    for (size_t k = 0; k < m; k++) s += jac[k * 6 + a] * jac[k * 6 + b];
                    N[a * 6 + b] = N[b * 6 + a] = s;
                }
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (size_t k = 0; k < m; k++) s += jac[k * 6 + a] * res[k];
                g[a] = s;
            }
            std::memcpy(before, pose, sizeof(before));
            take_step();
            if (iters == 0) prev_norm = norm2(res.data(), m);
            evaluate(pose, false);
            double now;
            for (;;) {
                now = norm2(res.data(), m);
                if (!(now > prev_norm && ++lambda_lg10 <= 16)) break;
                take_step();
                evaluate(pose, false);
            }
            lambda_lg10 = std::max(lambda_lg10 - 1, -16);
            double moved[6];
            for (int i = 0; i < 6; i++) moved[i] = pose[i] - before[i];
            if (++iters >= 20 || norm2(moved, 6) / norm2(before, 6) < FLT_EPSILON) return;
            prev_norm = now;
            evaluate(pose, true);
        }
    }
};

}  // namespace

int pnp_ransac_device(const double* obj_xyz, const double* img_xy, int n, const double* K, int iterations, float reproj_thr, double confidence, int method,
                      double* rvec, double* tvec, int32_t* inliers, int* n_inliers, hipStream_t s) {
    APDS_REQUIRE(n_inliers, APDS_ERR_BAD_ARG, "null argument");
    *n_inliers = 0;
    APDS_REQUIRE(obj_xyz && img_xy && K && rvec && tvec && inliers, APDS_ERR_BAD_ARG, "null argument");
    APDS_REQUIRE(n >= 4, APDS_ERR_ASSERT, "solvePnPRansac needs at least 4 correspondences");
    // solvePnPGeneric: SOLVEPNP_DLS and SOLVEPNP_UPNP are "broken implementations" that run EPnP
    if (method == APDS_SOLVEPNP_DLS || method == APDS_SOLVEPNP_UPNP) method = APDS_SOLVEPNP_EPNP;
    APDS_REQUIRE(method == APDS_SOLVEPNP_EPNP || method == APDS_SOLVEPNP_P3P || method == APDS_SOLVEPNP_ITERATIVE || method == APDS_SOLVEPNP_AP3P ||
                     method == APDS_SOLVEPNP_SQPNP || method == APDS_SOLVEPNP_IPPE_SQUARE || method == APDS_SOLVEPNP_IPPE,
                 APDS_ERR_NOT_IMPLEMENTED, "unknown cv::SolvePnPMethod (MAX_COUNT and beyond)");
    // kernel choice of solvePnPRansac: P3P / AP3P on 4 points when asked for, P3P when there are only 4 points; EPnP on 5 otherwise
    const bool ap3p = method == APDS_SOLVEPNP_AP3P;
    const bool p3p = method == APDS_SOLVEPNP_P3P || ap3p || n == 4;
    const int model_points = p3p ? 4 : 5;
    const Camera cam{K[0], K[4], K[2], K[5]};
    // solvePnPRansac converts CV_64F points to CV_32F before anything else
    std::vector<float> op(3 * (size_t)n), ip(2 * (size_t)n);
    for (size_t i = 0; i < op.size(); i++) op[i] = (float)obj_xyz[i];
    for (size_t i = 0; i < ip.size(); i++) ip[i] = (float)img_xy[i];
    if (n == model_points) {   // model_points == npoints: one direct solve, every point an inlier
        if (p3p) {
            if (!host_p3p<float>(op.data(), ip.data(), cam, rvec, tvec, ap3p)) return 0;
        } else {
            host_epnp<float>(op.data(), ip.data(), n, cam, rvec, tvec);
        }
        for (int i = 0; i < n; i++) inliers[i] = i;
        *n_inliers = n;
        return 1;
    }
    ThreadCtx& c = ctx();
    float* obj_dev = c.alloc_n<float>(op.size());
    float* img_dev = c.alloc_n<float>(ip.size());
    HIP_CHECK(hipMemcpyAsync(obj_dev, op.data(), op.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(img_dev, ip.data(), ip.size() * sizeof(float), hipMemcpyHostToDevice, s));

    // a batch costs about the same wall time up to ~16 k samples (one thread each, latency-bound): speculate deep
    const int batch_env = config().pnp_batch;
    int niters = std::max(iterations, 1), maxGood = 0, iter = 0;
    const int batch = std::max(PNP_HT, std::min(batch_env, niters));
    const float t = (float)((double)reproj_thr * (double)reproj_thr);
    int* idx_dev = c.alloc_n<int>((size_t)batch * model_points);
    double* models_dev = c.alloc_n<double>((size_t)batch * 6);
    int* good_dev = c.alloc_n<int>(batch);
    uint8_t* valid_dev = c.alloc_n<uint8_t>(batch);
    uint8_t* mask_dev = c.alloc_n<uint8_t>(n);
    std::vector<int> idx((size_t)batch * model_points), good(batch);
    std::vector<uint8_t> valid(batch, 1);
    std::vector<double> models((size_t)batch * 6);
    double best[6] = {0, 0, 0, 0, 0, 0};
    pnp::MwcRng rng{(uint64_t)-1};
    while (iter < niters) {
        const int B = std::min(batch, niters - iter);
        for (int b = 0; b < B; b++) next_sample(n, &idx[(size_t)b * model_points], rng, model_points);
        HIP_CHECK(hipMemcpyAsync(idx_dev, idx.data(), (size_t)B * model_points * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(good_dev, 0, (size_t)B * sizeof(int), s));
        if (ap3p)
            hipLaunchKernelGGL(p3p_hypothesis_kernel<true>, dim3(ceil_div(B, 64)), dim3(64), 0, s, (const float*)obj_dev, (const float*)img_dev, (const int*)idx_dev, B,
                               cam, models_dev, valid_dev);
        else if (p3p)
            hipLaunchKernelGGL(p3p_hypothesis_kernel<false>, dim3(ceil_div(B, 64)), dim3(64), 0, s, (const float*)obj_dev, (const float*)img_dev, (const int*)idx_dev, B,
                               cam, models_dev, valid_dev);
        else
            hipLaunchKernelGGL(pnp_hypothesis_kernel, dim3(ceil_div(B, PNP_THREADS)), dim3(PNP_THREADS), PNP_LDS_BYTES, s, (const float*)obj_dev,
                               (const float*)img_dev, (const int*)idx_dev, B, cam, models_dev);
        {
            KernelTimer timer("pnp_score", s);
            const int parts = std::max(1, std::min(64, ceil_div(256 * 8, ceil_div(B, PNP_HT))));
            hipLaunchKernelGGL(pnp_score_kernel, dim3(ceil_div(B, PNP_HT), parts), dim3(256), 0, s, (const float*)obj_dev, (const float*)img_dev, n,
                               (const double*)models_dev, B, cam, t, good_dev);
        }
        HIP_CHECK(hipMemcpyAsync(good.data(), good_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(models.data(), models_dev, (size_t)B * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
        if (p3p) HIP_CHECK(hipMemcpyAsync(valid.data(), valid_dev, (size_t)B, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        // replay the sequential loop over the speculated iterations (samples beyond a shortened budget are discarded)
        for (int b = 0; b < B && iter < niters; b++, iter++) {
            if (!valid[b]) continue;   // P3P found no pose for this sample: the iteration yields no model
            if (good[b] > std::max(maxGood, model_points - 1)) {
                std::memcpy(best, &models[(size_t)b * 6], sizeof(best));
                maxGood = good[b];
                niters = update_num_iters(confidence, (double)(n - good[b]) / n, model_points, niters);
            }
        }
    }
    HIP_CHECK(hipGetLastError());
    if (maxGood <= 0) return 0;
    std::vector<uint8_t> mask(n);
    HIP_CHECK(hipMemcpyAsync(models_dev, best, sizeof(best), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(pnp_mask_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, (const float*)obj_dev, (const float*)img_dev, n, (const double*)models_dev, cam, t,
                       mask_dev);
    HIP_CHECK(hipMemcpyAsync(mask.data(), mask_dev, n, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    // final pose from all inliers (as doubles converted back from the float copies), in index order
    std::vector<double> oi, ii;
    int cnt = 0;
    for (int i = 0; i < n; i++)
        if (mask[i]) {
            for (int k = 0; k < 3; k++) oi.push_back((double)op[3 * (size_t)i + k]);
            for (int k = 0; k < 2; k++) ii.push_back((double)ip[2 * (size_t)i + k]);
            inliers[cnt++] = i;
        }
    APDS_REQUIRE(cnt == maxGood, APDS_ERR_INTERNAL, "inlier mask disagrees with the scored count");
    // SOLVEPNP_IPPE_SQUARE: the RANSAC kernel was EPnP on 5 points (or the direct P3P solve above for n == 4), and the final solvePnP over
    // the >= 5 inliers starts with CV_Assert(npoints == 4): solvePnPRansac rethrows it, the reference returns Err(MatError::Opencv)
    APDS_REQUIRE(method != APDS_SOLVEPNP_IPPE_SQUARE, APDS_ERR_ASSERT, "SOLVEPNP_IPPE_SQUARE: solvePnP asserts npoints == 4 on the inlier set");
    if (method == APDS_SOLVEPNP_ITERATIVE) {   // no extrinsic guess (mod.rs:354): homography / DLT start, then Levenberg-Marquardt
        double pose[6];
        std::memcpy(pose, best, sizeof(pose));   // (what stays when five non-planar inliers cannot start the DLT)
        if (iterative_start_pose(oi.data(), ii.data(), cnt, cam, pose, s)) PoseRefiner{oi.data(), ii.data(), cnt, cam, {}, {}}.run(pose);
        std::memcpy(rvec, pose, 3 * sizeof(double));
        std::memcpy(tvec, pose + 3, 3 * sizeof(double));
    } else if (method == APDS_SOLVEPNP_SQPNP || method == APDS_SOLVEPNP_IPPE) {
        // the RANSAC kernel stayed EPnP; the last solvePnP over the inliers is SQPnP (sqpnp_core.h) or IPPE (ippe_core.h: planar targets;
        // inliers that are not coplanar have no IPPE pose)
        const bool posed = method == APDS_SOLVEPNP_IPPE ? ippe::solve(oi.data(), ii.data(), cnt, cam, rvec, tvec) : sqpnp::solve(oi.data(), ii.data(), cnt, cam, rvec, tvec);
        if (!posed) {   // no pose: solvePnPRansac hands back the RANSAC model and returns false
            std::memcpy(rvec, best, 3 * sizeof(double));
            std::memcpy(tvec, best + 3, 3 * sizeof(double));
            return 0;
        }
    } else {
        host_epnp<double>(oi.data(), ii.data(), cnt, cam, rvec, tvec);
    }
    *n_inliers = cnt;
    return 1;
}

// solvePnP(SOLVEPNP_IPPE) alone, for the parity tests (host arithmetic; no device work)
int pnp_ippe_host(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec) {
    APDS_REQUIRE(obj_xyz && img_xy && K && rvec && tvec, APDS_ERR_BAD_ARG, "null argument");
    APDS_REQUIRE(n >= 4, APDS_ERR_ASSERT, "IPPE needs at least 4 correspondences");
    const Camera cam{K[0], K[4], K[2], K[5]};
    return ippe::solve(obj_xyz, img_xy, n, cam, rvec, tvec) ? 1 : 0;
}

// solvePnP(SOLVEPNP_SQPNP) alone, for the parity tests (host arithmetic; no device work)
int pnp_sqpnp_host(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec) {
    APDS_REQUIRE(obj_xyz && img_xy && K && rvec && tvec, APDS_ERR_BAD_ARG, "null argument");
    APDS_REQUIRE(n >= 3, APDS_ERR_ASSERT, "SQPnP needs at least 3 correspondences");
    const Camera cam{K[0], K[4], K[2], K[5]};
    return sqpnp::solve(obj_xyz, img_xy, n, cam, rvec, tvec) ? 1 : 0;
}

// per-stage hooks for the parity tests: the pose of explicit samples (model_points 5: EPnP, 4: P3P; a P3P sample without a pose
// comes back as NaNs)
void pnp_hypotheses_device(const double* obj_xyz, const double* img_xy, int n, const double* K, const int32_t* idx5, int B, int model_points,
                           double* models_host, hipStream_t s) {
    const bool ap3p = model_points == 40;   // 40: four points through the AP3P solver
    if (ap3p) model_points = 4;
    APDS_REQUIRE(obj_xyz && img_xy && K && idx5 && models_host && (model_points == 4 || model_points == 5) && n >= model_points && B >= 1, APDS_ERR_BAD_ARG,
                 "bad argument");
    for (int i = 0; i < B * model_points; i++) APDS_REQUIRE(idx5[i] >= 0 && idx5[i] < n, APDS_ERR_OUT_OF_RANGE, "sample index out of range");
    const Camera cam{K[0], K[4], K[2], K[5]};
    std::vector<float> op(3 * (size_t)n), ip(2 * (size_t)n);
    for (size_t i = 0; i < op.size(); i++) op[i] = (float)obj_xyz[i];
    for (size_t i = 0; i < ip.size(); i++) ip[i] = (float)img_xy[i];
    ThreadCtx& c = ctx();
    float* obj_dev = c.alloc_n<float>(op.size());
    float* img_dev = c.alloc_n<float>(ip.size());
    int* idx_dev = c.alloc_n<int>((size_t)B * model_points);
    double* models_dev = c.alloc_n<double>((size_t)B * 6);
    uint8_t* valid_dev = c.alloc_n<uint8_t>(B);
    HIP_CHECK(hipMemcpyAsync(obj_dev, op.data(), op.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(img_dev, ip.data(), ip.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(idx_dev, idx5, (size_t)B * model_points * sizeof(int), hipMemcpyHostToDevice, s));
    if (model_points == 4 && ap3p)
        hipLaunchKernelGGL(p3p_hypothesis_kernel<true>, dim3(ceil_div(B, 64)), dim3(64), 0, s, (const float*)obj_dev, (const float*)img_dev, (const int*)idx_dev, B, cam,
                           models_dev, valid_dev);
    else if (model_points == 4)
        hipLaunchKernelGGL(p3p_hypothesis_kernel<false>, dim3(ceil_div(B, 64)), dim3(64), 0, s, (const float*)obj_dev, (const float*)img_dev, (const int*)idx_dev, B, cam,
                           models_dev, valid_dev);
    else
        hipLaunchKernelGGL(pnp_hypothesis_kernel, dim3(ceil_div(B, PNP_THREADS)), dim3(PNP_THREADS), PNP_LDS_BYTES, s, (const float*)obj_dev,
                           (const float*)img_dev, (const int*)idx_dev, B, cam, models_dev);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(models_host, models_dev, (size_t)B * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (model_points == 4) {
        std::vector<uint8_t> valid(B);
        HIP_CHECK(hipMemcpyAsync(valid.data(), valid_dev, (size_t)B, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (int b = 0; b < B; b++)
            if (!valid[b])
                for (int k = 0; k < 6; k++) models_host[(size_t)b * 6 + k] = NAN;
    }
    HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace apds
