// csrc/homography_rho.hip — HomographyMethod::RHO on gfx950.
//
// Replaces cv::findHomography(..., RHO, ...) behind /root/reference/homographier/src/homographier/mod.rs:25-31,241-250
// (OpenCV calib3d/src/rho.cpp: PROSAC sampling + SPRT verification + non-randomness iteration bound + LM refinement, binary32).
//
// rho.cpp is one sequential loop: draw a sample, solve, walk the points until the SPRT accepts or rejects, adapt the test and the
// iteration bound, repeat. Its sample stream depends on the results only through the PROSAC pool limit (phMax), which changes on
// the rare "new best model" events. So the loop is SPECULATED in batches: the host draws the samples of the next B iterations
// (xorshift128+ stream, degeneracy test, 4-point solve: a few hundred flops each), ONE kernel evaluates every model against
// every point (B x N reprojections, the only heavy part) and leaves the inlier flags as bit rows, and the host then REPLAYS
// rho.cpp's loop over those rows: the SPRT walk reads bits instead of reprojecting, so every accept / reject, every update of
// (epsilon, delta, A), of the best model and of the bounds happens exactly where the sequential program takes it. When a new best
// model moves phMax, the rest of the batch was drawn under the wrong pool: it is dropped and drawn again from the saved generator
// state. The result (inlier set, H) is the sequential algorithm's, bit for bit; the reference returns no mask for RHO
// (mod.rs:253-257) but the C ABI fills it like OpenCV does.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace apds {

namespace {

struct F2 {
    float x, y;
};

// One thread per (point, model): binary32 reprojection, the operations and their order as in rho.cpp's evaluateModelSPRT (and as in
// oracle/rho_oracle.cpp); one 64-bit word of flags per wave. models: B x 9 floats with H[8] = 1.
__global__ __launch_bounds__(256) void rho_inlier_bits_kernel(const F2* __restrict__ src, const F2* __restrict__ dst, int n, const float* __restrict__ models,
                                                              float maxDsq, unsigned long long* __restrict__ bits, int words) {
    APDS_RAISE_WAVE_PRIORITY();
    const float* H = models + (size_t)blockIdx.y * 9;
    const float h0 = H[0], h1 = H[1], h2 = H[2], h3 = H[3], h4 = H[4], h5 = H[5], h6 = H[6], h7 = H[7];
    for (int base = blockIdx.x * 256; base < words * 64; base += gridDim.x * 256) {
        const int i = base + threadIdx.x;
        bool in = false;
        if (i < n) {
            const F2 p = src[i], q = dst[i];
            float rx = h0 * p.x + h1 * p.y + h2;
            float ry = h3 * p.x + h4 * p.y + h5;
            const float rz = h6 * p.x + h7 * p.y + 1.0f;
            rx /= rz;
            ry /= rz;
            rx -= q.x;
            ry -= q.y;
            rx *= rx;
            ry *= ry;
            in = rx + ry <= maxDsq;
        }
        const unsigned long long b = __ballot(in);
        if ((threadIdx.x & 63) == 0 && (i >> 6) < words) bits[(size_t)blockIdx.y * words + (i >> 6)] = b;   // (a block may reach past the last word)
    }
}

// ---- host side: the sequential controller -------------------------------------------------------------------------------------
struct Xorshift128p {
    uint64_t a, b;
    void seed(uint64_t v) {
        a = v;
        b = ~v;
        for (int k = 0; k < 20; k++) next();
    }
    double next() {
        uint64_t x = a;
        const uint64_t y = b;
        x ^= x << 23;
        x ^= x >> 17;
        x ^= y ^ (y >> 26);
        a = y;
        b = x;
        return (double)(x + y) * 5.421010862427522e-20;
    }
};

struct Prosac {   // everything the sample of iteration `it` depends on
    Xorshift128p rng;
    unsigned phNum, phEndI;
    double phEndFpI;
};

unsigned iteration_bound(double confidence, double inlierRate, unsigned sampleSize, unsigned cap) {
    confidence = std::min(std::max(confidence, 0.0), 1.0);
    inlierRate = std::min(std::max(inlierRate, 0.0), 1.0);
    const double pBad = 1. - std::pow(inlierRate, (double)sampleSize);
    if (pBad >= 1.) return cap;
    if (pBad <= 0.) return 1;
    const double need = std::ceil(std::log(1. - confidence) / std::log(pBad));
    return need >= (double)cap ? cap : (unsigned)need;
}

void draw(Xorshift128p& g, unsigned k, unsigned* out, unsigned pool) {
    if (2 * k > pool) {
        unsigned taken = 0;
        for (unsigned i = 0; i < pool && taken < k; i++) {
            const double u = g.next();
            if ((double)(k - taken) > (double)(pool - i) * u) out[taken++] = i;
        }
        return;
    }
    for (unsigned i = 0; i < k; i++) {
        for (;;) {
            out[i] = (unsigned)(pool * g.next());
            bool seen = false;
            for (unsigned j = 0; j < i; j++) seen |= out[j] == out[i];
            if (!seen) break;
        }
    }
}

// the sample's eight points; true if rho.cpp would reject the sample (shared coordinates, or the two quadrilaterals turn differently)
bool bad_sample(const F2* src, const F2* dst, const unsigned* s, F2* a, F2* b) {
    for (int k = 0; k < 4; k++) {
        a[k] = src[s[k]];
        b[k] = dst[s[k]];
    }
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++)
            if (a[i].x == a[j].x || a[i].y == a[j].y) return true;
    auto turn = [](const F2* P, int p, int q, int r) {
        const float l0 = P[p].y - P[q].y, l1 = P[q].x - P[p].x, l2 = P[p].x * P[q].y - P[p].y * P[q].x;
        return l0 * P[r].x + l1 * P[r].y + l2;
    };
    static const int T[4][3] = {{0, 1, 2}, {0, 1, 3}, {2, 3, 0}, {2, 3, 1}};
    for (const auto& t : T)
        if ((((int)turn(a, t[0], t[1], t[2])) ^ ((int)turn(b, t[0], t[1], t[2]))) < 0) return true;
    return false;
}

// H (h33 = 1) through four correspondences: Gauss-Jordan elimination with partial pivoting on [A | b], binary32 (documented deviation
// from rho.cpp's hand-reduced elimination: oracle/rho_oracle.cpp header)
bool four_point_h(const F2* a, const F2* b, float* H) {
    float m[8][9];
    for (int k = 0; k < 4; k++) {
        const float x = a[k].x, y = a[k].y, X = b[k].x, Y = b[k].y;
        const float top[9] = {x, y, 1, 0, 0, 0, -(x * X), -(y * X), X};
        const float bot[9] = {0, 0, 0, x, y, 1, -(x * Y), -(y * Y), Y};
        std::memcpy(m[2 * k], top, sizeof(top));
        std::memcpy(m[2 * k + 1], bot, sizeof(bot));
    }
    for (int col = 0; col < 8; col++) {
        int piv = col;
        for (int r = col + 1; r < 8; r++)
            if (std::fabs(m[r][col]) > std::fabs(m[piv][col])) piv = r;
        if (!(std::fabs(m[piv][col]) > 0.0f)) return false;
        if (piv != col) {
            float tmp[9];
            std::memcpy(tmp, m[col], sizeof(tmp));
            std::memcpy(m[col], m[piv], sizeof(tmp));
            std::memcpy(m[piv], tmp, sizeof(tmp));
        }
        for (int r = 0; r < 8; r++) {
            if (r == col) continue;
            const float f = m[r][col] / m[col][col];
            for (int j = col; j < 9; j++) m[r][j] = m[r][j] - f * m[col][j];
        }
    }
    bool finite = true;
    for (int i = 0; i < 8; i++) {
        H[i] = m[i][8] / m[i][i];
        finite &= std::isfinite(H[i]);
    }
    H[8] = 1.0f;
    return finite;
}

struct Sprt {
    double eps = 0.1, delta = 0.01, A = 0, onInlier = 0, onOutlier = 0;
    void design() {
        const double C = (1 - delta) * std::log((1 - delta) / (1 - eps)) + delta * std::log(delta / eps);
        const double K = 25.0 * C / 1.0 + 1;
        double cur = K, last;
        unsigned trips = 0;
        do {
            last = cur;
            cur = K + std::log(cur);
        } while ((cur - last > 1.5e-8) && (++trips < 10));
        A = cur;
        onOutlier = (1.0 - delta) / (1.0 - eps);
        onInlier = delta / eps;
    }
};

inline bool bit(const unsigned long long* row, int i) { return (row[i >> 6] >> (i & 63)) & 1ull; }

// final Levenberg-Marquardt refinement over the inliers (rho.cpp refine()): binary32, sequential sums in index order
struct Refiner {
    const F2* src;
    const F2* dst;
    const std::vector<int>& inl;   // inlier indices, ascending
    float sse(const float* H, float* JtJ, float* Jte) const {
        float S = 0.0f;
        if (JtJ) std::fill(JtJ, JtJ + 64, 0.0f);
        if (Jte) std::fill(Jte, Jte + 8, 0.0f);
        for (int i : inl) {
            const float x = src[i].x, y = src[i].y, X = dst[i].x, Y = dst[i].y;
            const float W = H[6] * x + H[7] * y + 1.0f;
            const float iW = std::fabs(W) > FLT_EPSILON ? 1.0f / W : 0.0f;
            const float u = (H[0] * x + H[1] * y + H[2]) * iW;
            const float v = (H[3] * x + H[4] * y + H[5]) * iW;
            const float eu = u - X, ev = v - Y;
            S += eu * eu + ev * ev;
            if (!JtJ && !Jte) continue;
            const float ju[8] = {x * iW, y * iW, iW, 0, 0, 0, -u * x * iW, -u * y * iW};
            const float jv[8] = {0, 0, 0, x * iW, y * iW, iW, -v * x * iW, -v * y * iW};
            if (Jte)
                for (int r = 0; r < 8; r++) Jte[r] += eu * ju[r] + ev * jv[r];
            if (JtJ)
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c <= r; c++) JtJ[r * 8 + c] += ju[r] * ju[c] + jv[r] * jv[c];
        }
        return S;
    }
    static bool cholesky(const float* A, float lambda, float* L) {
        const float scale = lambda + 1.0f;
        for (int r = 0; r < 8; r++)
            for (int c = 0; c <= r; c++) {
                float v = A[r * 8 + c];
                if (r == c) v *= scale;
                for (int k = 0; k < c; k++) v -= L[r * 8 + k] * L[c * 8 + k];
                if (r == c) {
                    if (!(v > 0.0f)) return false;
                    L[r * 8 + r] = std::sqrt(v);
                } else {
                    L[r * 8 + c] = v / L[c * 8 + c];
                }
            }
        return true;
    }
    static void solve(const float* L, const float* rhs, float* x) {
        float y[8];
        for (int r = 0; r < 8; r++) {
            float v = rhs[r];
            for (int k = 0; k < r; k++) v -= L[r * 8 + k] * y[k];
            y[r] = v / L[r * 8 + r];
        }
        for (int r = 7; r >= 0; r--) {
            float v = y[r];
            for (int k = r + 1; k < 8; k++) v -= L[k * 8 + r] * x[k];
            x[r] = v / L[r * 8 + r];
        }
    }
    void run(float* H) const {
        float JtJ[64], Jte[8], L[64], step[8], trial[9];
        float lambda = 100.0f;
        float S = sse(H, JtJ, Jte);
        for (int it = 0; it < 100; it++) {
            while (!cholesky(JtJ, lambda, L)) lambda *= 2.0f;
            solve(L, Jte, step);
            for (int k = 0; k < 8; k++) trial[k] = H[k] - step[k];
            trial[8] = 1.0f;
            const float St = sse(trial, nullptr, nullptr);
            const float actual = S - St;
            float predicted = 0.0f;
            for (int k = 0; k < 8; k++) predicted += step[k] * (lambda * step[k] + Jte[k]);
            const float gain = std::fabs(predicted) < FLT_EPSILON ? actual : actual / predicted;
            if (gain < 0.25f) {
                lambda *= 8;
                if (lambda > 1000.0f / FLT_EPSILON) break;
            } else if (gain > 0.75f) {
                lambda *= 0.5f;
            }
            if (gain > 0) {
                std::memcpy(H, trial, sizeof(trial));
                S = sse(H, JtJ, Jte);
            }
        }
    }
};

}  // namespace

// returns 1 (model found; H_host filled, mask_dev filled if non-null) or 0
int find_homography_rho_device(const float* src_dev, const float* dst_dev, int n, double thr, int max_iters, double confidence, double* H_host,
                               uint8_t* mask_dev, hipStream_t s) {
    ThreadCtx& c = ctx();
    const F2* dS = reinterpret_cast<const F2*>(src_dev);
    const F2* dD = reinterpret_cast<const F2*>(dst_dev);
    std::vector<F2> src(n), dst(n);
    HIP_CHECK(hipMemcpyAsync(src.data(), dS, (size_t)n * sizeof(F2), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(dst.data(), dD, (size_t)n * sizeof(F2), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    const unsigned N = (unsigned)n;
    const float maxD = (float)(thr <= 0 ? 3 : thr);
    const float maxDsq = maxD * maxD;
    unsigned maxI = (unsigned)std::max(max_iters, 1);
    const unsigned rConvg = maxI;
    const double beta = 0.35;

    // PROSAC / non-randomness tables
    Prosac ps;
    ps.rng.seed(~(uint64_t)0);
    ps.phNum = 4;
    ps.phEndI = 1;
    {
        double numer = 1, denom = 1;
        for (unsigned i = 0; i < 4; i++) {
            numer *= 4 - i;
            denom *= N - i;
        }
        ps.phEndFpI = rConvg * numer / denom;
    }
    unsigned phMax = N, phNumInl = 0;
    std::vector<unsigned> nonRandom(N + 1, 0);
    {
        const double spread = std::sqrt(beta * (1.0 - beta)) * 1.645;
        for (unsigned k = 5; k <= N; k++) nonRandom[k] = (unsigned)std::ceil(4 + k * beta + std::sqrt((double)k) * spread);
    }
    Sprt sprt;
    sprt.design();

    const int words = (n + 63) / 64;
    const int first_batch = 256, max_batch = 2048;
    float* models_dev = c.alloc_n<float>((size_t)max_batch * 9);
    unsigned long long* bits_dev = c.alloc_n<unsigned long long>((size_t)max_batch * words);
    std::vector<float> models((size_t)max_batch * 9);
    std::vector<unsigned long long> bits((size_t)max_batch * words);
    std::vector<Prosac> before(max_batch);     // generator state before each speculated iteration
    std::vector<int> slot(max_batch);          // row of the iteration's model in `models` / `bits`, or -1 (no model: rejected sample)

    float bestH[9] = {0};
    // rho.cpp keeps two inlier arrays, `curr.inl` and `best.inl`, and SWAPS them when a model becomes the best one. evaluateModelSPRT writes
    // only the flags of the points it tested, so behind an early SPRT stop an array still holds what an earlier model left there. The two
    // bit rows below are those arrays (zero at the start of a run), prefix writes and swap included: a model that the SPRT rejected can
    // still become the best one (isBestModel() compares the counts only), and then the stale flags are part of the state (the
    // non-randomness walk, the refinement and the returned mask read them).
    std::vector<unsigned long long> curRow(words, 0), bestRow(words, 0);
    unsigned bestCount = 0;
    unsigned it = 0;
    bool first = true;
    // rho.cpp: `for(ctrl.i = 0; ctrl.i < arg.maxI || ctrl.i < 100; ctrl.i++)` - never fewer than 100 iterations, however far the
    // confidence bound has pulled maxI down
    auto limit = [&]() { return std::max(maxI, 100u); };
    while (it < limit()) {
        // ---- speculate the samples of iterations it .. it + B - 1 under the current pool limit
        const int Bcap = first ? first_batch : max_batch;
        first = false;
        int B = 0, rows = 0;
        for (; B < Bcap && it + B < limit(); B++) {
            before[B] = ps;
            const unsigned i = it + B;
            if (i >= ps.phEndI && ps.phNum < phMax) {
                ps.phNum++;
                const double next = (ps.phEndFpI * ps.phNum) / (ps.phNum - 4);
                ps.phEndI += (unsigned)std::ceil(next - ps.phEndFpI);
                ps.phEndFpI = next;
            }
            unsigned pick[4];
            if (i > ps.phEndI) {
                draw(ps.rng, 4, pick, ps.phNum);
            } else {
                draw(ps.rng, 3, pick, ps.phNum - 1);
                pick[3] = ps.phNum - 1;
            }
            F2 a[4], b[4];
            slot[B] = -1;
            if (bad_sample(src.data(), dst.data(), pick, a, b)) continue;
            if (!four_point_h(a, b, &models[(size_t)rows * 9])) continue;
            slot[B] = rows++;
        }
        if (rows > 0) {
            HIP_CHECK(hipMemcpyAsync(models_dev, models.data(), (size_t)rows * 9 * sizeof(float), hipMemcpyHostToDevice, s));
            {
                KernelTimer timer("rho_score", s);
                hipLaunchKernelGGL(rho_inlier_bits_kernel, dim3(std::min(ceil_div(words * 64, 256), 1024), rows), dim3(256), 0, s, dS, dD, n, (const float*)models_dev,
                                   maxDsq, bits_dev, words);
            }
            HIP_CHECK(hipMemcpyAsync(bits.data(), bits_dev, (size_t)rows * words * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        // ---- replay rho.cpp's loop over the speculated iterations
        int b = 0;
        bool redraw = false;
        for (; b < B && it < limit(); b++, it++) {
            if (slot[b] < 0) continue;
            const unsigned long long* row = &bits[(size_t)slot[b] * words];
            // SPRT walk over the points in order
            double lambda = 1.0;
            unsigned count = 0, tested = 0;
            bool good = true;
            for (; tested < N && good; tested++) {
                const bool in = bit(row, (int)tested);
                count += in;
                lambda *= in ? sprt.onInlier : sprt.onOutlier;
                good = lambda <= sprt.A;
            }
            {   // curr.inl[0 .. tested) = this model's flags; the rest keeps its old content
                const unsigned full = tested >> 6, rest = tested & 63;
                std::memcpy(curRow.data(), row, (size_t)full * sizeof(unsigned long long));
                if (rest) {
                    const unsigned long long m = (1ull << rest) - 1;
                    curRow[full] = (curRow[full] & ~m) | (row[full] & m);
                }
            }
            // updateSPRT()
            if (good) {
                if (count > bestCount) {
                    sprt.eps = (double)count / N;
                    sprt.design();
                }
            } else {
                const double nd = (double)count / tested;
                if (nd > 0 && std::fabs(sprt.delta - nd) / sprt.delta > 0.1) {
                    sprt.delta = nd;
                    sprt.design();
                }
            }
            // isBestModel(): the counts alone decide, accepted by the SPRT or not
            if (count > bestCount) {
                std::memcpy(bestH, &models[(size_t)slot[b] * 9], sizeof(bestH));
                curRow.swap(bestRow);
                bestCount = count;
                maxI = iteration_bound(confidence, (double)bestCount / N, 4, maxI);   // updateBounds()
                // nStarOptimize(): the shortest prefix of the (quality-ordered) points whose inlier share beats the whole set's
                unsigned best_n = N, bestInl = bestCount, testInl = bestCount;
                for (unsigned test_n = N; test_n > 20 && testInl; test_n--) {
                    if ((uint64_t)testInl * best_n > (uint64_t)bestInl * test_n) {
                        if (testInl < nonRandom[test_n]) break;
                        best_n = test_n;
                        bestInl = testInl;
                    }
                    testInl -= bit(bestRow.data(), (int)test_n - 1) ? 1 : 0;
                }
                if ((uint64_t)bestInl * phMax > (uint64_t)phNumInl * best_n) {
                    if (phMax != best_n) redraw = true;   // the samples after this iteration were drawn under another pool limit
                    phMax = best_n;
                    phNumInl = bestInl;
                    maxI = iteration_bound(confidence, (double)phNumInl / phMax, 4, maxI);
                }
                if (redraw) {
                    b++;
                    it++;
                    break;
                }
            }
        }
        if (b < B) ps = before[b];   // the generator goes back to where the replay stopped (redraw, or the bound fell below the batch)
    }

    const bool ok = bestCount >= 4;
    std::vector<uint8_t> hmask(n, 0);
    if (ok) {
        std::vector<int> inl;
        inl.reserve(bestCount);
        for (int i = 0; i < n; i++)
            if (bit(bestRow.data(), i)) {
                inl.push_back(i);
                hmask[i] = 1;
            }
        if (bestCount > 4) Refiner{src.data(), dst.data(), inl}.run(bestH);
    }
    for (int i = 0; i < 9; i++) H_host[i] = ok ? (double)bestH[i] : 0.0;
    if (mask_dev) {
        HIP_CHECK(hipMemcpyAsync(mask_dev, hmask.data(), (size_t)n, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    return ok ? 1 : 0;
}

}  // namespace apds
