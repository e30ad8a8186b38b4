// csrc/akaze_level_strips.hip — one AKAZE evolution level step on register strips (a translation unit of its own: the sixteen
// fully unrolled variants take minutes to compile, beside akaze_filters.hip instead of inside it).
//
// Replaces the OpenCV work behind /root/reference/feature_extraction/src/lib.rs:64-79 (AKAZE::create(...).detect_and_compute) for
// the large levels of the nonlinear scale space: Gaussian smoothing, Scharr derivatives, Perona-Malik g2 conductivity and the first
// FED steps of a level in one pass. Float contract as in akaze_filters.hip (one IEEE binary32 operation per source operation,
// -ffp-contract=off).
#include "akaze.h"

namespace apds {

__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
#define APDS_BOFS(p) p = bofs(p, bstride)
__device__ __forceinline__ float dpp_next(float v) {   // lane i <- lane i + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_prev(float v) {   // lane i <- lane i - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) { return dpp_next(v); }   // wave_shl:1 (lane 63: unspecified, a halo lane)
__device__ __forceinline__ float dpp_from_prev_lane(float v) { return dpp_prev(v); }   // wave_shr:1 (lane 0: unspecified, a halo lane)
struct NldSteps {
    float v[8];
};
#ifndef APDS_STRIP_WAVES
#define APDS_STRIP_WAVES 4
#endif
#ifndef APDS_STRIP_RB
#define APDS_STRIP_RB 16
#endif
#ifndef APDS_STRIP_WIDE_FROM
#define APDS_STRIP_WIDE_FROM 3   // level_strip_kernel<S> with S >= this may use up to 168 VGPRs (three waves per SIMD) instead of spilling at 128
#endif

// ---- a whole level step on register strips (the large levels) ---------------------------------------------------------------------
// smooth_flow_strip_kernel and nld_strip_kernel in one pass: Lsmooth = Gaussian(Lt_prev), conductivity = g2(Scharr(Lsmooth)) and the
// level's first S <= 4 FED steps, with the conductivity (and the Lt rows between the two) never leaving the registers: 12 + 16 B per
// pixel of HBM traffic become 4 (read, + halo) + 8 (Lsmooth, Lt) — the large levels are bandwidth-bound. A wave owns 64 columns x
// (RB + 2 (S + 3)) rows; lanes [S + 3, 61 - S) and strip rows [S + 3, S + 3 + RB) are final. Every value is produced by the operations
// of smooth_flow_kernel / nld_point in the same order. The waves on the image border (BORDER) load through clamped coordinates
// (replicate: what the Gaussian wants), take the reflected neighbour in the Scharr pass (at x = 0 the left neighbour is the right one,
// at y = 0 the row above is the row below, ...) and apply nld_strip's border rules in the FED steps: no separate pass over the frame
// of border tiles (which cost 17 - 33 us per large level behind the strips of smooth_flow_strip_kernel).
template <int S, int RB, bool BORDER, bool FLOW_OUT>
__device__ __forceinline__ void level_strip(const float* __restrict__ src, float* __restrict__ smooth, float* __restrict__ flow_out, float* __restrict__ Lnew,
                                            int w, int h, const GaussTaps& taps, float k2inv, const NldSteps& steps, int gx0, int y0) {
    constexpr int H = S + 3;              // halo: S (FED) + 1 (Scharr ring) + 2 (Gaussian)
    constexpr int R = RB + 2 * H;         // loaded rows; strip row 0 is image row y0 - H
    constexpr int RS = RB + 2 * S + 2;    // Lsmooth rows: index q is strip row q + 2
    constexpr int RF = RB + 2 * S;        // conductivity / Lt rows: index r is strip row r + 3 (image row y0 - S + r)
    const int lane = threadIdx.x & 63;
    const int gx = gx0 + lane;
    const int ys = y0 - H;
    const int plane_bytes = w * h * 4;
    float sr[R];
    {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, plane_bytes, 0x00020000);
        const int cx4 = 4 * (BORDER ? clampi(gx, w) : gx);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row4 = (BORDER ? clampi(ys + r, h) : ys + r) * w * 4;
            sr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, cx4, row4, 0));
        }
    }
    float t[RF];
#pragma unroll
    for (int r = 0; r < RF; r++) t[r] = sr[r + 3];
#pragma unroll
    for (int r = 0; r < R; r++) {         // Gaussian rows, in place
        const float v = sr[r];
        const float l1 = dpp_prev(v), r1 = dpp_next(v);
        const float l2 = dpp_prev(l1), r2 = dpp_next(r1);
        float acc = taps.k[0] * v;
        acc += taps.k[1] * (l1 + r1);
        acc += taps.k[2] * (l2 + r2);
        sr[r] = acc;
    }
    const bool mine = lane >= H && lane < 64 - H && gx < w;   // (gx >= 0 for these lanes: gx0 >= -H)
    float rd[RS], rs[RS];
    {
        const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(smooth, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < RS; q++) {    // Gaussian columns -> Lsmooth row q; its Scharr row terms
            const int r = q + 2;
            float acc = taps.k[0] * sr[r];
            acc += taps.k[1] * (sr[r - 1] + sr[r + 1]);
            acc += taps.k[2] * (sr[r - 2] + sr[r + 2]);
            if (q >= S + 1 && q < S + 1 + RB) {
                const int gy = y0 + q - (S + 1);
                if (mine && (!BORDER || gy < h)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc), rs_sm, 4 * gx, gy * w * 4, 0);
            }
            float l = dpp_prev(acc), rr = dpp_next(acc);
            if (BORDER) {                 // reflect-101 in x
                const float l0 = l;
                l = gx == 0 ? rr : l;
                rr = gx == w - 1 ? l0 : rr;
            }
            rd[q] = rr - l;
            float a = 10.0f * acc;
            a += 3.0f * (l + rr);
            rs[q] = a;
        }
    }
    float f[RF];
#pragma unroll
    for (int r = 0; r < RF; r++) {
        const int q = r + 1;
        float rdu = rd[q - 1], rdd = rd[q + 1], rsu = rs[q - 1], rsd = rs[q + 1];
        if (BORDER) {                     // reflect-101 in y (wave-uniform)
            const int gy = ys + r + 3;
            if (gy == 0) {
                rdu = rd[q + 1];
                rsu = rs[q + 1];
            }
            if (gy == h - 1) {
                rdd = rd[q - 1];
                rsd = rs[q - 1];
            }
        }
        float ax = 10.0f * rd[q];
        ax += 3.0f * (rdu + rdd);
        const float ay = rsd - rsu;
        f[r] = 1.0f / (1.0f + ((ax * ax + ay * ay) * k2inv));
    }
    if (FLOW_OUT) {
        const __amdgpu_buffer_rsrc_t rs_fl = __builtin_amdgcn_make_buffer_rsrc(flow_out, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int r = S; r < S + RB; r++) {
            const int gy = y0 + r - S;
            if (mine && (!BORDER || gy < h)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, f[r]), rs_fl, 4 * gx, gy * w * 4, 0);
        }
    }
    // FED steps: nld_strip on the rows / lanes that are left (strip row 3 is row 0 here)
    const int yf = y0 - S;
    const bool flux_x_inside = gx >= 0 && gx + 1 <= w - 1;
    const bool edge_col = gx == 0 || gx == w - 1;
#pragma unroll
    for (int j = 1; j <= S; j++) {
        const float tau = steps.v[j - 1];
        float qprev = (f[j - 1] + f[j]) * (t[j] - t[j - 1]);
        if (BORDER && !(yf + j - 1 >= 0 && yf + j <= h - 1)) qprev = 0.0f;
#pragma unroll
        for (int r = j; r < RF - j; r++) {
            const float tc = t[r];
            const float d = dpp_from_next_lane(tc) - tc;
            float P = (f[r] + dpp_from_next_lane(f[r])) * d;
            if (BORDER && !flux_x_inside) P = 0.0f;
            float q = (f[r] + f[r + 1]) * (t[r + 1] - tc);
            if (BORDER && !(yf + r >= 0 && yf + r + 1 <= h - 1)) q = 0.0f;
            float sum = P - dpp_from_prev_lane(P);
            sum = sum + q;
            sum = sum - qprev;
            float out = tc + sum * tau;
            if (BORDER && edge_col && (yf + r == 0 || yf + r == h - 1)) out = tc;
            qprev = q;
            t[r] = out;
        }
    }
    if (mine) {
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(Lnew, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int r = S; r < S + RB; r++) {
            const int gy = y0 + r - S;
            if (!BORDER || gy < h) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t[r]), rn, 4 * gx, gy * w * 4, 0);
        }
    }
}

template <int S, int RB, bool FLOW_OUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S >= APDS_STRIP_WIDE_FROM ? 3 : APDS_STRIP_WAVES, 8)))
void level_strip_kernel(const float* __restrict__ src, float* __restrict__ smooth, float* __restrict__ flow_out, float* __restrict__ Lnew, int w, int h,
                        GaussTaps taps, const float* __restrict__ kptr, NldSteps steps, int strips, int nwaves, size_t bstride, ForkSignal sig) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_FORK_SIGNAL(sig);
    APDS_BOFS(src);
    APDS_BOFS(smooth);
    APDS_BOFS(Lnew);
    APDS_BOFS(kptr);
    if (FLOW_OUT) APDS_BOFS(flow_out);
    constexpr int H = S + 3, VW = 64 - 2 * H;
    const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (id >= nwaves) return;                   // the kernel has no barriers
    const int band = __builtin_amdgcn_readfirstlane(id / strips);
    const int strip = id - band * strips;
    const int gx0 = strip * VW - H, y0 = band * RB;
    const float k = *kptr;
    const float k2inv = 1.0f / (k * k);
    const bool border = gx0 < 0 || gx0 + 64 > w || y0 - H < 0 || y0 + RB + H > h;
    if (border) level_strip<S, RB, true, FLOW_OUT>(src, smooth, flow_out, Lnew, w, h, taps, k2inv, steps, gx0, y0);
    else level_strip<S, RB, false, FLOW_OUT>(src, smooth, flow_out, Lnew, w, h, taps, k2inv, steps, gx0, y0);
}

// Lsmooth, conductivity and the level's first `nsteps` (1 .. 4) FED steps on register strips, borders included: src -> smooth, Lnew
// (and flow_out when the caller continues with more steps). False when the level does not fit 32-bit byte offsets.
template <int S>
static void level_strip_launch(const float* src, float* smooth, float* flow_out, float* Lnew, int w, int h, const GaussTaps& taps, const float* kptr,
                               const NldSteps& st, hipStream_t s, const Batch& b) {
    constexpr int RB = APDS_STRIP_RB;
    const int strips = ceil_div(w, 64 - 2 * (S + 3)), nwaves = strips * ceil_div(h, RB);
    const dim3 grid(ceil_div(nwaves, 4), 1, b.n);
    const ForkSignal sig = ctx().take_fork_signal();
    if (flow_out)
        hipLaunchKernelGGL((level_strip_kernel<S, RB, true>), grid, dim3(256), 0, s, src, smooth, flow_out, Lnew, w, h, taps, kptr, st, strips, nwaves, b.stride, sig);
    else
        hipLaunchKernelGGL((level_strip_kernel<S, RB, false>), grid, dim3(256), 0, s, src, smooth, flow_out, Lnew, w, h, taps, kptr, st, strips, nwaves, b.stride, sig);
}
bool launch_level_strips(const float* src, float* smooth, float* flow_out, float* Lnew, int w, int h, const GaussTaps& taps, const float* kptr,
                         const float* step_sizes, int nsteps, hipStream_t s, const Batch& b) {
    if (nsteps < 1 || nsteps > 4 || (size_t)w * h >= ((size_t)1 << 29) || w < 2 || h < 2) return false;
    NldSteps st{};
    for (int i = 0; i < nsteps; i++) st.v[i] = step_sizes[i];
    switch (nsteps) {
        case 1: level_strip_launch<1>(src, smooth, flow_out, Lnew, w, h, taps, kptr, st, s, b); break;
        case 2: level_strip_launch<2>(src, smooth, flow_out, Lnew, w, h, taps, kptr, st, s, b); break;
        case 3: level_strip_launch<3>(src, smooth, flow_out, Lnew, w, h, taps, kptr, st, s, b); break;
        default: level_strip_launch<4>(src, smooth, flow_out, Lnew, w, h, taps, kptr, st, s, b); break;
    }
    return true;
}

}  // namespace apds
