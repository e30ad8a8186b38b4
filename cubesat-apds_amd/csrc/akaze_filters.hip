// csrc/akaze_filters.hip — AKAZE nonlinear scale space and determinant-of-Hessian planes on gfx950.
//
// Replaces the OpenCV work behind /root/reference/feature_extraction/src/lib.rs:64-79 (AKAZE::create(...)
// .detect_and_compute): gray conversion, Gaussian smoothing, Scharr derivatives, Perona-Malik g2 conductivity,
// FED diffusion steps, 2x2 half-sampling and the scaled second derivatives.
//
// All planes are f32, row-major, pitch == width, resident in HBM. These kernels are HBM-bandwidth bound:
// every stencil stages its input tile (+halo, border already applied) in LDS once, so each plane is read
// from HBM once per pass and written once; rows are read as contiguous 256-byte wave accesses.
//
// Float contract (identical to oracle/akaze_oracle.cpp, and what makes keypoint sets bit-equal): one IEEE
// binary32 operation per source operation, no FMA contraction (-ffp-contract=off), correctly rounded
// division; symmetric taps as  acc = k0*c; acc += k1*(lo1+hi1); ...  antisymmetric taps as  hi - lo.
#include "akaze.h"
#include "config.h"

namespace apds {

__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// APDS_BOFS(p): the plane of image blockIdx.z of a batched launch (akaze.h: bofs)
#define APDS_BOFS(p) p = bofs(p, bstride)
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- a1.1: BGR(A)/gray u8 -> f32 in [0,1] ----------------------------------------------------------
__global__ void gray_kernel(const uint8_t* __restrict__ img, int rows, int cols, int channels, size_t stride, float* __restrict__ out, size_t img_bstride,
                            size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    img += (size_t)blockIdx.z * img_bstride;
    APDS_BOFS(out);
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols) return;
    const uint8_t* p = img + (size_t)y * stride;
    int g;
    if (channels == 1) {
        g = p[x];
    } else if (channels == 4) {
        const uint32_t v = reinterpret_cast<const uint32_t*>(p)[x];   // B,G,R,A little endian (rows are 4-byte aligned)
        g = (int)((v & 0xFF) * 3735u + ((v >> 8) & 0xFF) * 19235u + ((v >> 16) & 0xFF) * 9798u + (1u << 14)) >> 15;
    } else {
        const uint8_t* q = p + (size_t)x * 3;
        g = (q[0] * 3735 + q[1] * 19235 + q[2] * 9798 + (1 << 14)) >> 15;
    }
    out[(size_t)y * cols + x] = (float)g * (float)(1.0 / 255.0);
}

// ---- separable symmetric blur, BORDER_REPLICATE (GaussianBlur) --------------------------------------
static constexpr int TW = 64, TH = 16;   // output tile, 256 threads

template <int R>
__global__ __launch_bounds__(256) void gauss_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, GaussTaps taps, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(src);
    APDS_BOFS(dst);
    __shared__ float s_src[(TH + 2 * R) * (TW + 2 * R)];
    __shared__ float s_tmp[(TH + 2 * R) * TW];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    constexpr int SW = TW + 2 * R, SH = TH + 2 * R;
    for (int i = threadIdx.x; i < SW * SH; i += 256) {
        const int ly = i / SW, lx = i - ly * SW;
        const int gx = clampi(x0 - R + lx, w), gy = clampi(y0 - R + ly, h);
        s_src[i] = src[(size_t)gy * w + gx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SH * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const float* p = &s_src[ly * SW + lx + R];
        float acc = taps.k[0] * p[0];
#pragma unroll
        for (int j = 1; j <= R; j++) acc += taps.k[j] * (p[-j] + p[j]);
        s_tmp[i] = acc;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < TH; ly += 4) {
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            const float* p = &s_tmp[(ly + R) * TW + lx];
            float acc = taps.k[0] * p[0];
#pragma unroll
            for (int j = 1; j <= R; j++) acc += taps.k[j] * (p[-j * TW] + p[j * TW]);
            dst[(size_t)gy * w + gx] = acc;
        }
    }
}

// ---- dilated 3x3 derivative pair, BORDER_REFLECT_101 ------------------------------------------------
// Lx = colsmooth(row: hi - lo), Ly = coldiff(row: smooth); taps at -s, 0, +s; smooth = {kside, kmid, kside}.
// MODE 0: write Lx, Ly.  MODE 1: write flow = 1/(1 + (Lx^2+Ly^2)/k^2)  (k read from device memory).
// MODE 2: write |grad| and atomically track its maximum over interior pixels (contrast factor pass).
template <int MODE>
__global__ __launch_bounds__(256) void deriv_pair_kernel(const float* __restrict__ src, float* __restrict__ outA, float* __restrict__ outB,
                                                         int w, int h, int s, float kside, float kmid, const float* __restrict__ kptr,
                                                         unsigned int* __restrict__ hmax_bits, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(src);
    APDS_BOFS(outA);
    APDS_BOFS(outB);
    APDS_BOFS(kptr);
    APDS_BOFS(hmax_bits);
    extern __shared__ float smem[];
    const int SW = TW + 2 * s, SH = TH + 2 * s;
    float* s_src = smem;                 // SH x SW
    float* s_rd = s_src + SH * SW;       // SH x TW   row pass, derivative
    float* s_rs = s_rd + SH * TW;        // SH x TW   row pass, smooth
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    for (int i = threadIdx.x; i < SW * SH; i += 256) {
        const int ly = i / SW, lx = i - ly * SW;
        const int gx = reflect101(x0 - s + lx, w), gy = reflect101(y0 - s + ly, h);
        s_src[i] = src[(size_t)gy * w + gx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SH * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const float* p = &s_src[ly * SW + lx + s];
        const float lo = p[-s], hi = p[s];
        s_rd[i] = hi - lo;
        float acc = kmid * p[0];
        acc += kside * (lo + hi);
        s_rs[i] = acc;
    }
    __syncthreads();
    float k2inv = 0.f;
    if (MODE == 1) {
        const float k = *kptr;
        k2inv = 1.0f / (k * k);
    }
    float local_max = 0.f;
    const int lx = threadIdx.x & 63;
    for (int ly = threadIdx.x >> 6; ly < TH; ly += 4) {
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            const int c = (ly + s) * TW + lx;
            float ax = kmid * s_rd[c];
            ax += kside * (s_rd[c - s * TW] + s_rd[c + s * TW]);
            const float ay = s_rs[c + s * TW] - s_rs[c - s * TW];
            const size_t o = (size_t)gy * w + gx;
            if (MODE == 0) {
                outA[o] = ax;
                outB[o] = ay;
            } else if (MODE == 1) {
                outA[o] = 1.0f / (1.0f + ((ax * ax + ay * ay) * k2inv));
            } else {
                const float m = sqrtf(ax * ax + ay * ay);
                outA[o] = m;
                if (gx >= 1 && gx < w - 1 && gy >= 1 && gy < h - 1) local_max = fmaxf(local_max, m);
            }
        }
    }
    if (MODE == 2) {
        // non-negative floats order like their bit patterns. One atomic per BLOCK, and only if it can raise the maximum
        // (a stale read of *hmax_bits is never larger than the truth, so no update is lost).
        __shared__ unsigned int s_wmax[4];
        unsigned int bits = __float_as_uint(local_max);
        for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned int)__shfl_xor((int)bits, off));
        if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = bits;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int b = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
            if (b > *reinterpret_cast<volatile unsigned int*>(hmax_bits)) atomicMax(hmax_bits, b);
        }
    }
}

// ---- per level: Lsmooth = Gaussian(5 taps, replicate) of the level's start image, and the PM-g2 conductivity from the
// Scharr gradient of Lsmooth (reflect-101), in one pass: the start image is read once, Lsmooth never comes back from HBM
// (16 -> 12 B/pixel and one launch less per level). Lsmooth is evaluated on the tile + 1 ring at in-image positions; a ring
// position outside the image is, under reflect-101, an in-image position at most one pixel inside the border, i.e. part of
// the same region, so the gradient stencil simply indexes it through the reflected coordinate. Arithmetic per value is
// that of gauss_kernel<2> followed by deriv_pair_kernel<1>.
#ifndef APDS_SF_THREADS
#define APDS_SF_THREADS 1024
#endif
static constexpr int FW = 64, FH = 32, FNT = APDS_SF_THREADS;

// Persistent blocks: each block walks a strided list of tiles of its XCD's band and issues the loads of its NEXT tile (into
// registers) before it computes the current one, so the HBM latency of a tile hides behind the three LDS passes of the previous
// tile instead of being paid once per tile per block.
__global__ __launch_bounds__(FNT) void smooth_flow_kernel(const float* __restrict__ src, float* __restrict__ smooth, float* __restrict__ flow, int w, int h,
                                                          GaussTaps taps, const float* __restrict__ kptr, int tiles_x, int ntiles, int txi, int tyi, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(src);
    APDS_BOFS(smooth);
    APDS_BOFS(flow);
    APDS_BOFS(kptr);
    constexpr int SW = FW + 6, SH = FH + 6;      // start image, halo 3 (= ring 1 + Gaussian radius 2), replicate on load
    constexpr int TWD = FW + 2;                  // row-pass / Lsmooth width: tile + ring 1
    constexpr int MH = FH + 2;
    constexpr int NL = (SW * SH + FNT - 1) / FNT;
    __shared__ float s_src[SH * SW];
    __shared__ float s_tmp[SH * TWD];
    __shared__ float s_sm[MH * TWD];
    // Item i of this block -> tile. Whole level: blocks go round-robin over the XCDs (gridDim.x is a multiple of 8) and each XCD
    // walks its own band of tiles. Frame only (txi > 0: tiles [1, txi) x [1, tyi) belong to smooth_flow_strip_kernel): the frame's
    // tiles are enumerated compactly — top rows, then (left column + right columns) of the middle rows, then the bottom rows — and
    // dealt out one per block, so that no block gets a whole column of them.
    const int tiles_y = ntiles / tiles_x;
    const int n_top = tiles_x, per_mid = 1 + (tiles_x - txi), n_mid = (tyi - 1) * per_mid;
    const int n_frame = n_top + n_mid + (tiles_y - tyi) * tiles_x;
    const int xcd = blockIdx.x & 7, per = txi > 0 ? (int)gridDim.x : (int)(gridDim.x >> 3);
    const int band = (ntiles + 7) >> 3;
    const int t_end = txi > 0 ? n_frame : min(ntiles, (xcd + 1) * band);
    int t = txi > 0 ? (int)blockIdx.x : xcd * band + (int)(blockIdx.x >> 3);
    auto tile_of = [&](int i) {
        if (txi <= 0) return i;
        if (i < n_top) return i;
        i -= n_top;
        if (i < n_mid) {
            const int row = i / per_mid, c = i - row * per_mid;
            return (1 + row) * tiles_x + (c == 0 ? 0 : txi + c - 1);
        }
        return tyi * tiles_x + (i - n_mid);
    };
    float v[NL];
    auto issue_loads = [&](int item) {
        const int tile = tile_of(item);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * FW, y0 = ty * FH;
        const bool inside = x0 >= 3 && y0 >= 3 && x0 + FW + 3 <= w && y0 + FH + 3 <= h;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = min((int)threadIdx.x + k * FNT, SW * SH - 1);
            const int ly = i / SW, lx = i - ly * SW;
            v[k] = inside ? src[(size_t)(y0 - 3 + ly) * w + (x0 - 3 + lx)] : src[(size_t)clampi(y0 - 3 + ly, h) * w + clampi(x0 - 3 + lx, w)];
        }
    };
    if (t < t_end) issue_loads(t);
    const float k = *kptr;
    const float k2inv = 1.0f / (k * k);
    const float kside = 3.0f, kmid = 10.0f;      // unnormalised Scharr, as launch_flow passes them
    for (int tn; t < t_end; t = tn) {
        tn = t + per;
        const int tile = tile_of(t);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * FW, y0 = ty * FH;
        // the tile and its halo inside the image (all but the outermost tiles): no bounds tests, no reflected coordinates
        const bool inside = x0 >= 3 && y0 >= 3 && x0 + FW + 3 <= w && y0 + FH + 3 <= h;
#pragma unroll
        for (int kk = 0; kk < NL; kk++) {
            const int i = threadIdx.x + kk * FNT;
            if (i < SW * SH) s_src[i] = v[kk];
        }
        __syncthreads();
        if (tn < t_end) issue_loads(tn);
        for (int i = threadIdx.x; i < SH * TWD; i += FNT) {
            const int ly = i / TWD, lx = i - ly * TWD;
            const float* p = &s_src[ly * SW + lx + 2];
            float acc = taps.k[0] * p[0];
            acc += taps.k[1] * (p[-1] + p[1]);
            acc += taps.k[2] * (p[-2] + p[2]);
            s_tmp[i] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < MH * TWD; i += FNT) {
            const int ly = i / TWD, lx = i - ly * TWD;
            const float* p = &s_tmp[(ly + 2) * TWD + lx];
            float acc = taps.k[0] * p[0];
            acc += taps.k[1] * (p[-TWD] + p[TWD]);
            acc += taps.k[2] * (p[-2 * TWD] + p[2 * TWD]);
            s_sm[i] = acc;
            const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
            if (lx >= 1 && lx <= FW && ly >= 1 && ly <= FH && (inside || (gx < w && gy < h))) smooth[(size_t)gy * w + gx] = acc;
        }
        __syncthreads();
        auto flow_point = [&](const float* r0, const float* r1, const float* r2, int cxm, int cx, int cxp) {
            const float rd0 = r0[cxp] - r0[cxm], rd1 = r1[cxp] - r1[cxm], rd2 = r2[cxp] - r2[cxm];
            float ax = kmid * rd1;
            ax += kside * (rd0 + rd2);
            float rs0 = kmid * r0[cx];
            rs0 += kside * (r0[cxm] + r0[cxp]);
            float rs2 = kmid * r2[cx];
            rs2 += kside * (r2[cxm] + r2[cxp]);
            const float ay = rs2 - rs0;
            return 1.0f / (1.0f + ((ax * ax + ay * ay) * k2inv));
        };
        if (inside) {
            for (int i = threadIdx.x; i < FW * FH; i += FNT) {
                const int ly = i / FW, lx = i - ly * FW;
                const float* r1 = &s_sm[(ly + 1) * TWD];
                flow[(size_t)(y0 + ly) * w + (x0 + lx)] = flow_point(r1 - TWD, r1, r1 + TWD, lx, lx + 1, lx + 2);
            }
        } else {
            for (int i = threadIdx.x; i < FW * FH; i += FNT) {
                const int ly = i / FW, lx = i - ly * FW;
                const int gx = x0 + lx, gy = y0 + ly;
                if (gx >= w || gy >= h) continue;
                const float* r1 = &s_sm[(ly + 1) * TWD];
                if (gx >= 1 && gx <= w - 2 && gy >= 1 && gy <= h - 2) {   // only the image's outermost pixels need reflected coordinates
                    flow[(size_t)gy * w + gx] = flow_point(r1 - TWD, r1, r1 + TWD, lx, lx + 1, lx + 2);
                    continue;
                }
                const int cxm = reflect101(gx - 1, w) - (x0 - 1), cxp = reflect101(gx + 1, w) - (x0 - 1);
                flow[(size_t)gy * w + gx] = flow_point(&s_sm[(reflect101(gy - 1, h) - (y0 - 1)) * TWD], r1,
                                                       &s_sm[(reflect101(gy + 1, h) - (y0 - 1)) * TWD], cxm, lx + 1, cxp);
            }
        }
        // the next iteration's s_src stores are ordered after this iteration's row pass by the two barriers above; its row pass
        // (s_tmp) and column pass (s_sm) are each a barrier away from this iteration's readers of those planes
    }
}

// The same pass on register strips, for the tiles that lie inside the image together with their halo (no clamping, no reflection):
// a wave owns 64 columns x (RB + 6) rows, one column per lane; the 5-tap row pass takes its neighbours through DP wave shifts, the
// column pass and the Scharr rows come out of the lane's own registers. Every value is produced by the same operations in the same
// order as in smooth_flow_kernel. Lanes [3, 61) and strip rows [3, 3 + RB) are final.
__device__ __forceinline__ float dpp_next(float v) {   // lane i <- lane i + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_prev(float v) {   // lane i <- lane i - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
static constexpr int SF_RB = 16, SF_VW = 58;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void smooth_flow_strip_kernel(const float* __restrict__ src, float* __restrict__ smooth, float* __restrict__ flow, int w, int h, GaussTaps taps,
                              const float* __restrict__ kptr, int rx0, int ry0, int rx1, int ry1, int strips, int nwaves, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(src);
    APDS_BOFS(smooth);
    APDS_BOFS(flow);
    APDS_BOFS(kptr);
    constexpr int RB = SF_RB, R = RB + 6;
    const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (id >= nwaves) return;                   // wave-uniform; no barriers in this kernel
    const int band = __builtin_amdgcn_readfirstlane(id / strips);
    const int strip = id - band * strips;
    const int lane = threadIdx.x & 63;
    const int gx = rx0 + strip * SF_VW - 3 + lane;
    const int y0 = ry0 + band * RB, ys = y0 - 3;
    const int plane_bytes = w * h * 4;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, plane_bytes, 0x00020000);
    float s[R];
#pragma unroll
    for (int r = 0; r < R; r++) s[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_src, 4 * gx, (ys + r) * w * 4, 0));
    const float k = *kptr;
    const float k2inv = 1.0f / (k * k);
    const float kside = 3.0f, kmid = 10.0f;
#pragma unroll
    for (int r = 0; r < R; r++) {               // row pass, in place
        const float v = s[r];
        const float l1 = dpp_prev(v), r1 = dpp_next(v);
        const float l2 = dpp_prev(l1), r2 = dpp_next(r1);
        float acc = taps.k[0] * v;
        acc += taps.k[1] * (l1 + r1);
        acc += taps.k[2] * (l2 + r2);
        s[r] = acc;
    }
    float sm[RB + 2];                           // Lsmooth rows y0 - 1 .. y0 + RB
#pragma unroll
    for (int q = 0; q < RB + 2; q++) {
        const int r = q + 2;
        float acc = taps.k[0] * s[r];
        acc += taps.k[1] * (s[r - 1] + s[r + 1]);
        acc += taps.k[2] * (s[r - 2] + s[r + 2]);
        sm[q] = acc;
    }
    const bool mine = lane >= 3 && lane < 61 && gx < rx1;
    float rd[RB + 2], rs[RB + 2];
#pragma unroll
    for (int q = 0; q < RB + 2; q++) {
        const float l = dpp_prev(sm[q]), r = dpp_next(sm[q]);
        rd[q] = r - l;
        float a = kmid * sm[q];
        a += kside * (l + r);
        rs[q] = a;
    }
    if (mine) {
        const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(smooth, 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_fl = __builtin_amdgcn_make_buffer_rsrc(flow, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int q = 1; q <= RB; q++) {
            const int gy = y0 + q - 1;
            if (gy < ry1) {
                float ax = kmid * rd[q];
                ax += kside * (rd[q - 1] + rd[q + 1]);
                const float ay = rs[q + 1] - rs[q - 1];
                const float fl = 1.0f / (1.0f + ((ax * ax + ay * ay) * k2inv));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sm[q]), rs_sm, 4 * gx, gy * w * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, fl), rs_fl, 4 * gx, gy * w * 4, 0);
            }
        }
    }
}

// ---- a1.1 + a1.2 + the gradient pass of a1.3 on register strips (large images) ---------------------------------------------------
// image -> gray (registers) -> { 9-tap Gaussian -> Lt[0] } and { 5-tap Gaussian -> unnormalised Scharr -> |grad| + its maximum over
// the interior }: one read of the image instead of gray_kernel, gauss_kernel<4>, gauss_kernel<2> and deriv_pair_kernel<2> with
// their three f32 round trips. A wave owns 64 columns x (RB + 8) rows, one column per lane; x +- 1 .. 4 come through chained DP
// wave shifts (the same shifted values serve both Gaussians). Borders: both Gaussians replicate THE GRAY IMAGE, which clamped
// loads give for free; the Scharr pass reflects (101) its input, the 5-tap result, so at the image edge the missing neighbour is
// the opposite one (lane x + 1 for x - 1 at x = 0, row 1 for row -1, ...): selects in the BORDER waves only. Same operations in
// the same order as the four kernels it replaces. Lanes [4, 60) and strip rows [4, 4 + RB) are final.
#ifndef APDS_BS_RB
#define APDS_BS_RB 8
#endif
#ifndef APDS_BS_WAVES
#define APDS_BS_WAVES 4
#endif
static constexpr int BS_RB = APDS_BS_RB, BS_VW = 56;

template <int CH, bool BORDER>
__device__ __forceinline__ void base_strip(const uint8_t* __restrict__ img, int w, int h, int stride, const GaussTaps& g16, const GaussTaps& g10,
                                           float* __restrict__ Lt0, float* __restrict__ modg, unsigned int* __restrict__ hmax_bits, int want_modg, int gx0,
                                           int y0) {
    constexpr int RB = BS_RB, R = RB + 8;
    const int lane = threadIdx.x & 63;
    const int gx = gx0 + lane, ys = y0 - 4;
    const int cx = BORDER ? clampi(gx, w) : gx;
    const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, h * stride, 0x00020000);
    float h9[R], h5[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int row = (BORDER ? clampi(ys + r, h) : ys + r) * stride;   // scalar
        int gi;
        if (CH == 4) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_img, 4 * cx, row, 0);   // B,G,R,A little endian
            gi = (int)((v & 0xFF) * 3735u + ((v >> 8) & 0xFF) * 19235u + ((v >> 16) & 0xFF) * 9798u + (1u << 14)) >> 15;
        } else if (CH == 3) {
            const int b = __builtin_amdgcn_raw_buffer_load_b8(r_img, 3 * cx, row, 0), g = __builtin_amdgcn_raw_buffer_load_b8(r_img, 3 * cx + 1, row, 0),
                      rr = __builtin_amdgcn_raw_buffer_load_b8(r_img, 3 * cx + 2, row, 0);
            gi = (b * 3735 + g * 19235 + rr * 9798 + (1 << 14)) >> 15;
        } else {
            gi = __builtin_amdgcn_raw_buffer_load_b8(r_img, cx, row, 0);
        }
        const float v = (float)gi * (float)(1.0 / 255.0);
        const float l1 = dpp_prev(v), r1 = dpp_next(v);
        const float l2 = dpp_prev(l1), r2 = dpp_next(r1);
        const float l3 = dpp_prev(l2), r3 = dpp_next(r2);
        const float l4 = dpp_prev(l3), r4 = dpp_next(r3);
        const float p1 = l1 + r1, p2 = l2 + r2;
        float a = g16.k[0] * v;
        a += g16.k[1] * p1;
        a += g16.k[2] * p2;
        a += g16.k[3] * (l3 + r3);
        a += g16.k[4] * (l4 + r4);
        h9[r] = a;
        float b5 = g10.k[0] * v;
        b5 += g10.k[1] * p1;
        b5 += g10.k[2] * p2;
        h5[r] = b5;
    }
    const bool mine = lane >= 4 && lane < 60 && gx < w;
    const int plane_bytes = w * h * 4;
    {
        const __amdgpu_buffer_rsrc_t r_lt = __builtin_amdgcn_make_buffer_rsrc(Lt0, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < RB; q++) {
            const int r = q + 4;
            float a = g16.k[0] * h9[r];
#pragma unroll
            for (int j = 1; j <= 4; j++) a += g16.k[j] * (h9[r - j] + h9[r + j]);
            if (mine && (!BORDER || y0 + q < h)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a), r_lt, 4 * gx, (y0 + q) * w * 4, 0);
        }
    }
    if (!want_modg) return;                     // uniform
    float rd[RB + 2], rs[RB + 2];               // image rows y0 - 1 .. y0 + RB
    const float kside = 3.0f, kmid = 10.0f;     // unnormalised Scharr, as launch_kcontrast passes them
#pragma unroll
    for (int q = 0; q < RB + 2; q++) {
        const int r = q + 3;
        float sm = g10.k[0] * h5[r];
        sm += g10.k[1] * (h5[r - 1] + h5[r + 1]);
        sm += g10.k[2] * (h5[r - 2] + h5[r + 2]);
        float lo = dpp_prev(sm), hi = dpp_next(sm);
        if (BORDER) {
            const float lo0 = lo;
            if (gx == 0) lo = hi;               // reflect-101 of the Gaussian's output
            if (gx == w - 1) hi = lo0;
        }
        rd[q] = hi - lo;
        float a = kmid * sm;
        a += kside * (lo + hi);
        rs[q] = a;
    }
    float local_max = 0.f;
    const __amdgpu_buffer_rsrc_t r_mg = __builtin_amdgcn_make_buffer_rsrc(modg, 0, plane_bytes, 0x00020000);
#pragma unroll
    for (int q = 1; q <= RB; q++) {
        const int gy = y0 + q - 1;
        float rd_up = rd[q - 1], rd_dn = rd[q + 1], rs_up = rs[q - 1], rs_dn = rs[q + 1];
        if (BORDER) {
            if (gy == 0) {
                rd_up = rd[q + 1];
                rs_up = rs[q + 1];
            }
            if (gy == h - 1) {
                rd_dn = rd[q - 1];
                rs_dn = rs[q - 1];
            }
        }
        float ax = kmid * rd[q];
        ax += kside * (rd_up + rd_dn);
        const float ay = rs_dn - rs_up;
        const float m = sqrtf(ax * ax + ay * ay);
        if (mine && (!BORDER || gy < h)) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, m), r_mg, 4 * gx, gy * w * 4, 0);
            if (!BORDER || (gx >= 1 && gx < w - 1 && gy >= 1 && gy < h - 1)) local_max = fmaxf(local_max, m);
        }
    }
    // non-negative floats order like their bit patterns; one atomic per wave, and only if it can raise the maximum
    unsigned int bits = __float_as_uint(local_max);
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned int)__shfl_xor((int)bits, off));
    if (lane == 0 && bits > *reinterpret_cast<volatile unsigned int*>(hmax_bits)) atomicMax(hmax_bits, bits);
}

template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(APDS_BS_WAVES, 8)))
void base_strip_kernel(const uint8_t* __restrict__ img, int w, int h, int stride, GaussTaps g16, GaussTaps g10, float* __restrict__ Lt0,
                       float* __restrict__ modg, unsigned int* __restrict__ hmax_bits, int want_modg, int strips, int nwaves, size_t img_bstride, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    img += (size_t)blockIdx.z * img_bstride;
    APDS_BOFS(Lt0);
    APDS_BOFS(modg);
    APDS_BOFS(hmax_bits);
    const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (id >= nwaves) return;                   // wave-uniform; no barriers in this kernel
    const int band = __builtin_amdgcn_readfirstlane(id / strips);
    const int strip = id - band * strips;
    const int gx0 = strip * BS_VW - 4, y0 = band * BS_RB;
    const bool border = gx0 < 0 || gx0 + 64 > w || y0 - 4 < 0 || y0 + BS_RB + 4 > h;
    if (border) base_strip<CH, true>(img, w, h, stride, g16, g10, Lt0, modg, hmax_bits, want_modg, gx0, y0);
    else base_strip<CH, false>(img, w, h, stride, g16, g10, Lt0, modg, hmax_bits, want_modg, gx0, y0);
}

// ---- contrast factor: 300-bin histogram of |grad|/hmax over interior pixels, 70th percentile -----------
// SUB sub-histograms per block (lane & (SUB - 1) picks one): gradient magnitudes crowd the low bins, so with one histogram most of a
// wave's 64 LDS atomics hit a handful of addresses and serialise; each thread reads four consecutive pixels of a row per step.
template <int SUB>
__global__ __launch_bounds__(256) void kcontrast_hist_kernel(const float* __restrict__ modg, int w, int h, const unsigned int* __restrict__ hmax_bits,
                                                              int* __restrict__ hist, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(modg);
    APDS_BOFS(hmax_bits);
    APDS_BOFS(hist);
    constexpr int PITCH = 301;                          // odd pitch: the copies of a bin sit in different banks
    __shared__ int s_hist[SUB * PITCH];
    for (int i = threadIdx.x; i < SUB * PITCH; i += 256) s_hist[i] = 0;
    __syncthreads();
    const float hmax = __uint_as_float(*hmax_bits);
    if (hmax != 0.0f) {
        const float scale = 299.0f / hmax;
        const int cw = w - 2, ch = h - 2;
        int* mine = s_hist + (threadIdx.x & (SUB - 1)) * PITCH;
        const int groups = (cw + 3) >> 2;               // 4-pixel groups per interior row
        const long long total = (long long)groups * ch;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int y = (int)(i / groups), x = (int)(i - (long long)y * groups) * 4;
            const float* p = modg + (size_t)(y + 1) * w + (x + 1);
            const int n = min(4, cw - x);
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = k < n ? p[k] : 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < n) atomicAdd(&mine[(int)(v[k] * scale)], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 300; i += 256) {
        int sum = 0;
#pragma unroll
        for (int sidx = 0; sidx < SUB; sidx++) sum += s_hist[sidx * PITCH + i];
        if (sum) atomicAdd(&hist[i], sum);
    }
}
// histogram -> kcontrast and its per-octave values k, k*0.75, (k*0.75)*0.75, ... One wave: lane l owns bins [5l, 5l + 5), a wave scan gives
// every bin the count of the bins in front of it, and the first bin b >= 1 with (bins 1 .. b-1) >= the 70 % threshold names k - the value the
// sequential loop of the reference (nldiffusion_functions.cpp compute_k_percentile) stops at. (Round 4 measured this inside the histogram
// kernel, done by the block that draws the last of gridDim.x tickets: 1024 same-address device-scope atomics behind 1024 fences took the
// histogram kernel from 43 to 81 us - 139 beside a Hessian kernel. A launch of its own, 300 loads wide instead of 300 loads deep, is cheaper.)
__global__ __launch_bounds__(64) void kcontrast_finish_kernel(const int* __restrict__ hist, const unsigned int* __restrict__ hmax_bits, int w, int h,
                                                               float* __restrict__ k_oct, int n_oct, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(hist);
    APDS_BOFS(hmax_bits);
    APDS_BOFS(k_oct);
    const int lane = threadIdx.x;
    const float hmax = __uint_as_float(*hmax_bits);
    constexpr int nbins = 300;
    int v[5], mine = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int bin = lane * 5 + j;
        v[j] = bin < nbins ? hist[bin] : 0;
        if (bin >= 1) mine += v[j];
    }
    const int h0 = __shfl(v[0], 0);
    int incl = mine;    // inclusive scan over lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    float k = 0.03f;
    if (hmax != 0.0f && w > 2 && h > 2) {
        const int total = (w - 2) * (h - 2);
        const int nthreshold = (int)((total - h0) * 0.7f);
        int before = incl - mine;   // bins 1 .. 5 * lane - 1
        int hit = nbins;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int bin = lane * 5 + j;
            if (bin >= 1 && bin < nbins && before >= nthreshold && hit == nbins) hit = bin;
            if (bin >= 1) before += v[j];
        }
        const unsigned long long any = __ballot(hit < nbins);
        if (any) {
            const int b = __shfl(hit, __ffsll((long long)any) - 1);
            k = hmax * b / nbins;
        }
    }
    if (lane == 0)
        for (int o = 0; o < n_oct; o++) {
            k_oct[o] = k;
            k *= 0.75f;
        }
}

// ---- explicit FED diffusion steps: Lnew = Lt + step_size * div(c grad Lt), several steps per pass (temporal blocking) ----
// Every intermediate value is exactly what a one-step-per-launch formulation produces (same expression, same inputs), so
// results do not depend on how the steps are grouped; HBM traffic and the launch count of the latency-bound small octaves
// drop with the group size.
static constexpr int T2W = 64, T2H = 32;

__device__ __forceinline__ float nld_point(const float* __restrict__ st, const float* __restrict__ sf, int c, int pitch, int x, int y, int w, int h,
                                           float step_size) {
    const float tc = st[c], fc = sf[c];
    const bool top = y == 0, bot = y == h - 1, left = x == 0, right = x == w - 1;
    const float xp = (fc + sf[c + 1]) * (st[c + 1] - tc);
    const float xm = (fc + sf[c - 1]) * (st[c - 1] - tc);
    const float yp = (fc + sf[c + pitch]) * (st[c + pitch] - tc);
    const float ym = (fc + sf[c - pitch]) * (st[c - pitch] - tc);
    float step_r;
    if ((top || bot) && (left || right)) step_r = 0.0f;
    else if (top) step_r = xp + xm + yp;
    else if (bot) step_r = xp + xm + ym;
    else if (left) step_r = xp + yp + ym;
    else if (right) step_r = xm + yp + ym;
    else step_r = xp + xm + yp + ym;
    return tc + step_r * step_size;
}

// S FED steps in one pass: inputs are loaded with an S-pixel halo; step j is evaluated on the tile + (S - j) rings, ping-ponging
// between two LDS planes; the last step writes the tile. Only positions inside the image are evaluated; a border pixel's
// out-of-image neighbour is read (whatever LDS holds) but never used, exactly as in the single-step kernel.
struct NldSteps {
    float v[8];
};

// a point with all four neighbours inside the image: nld_point without the border cases (same expression, same order)
__device__ __forceinline__ float nld_point_interior(const float* __restrict__ st, const float* __restrict__ sf, int c, int pitch, float step_size) {
    const float tc = st[c], fc = sf[c];
    const float xp = (fc + sf[c + 1]) * (st[c + 1] - tc);
    const float xm = (fc + sf[c - 1]) * (st[c - 1] - tc);
    const float yp = (fc + sf[c + pitch]) * (st[c + pitch] - tc);
    const float ym = (fc + sf[c - pitch]) * (st[c - pitch] - tc);
    return tc + (xp + xm + yp + ym) * step_size;
}

// two horizontally adjacent interior points at once: the same IEEE operations per point, issued as packed-f32 instructions
// (v_pk_add_f32 / v_pk_mul_f32 work on register pairs), so the per-item index arithmetic is also paid once per two points
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lds_pair(const float* p) { return f32x2{p[0], p[1]}; }
__device__ __forceinline__ f32x2 nld_pair_interior(const float* __restrict__ st, const float* __restrict__ sf, int c, int pitch, float step_size) {
    const f32x2 t = lds_pair(st + c), f = lds_pair(sf + c);
    const f32x2 xp = (f + lds_pair(sf + c + 1)) * (lds_pair(st + c + 1) - t);
    const f32x2 xm = (f + lds_pair(sf + c - 1)) * (lds_pair(st + c - 1) - t);
    const f32x2 yp = (f + lds_pair(sf + c + pitch)) * (lds_pair(st + c + pitch) - t);
    const f32x2 ym = (f + lds_pair(sf + c - pitch)) * (lds_pair(st + c - pitch) - t);
    return t + (xp + xm + yp + ym) * step_size;
}

// INTERIOR: the tile with its halo lies inside the image and touches no image border (block-uniform; all but the outermost
// tiles): no clamping, no in-image tests, no border cases — a third of the instructions of the general path.
template <int S, int NT, bool INTERIOR>
__device__ __forceinline__ void nld_multi_tile(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ Lnew, int w, int h,
                                               const NldSteps& steps, float* s_f, float* s_a, float* s_b, int x0, int y0) {
    constexpr int SW = T2W + 2 * S, SH = T2H + 2 * S;
    {
        constexpr int NL = (SW * SH + NT - 1) / NT;
        float va[NL], vf[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {   // every load of both tiles in flight before the first LDS store
            const int i = min((int)threadIdx.x + k * NT, SW * SH - 1);
            const int ly = i / SW, lx = i - ly * SW;
            const size_t o = INTERIOR ? (size_t)(y0 + ly) * w + (x0 + lx) : (size_t)clampi(y0 + ly, h) * w + clampi(x0 + lx, w);
            va[k] = Lt[o];
            vf[k] = Lf[o];
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = threadIdx.x + k * NT;
            if (i < SW * SH) {
                s_a[i] = va[k];
                s_f[i] = vf[k];
            }
        }
    }
    __syncthreads();
    const float* src = s_a;
    float* dst = s_b;
#pragma unroll
    for (int j = 1; j < S; j++) {
        const int rw = SW - 2 * j, rh = SH - 2 * j;   // region of this step: local [j, SW - j) x [j, SH - j)
        if constexpr (INTERIOR) {
            const int hw = rw / 2;   // the region's width is even: two points per item
            for (int i = threadIdx.x; i < hw * rh; i += NT) {
                const int ry = i / hw, rx = i - ry * hw;
                const int c = (ry + j) * SW + 2 * rx + j;
                const f32x2 v = nld_pair_interior(src, s_f, c, SW, steps.v[j - 1]);
                dst[c] = v.x;
                dst[c + 1] = v.y;
            }
        } else {
            for (int i = threadIdx.x; i < rw * rh; i += NT) {
                const int ry = i / rw, rx = i - ry * rw;
                const int lx = rx + j, ly = ry + j;
                const int gx = x0 + lx, gy = y0 + ly;
                if (gx >= 0 && gx < w && gy >= 0 && gy < h) dst[ly * SW + lx] = nld_point(src, s_f, ly * SW + lx, SW, gx, gy, w, h, steps.v[j - 1]);
            }
        }
        __syncthreads();
        const float* t = src;
        src = dst;
        dst = const_cast<float*>(t);
    }
    if constexpr (INTERIOR) {
        for (int i = threadIdx.x; i < (T2W / 2) * T2H; i += NT) {
            const int ly = i / (T2W / 2), lx = 2 * (i - ly * (T2W / 2));
            const f32x2 v = nld_pair_interior(src, s_f, (ly + S) * SW + lx + S, SW, steps.v[S - 1]);
            float* o = &Lnew[(size_t)(y0 + S + ly) * w + (x0 + S + lx)];
            o[0] = v.x;
            o[1] = v.y;
        }
    } else {
        for (int i = threadIdx.x; i < T2W * T2H; i += NT) {
            const int ly = i / T2W, lx = i - ly * T2W;
            const int gx = x0 + S + lx, gy = y0 + S + ly;
            if (gx < w && gy < h) Lnew[(size_t)gy * w + gx] = nld_point(src, s_f, (ly + S) * SW + lx + S, SW, gx, gy, w, h, steps.v[S - 1]);
        }
    }
}

template <int S, int NT>
__global__ __launch_bounds__(NT) void nld_multi_kernel(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ Lnew, int w, int h,
                                                        NldSteps steps, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(Lt);
    APDS_BOFS(Lf);
    APDS_BOFS(Lnew);
    constexpr int SW = T2W + 2 * S, SH = T2H + 2 * S;
    __shared__ float s_f[SH * SW];
    __shared__ float s_a[SH * SW];
    __shared__ float s_b[SH * SW];
    const int x0 = blockIdx.x * T2W - S, y0 = blockIdx.y * T2H - S;   // global coordinate of local (0, 0)
    // every evaluated point (local [1, SW-1) x [1, SH-1) at the first step) has its four neighbours inside the image
    if (x0 >= 0 && y0 >= 0 && x0 + SW <= w && y0 + SH <= h) nld_multi_tile<S, NT, true>(Lt, Lf, Lnew, w, h, steps, s_f, s_a, s_b, x0, y0);
    else nld_multi_tile<S, NT, false>(Lt, Lf, Lnew, w, h, steps, s_f, s_a, s_b, x0, y0);
}

// ---- a whole level in one launch (the small, latency-bound levels) -----------------------------------------------
// Lsmooth = Gaussian(Lt_prev), conductivity = g2(Scharr(Lsmooth)), then all of the level's FED steps, in ONE kernel per level: a
// 512^2 or 1024^2 level used to take four or five dependent launches of ~10 us each, nearly all of it launch and drain latency.
// A block owns a 32x32 tile of the level and recomputes everything it needs around it (temporal blocking): with S FED steps the
// start image is loaded with a halo of S + 3 (S for the steps, 1 for the Scharr ring, 2 for the Gaussian), and every stage is
// evaluated on the region the following stages read. Per value the operations (and their order) are those of smooth_flow_kernel
// and nld_point, so the planes are bit-identical to the unfused path. Four LDS planes with one pitch and one origin: local (0, 0)
// is the global pixel (x0 - S - 3, y0 - S - 3).
//   out-of-image positions: the start image is loaded through clamped coordinates (BORDER_REPLICATE, what the Gaussian wants);
//   the Scharr ring of a border pixel goes through reflect-101 coordinates (in-image, inside the region); a FED step of a border
//   pixel is nld_point's border case, which never uses the out-of-image neighbour. Positions outside the image are evaluated
//   with whatever LDS holds and never read by an in-image position.
struct LevelSteps {
    int n;
    float v[32];
};
static constexpr int LFT = 32;              // output tile (square)
static constexpr int LF_MAX_STEPS = 29;     // 1024 threads: 3 x 3 patches cover (32 + 2 * 29)^2; 3 planes x 96^2 floats = 108 KB of LDS
static constexpr int LF_MAX_STEPS_512 = 17; // 512 threads: 3 x 3 patches cover (32 + 2 * 17)^2

__device__ __forceinline__ int div_small(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }   // floor(i / d): exact for i < 2^15, d < 2^7

__device__ __forceinline__ float pm_g2_point(const float* r0, const float* r1, const float* r2, int cxm, int cx, int cxp, float k2inv) {
    const float kside = 3.0f, kmid = 10.0f;
    const float rd0 = r0[cxp] - r0[cxm], rd1 = r1[cxp] - r1[cxm], rd2 = r2[cxp] - r2[cxm];
    float ax = kmid * rd1;
    ax += kside * (rd0 + rd2);
    float rs0 = kmid * r0[cx];
    rs0 += kside * (r0[cxm] + r0[cxp]);
    float rs2 = kmid * r2[cx];
    rs2 += kside * (r2[cxm] + r2[cxp]);
    const float ay = rs2 - rs0;
    return 1.0f / (1.0f + ((ax * ax + ay * ay) * k2inv));
}

// The FED steps of level_fused_kernel: every thread keeps a PW x PW patch of the (32 + 2S)^2 region in registers for ALL steps and
// exchanges only its values with the four neighbouring patches through LDS (two planes, alternating: one barrier per step). The
// conductivity enters only as the pair sums f[x] + f[x+1] of the fluxes
//   P[x] = (f[x] + f[x+1]) * (t[x+1] - t[x])        Q[r] = (f[r] + f[r+1]) * (t[r+1] - t[r])
// which are loop constants in registers; a step is  t += (((P[x] - P[x-1]) + Q[r]) - Q[r-1]) * step  — nld_point's four terms exactly
// (see nld_strip: a - b == -(b - a), (-a) * b == -(a * b), x + (-y) == x - y). A flux that crosses the image border, or lies outside
// the image, gets the pair sum 0: the reference drops that term, and adding or subtracting a zero product changes nothing (the sum's
// first operand is never -0 ...); out-of-image positions therefore keep their (finite, replicated) start value. The four image
// corners keep their value (the reference's step is 0 there). Nothing shrinks: every step evaluates the whole region, and the
// values within j positions of the region's edge are wrong after step j — the tile in the middle is S positions away.
template <int PW, int NT>
__device__ __forceinline__ void fed_patches(const float* __restrict__ s_t, const float* __restrict__ s_f, float* __restrict__ e0, float* __restrict__ e1,
                                            int P, int S, const LevelSteps& steps, int gx0, int gy0, int w, int h, bool inside,
                                            float* __restrict__ Lnew, bool shrink, float* __restrict__ half) {
    const int R = P - 6;                       // the region is local [3, P - 3)^2
    const int TG = (R + PW - 1) / PW;          // patches per row
    const int tid = threadIdx.x;
    // (every wave an 8 x 8 block of patches instead of two rows of them - so that the shrinking zone idles waves in both directions - with
    // the planes' pitch padded against bank conflicts was built and measured in round 4: 24.8 / 25.7 / 30.6 / 32.0 us for the four 512^2
    // levels against 23.0 / 24.7 / 29.6 / 32.0: a step of the one-block-per-CU levels is bound by its barrier and LDS round trips, not by
    // issue slots. Removed.)
    const int ty = div_small(tid, 1.0f / (float)TG), tx = tid - ty * TG;
    const bool active = ty < TG;
    const int x0 = 3 + PW * tx, y0 = 3 + PW * ty;   // (patches of the last row / column may reach up to PW - 1 positions past the region)
    float t[PW][PW], fx[PW][PW + 1], fy[PW + 1][PW];
    unsigned corner = 0;
    if (active) {
#pragma unroll
        for (int a = 0; a < PW; a++)
#pragma unroll
            for (int b = 0; b < PW; b++) t[a][b] = s_t[(y0 + a) * P + x0 + b];
#pragma unroll
        for (int a = 0; a < PW; a++)
#pragma unroll
            for (int b = 0; b <= PW; b++) {     // flux between (x0 + b - 1, y0 + a) and (x0 + b, y0 + a)
                const float* q = &s_f[(y0 + a) * P + x0 + b];
                float v = q[-1] + q[0];
                if (!inside) {
                    const int gx = gx0 + x0 + b, gy = gy0 + y0 + a;
                    if (gx - 1 < 0 || gx >= w || gy < 0 || gy >= h) v = 0.0f;
                }
                fx[a][b] = v;
            }
#pragma unroll
        for (int a = 0; a <= PW; a++)
#pragma unroll
            for (int b = 0; b < PW; b++) {      // flux between (x0 + b, y0 + a - 1) and (x0 + b, y0 + a)
                const float* q = &s_f[(y0 + a) * P + x0 + b];
                float v = q[-P] + q[0];
                if (!inside) {
                    const int gx = gx0 + x0 + b, gy = gy0 + y0 + a;
                    if (gy - 1 < 0 || gy >= h || gx < 0 || gx >= w) v = 0.0f;
                }
                fy[a][b] = v;
            }
        if (!inside) {
#pragma unroll
            for (int a = 0; a < PW; a++)
#pragma unroll
                for (int b = 0; b < PW; b++) {
                    const int gx = gx0 + x0 + b, gy = gy0 + y0 + a;
                    if ((gx == 0 || gx == w - 1) && (gy == 0 || gy == h - 1)) corner |= 1u << (a * PW + b);
                }
        }
    }
    __syncthreads();   // every thread has read its start values: e0 / e1 (the smoothing planes) may be overwritten
    // Shrinking (round 4): the tile is all that leaves the kernel, so after step k only the positions within m = S - 1 - k of it have to be
    // right. A patch computes step k if it reaches into that zone, and stores its values before step k if it computed step k - 1 (its
    // neighbours in the zone read them); the others sit out - whole waves of them skip the step's LDS traffic and arithmetic. What an idle
    // or half-needed patch leaves in LDS is stale, and so is what gets computed from it at positions outside the zone, but a position inside
    // the zone of step k only ever reads positions inside the zone of step k - 1: the tile's values are bit for bit those of the full sweep.
    const int c_lo0 = S + 3, c_hi0 = S + 3 + LFT;
    for (int k = 0; k < S; k++) {
        float* __restrict__ e = (k & 1) ? e1 : e0;
        const int m = shrink ? S - 1 - k : P;
        const bool computes = active && x0 < c_hi0 + m && x0 + PW > c_lo0 - m && y0 < c_hi0 + m && y0 + PW > c_lo0 - m;
        const bool stores = active && (k == 0 || (x0 < c_hi0 + m + 1 && x0 + PW > c_lo0 - m - 1 && y0 < c_hi0 + m + 1 && y0 + PW > c_lo0 - m - 1));
        if (stores) {
#pragma unroll
            for (int a = 0; a < PW; a++)
#pragma unroll
                for (int b = 0; b < PW; b++) e[(y0 + a) * P + x0 + b] = t[a][b];
        }
        __syncthreads();
        if (computes) {
            float lf[PW], rt[PW], up[PW], dn[PW];
#pragma unroll
            for (int a = 0; a < PW; a++) {
                lf[a] = e[(y0 + a) * P + x0 - 1];
                rt[a] = e[(y0 + a) * P + x0 + PW];
            }
#pragma unroll
            for (int b = 0; b < PW; b++) {
                up[b] = e[(y0 - 1) * P + x0 + b];
                dn[b] = e[(y0 + PW) * P + x0 + b];
            }
            float px[PW][PW + 1], qy[PW + 1][PW];
#pragma unroll
            for (int a = 0; a < PW; a++)
#pragma unroll
                for (int b = 0; b <= PW; b++) {
                    const float hi = b == PW ? rt[a] : t[a][b], lo = b == 0 ? lf[a] : t[a][b - 1];
                    px[a][b] = fx[a][b] * (hi - lo);
                }
#pragma unroll
            for (int a = 0; a <= PW; a++)
#pragma unroll
                for (int b = 0; b < PW; b++) {
                    const float hi = a == PW ? dn[b] : t[a][b], lo = a == 0 ? up[b] : t[a - 1][b];
                    qy[a][b] = fy[a][b] * (hi - lo);
                }
            const float step = steps.v[k];
#pragma unroll
            for (int a = 0; a < PW; a++)
#pragma unroll
                for (int b = 0; b < PW; b++) {
                    const float sr = ((px[a][b + 1] - px[a][b]) + qy[a + 1][b]) - qy[a][b];
                    const float v = t[a][b] + sr * step;
                    t[a][b] = (corner >> (a * PW + b)) & 1u ? t[a][b] : v;
                }
        }
    }
    if (half) __syncthreads();   // the last step's neighbour reads of e0 / e1 are done: e0 takes the tile once more
    if (active) {
        const int c_lo = S + 3, c_hi = S + 3 + LFT;
#pragma unroll
        for (int a = 0; a < PW; a++)
#pragma unroll
            for (int b = 0; b < PW; b++) {
                const int lx = x0 + b, ly = y0 + a, gx = gx0 + lx, gy = gy0 + ly;
                if (lx >= c_lo && lx < c_hi && ly >= c_lo && ly < c_hi && gx < w && gy < h) {
                    Lnew[(size_t)gy * w + gx] = t[a][b];
                    if (half) e0[ly * P + lx] = t[a][b];
                }
            }
    }
    if (half) {
        // the last level of an octave: the next octave's start image from the tile, ((a + b) + (c + d)) * 0.25 as half_sample_kernel forms
        // it (tiles start on multiples of 32: the 2 x 2 blocks never straddle tiles); block-uniform branch
        __syncthreads();
        const int c_lo = S + 3, hw = w >> 1, hh = h >> 1;
        for (int i = tid; i < (LFT / 2) * (LFT / 2); i += NT) {
            const int oy = i / (LFT / 2), ox = i - oy * (LFT / 2);
            const int gx = (gx0 + c_lo) / 2 + ox, gy = (gy0 + c_lo) / 2 + oy;
            if (gx < hw && gy < hh) {
                const float* q = &e0[(c_lo + 2 * oy) * P + c_lo + 2 * ox];
                half[(size_t)gy * hw + gx] = ((q[0] + q[1]) + (q[P] + q[P + 1])) * 0.25f;
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void level_fused_kernel(const float* __restrict__ src, float* __restrict__ smooth, float* __restrict__ flow_out,
                                                          const float* __restrict__ flow_in, float* __restrict__ Lnew, int w, int h, GaussTaps taps,
                                                          const float* __restrict__ kptr, LevelSteps steps, size_t bstride, int shrink, float* __restrict__ half,
                                                          ForkSignal sig) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_FORK_SIGNAL(sig);
    APDS_BOFS(src);
    APDS_BOFS(smooth);
    APDS_BOFS(Lnew);
    if (half) APDS_BOFS(half);
    APDS_BOFS(kptr);
    if (flow_out) APDS_BOFS(flow_out);
    if (flow_in) APDS_BOFS(flow_in);
    extern __shared__ __attribute__((aligned(16))) float lf_lds[];
    const int S = steps.n;
    const int P = LFT + 2 * S + 6;
    // three planes: start image | Gaussian rows, then the conductivity | Lsmooth; the FED steps exchange through the first and the last
    float* s_a = lf_lds;
    float* s_b = s_a + P * P;
    float* s_sm = s_b + P * P;
    float* s_f = s_b;
    const int gx0 = blockIdx.x * LFT - S - 3, gy0 = blockIdx.y * LFT - S - 3;
    // block-uniform: the whole region lies inside the image and holds no border pixel
    const bool inside = gx0 >= 0 && gy0 >= 0 && gx0 + P <= w && gy0 + P <= h;
    const int tid = threadIdx.x;
    {
        constexpr int PMAX = LFT + 2 * (NT == 1024 ? LF_MAX_STEPS : LF_MAX_STEPS_512) + 6;
        constexpr int NL = (PMAX * PMAX + NT - 1) / NT;
        const float invP = 1.0f / (float)P;
        float v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {   // all loads in flight before the first LDS store
            const int i = tid + k * NT;
            if (i < P * P) {
                const int ly = div_small(i, invP), lx = i - ly * P;
                v[k] = inside ? src[(size_t)(gy0 + ly) * w + (gx0 + lx)] : src[(size_t)clampi(gy0 + ly, h) * w + clampi(gx0 + lx, w)];
            }
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = tid + k * NT;
            if (i < P * P) s_a[i] = v[k];
        }
    }
    if (flow_in) {   // the caller has run the smoothing pass (its Lsmooth was wanted early): the conductivity comes from HBM
        const int fw = P - 6;
        const float inv = 1.0f / (float)fw;
        for (int i = tid; i < fw * fw; i += NT) {
            const int ry = div_small(i, inv);
            const int ly = 3 + ry, lx = 3 + (i - ry * fw);
            s_f[ly * P + lx] = flow_in[(size_t)clampi(gy0 + ly, h) * w + clampi(gx0 + lx, w)];
        }
        __syncthreads();
    } else {
        const float kc = *kptr;
        const float k2inv = 1.0f / (kc * kc);
        __syncthreads();
        // every stage is evaluated on (its region) x (the image): block-uniform bounds in local coordinates
        const int ix0 = -gx0, ix1 = w - gx0, iy0 = -gy0, iy1 = h - gy0;   // the image is local [ix0, ix1) x [iy0, iy1)
        const int c_lo = S + 3, c_hi = S + 3 + LFT;                        // the tile
        {   // Gaussian rows: rows within 2 of the image (the column pass reads the replicated rows), in-image columns of [2, P - 2)
            const int xl = max(2, ix0), xh = min(P - 2, ix1), yl = max(0, iy0 - 2), yh = min(P, iy1 + 2);
            const int rw = xh - xl, n = rw * (yh - yl);
            const float inv = 1.0f / (float)rw;
            for (int i = tid; i < n; i += NT) {
                const int ry = div_small(i, inv);
                const int ly = yl + ry, lx = xl + (i - ry * rw);
                const float* p = &s_a[ly * P + lx];
                float acc = taps.k[0] * p[0];
                acc += taps.k[1] * (p[-1] + p[1]);
                acc += taps.k[2] * (p[-2] + p[2]);
                s_b[ly * P + lx] = acc;
            }
        }
        __syncthreads();
        {   // Gaussian columns: Lsmooth on the in-image part of [2, P - 2)^2
            const int xl = max(2, ix0), xh = min(P - 2, ix1), yl = max(2, iy0), yh = min(P - 2, iy1);
            const int rw = xh - xl, n = rw * (yh - yl);
            const float inv = 1.0f / (float)rw;
            for (int i = tid; i < n; i += NT) {
                const int ry = div_small(i, inv);
                const int ly = yl + ry, lx = xl + (i - ry * rw);
                const float* p = &s_b[ly * P + lx];
                float acc = taps.k[0] * p[0];
                acc += taps.k[1] * (p[-P] + p[P]);
                acc += taps.k[2] * (p[-2 * P] + p[2 * P]);
                s_sm[ly * P + lx] = acc;
                if (lx >= c_lo && lx < c_hi && ly >= c_lo && ly < c_hi) smooth[(size_t)(gy0 + ly) * w + (gx0 + lx)] = acc;
            }
        }
        __syncthreads();
        {   // conductivity on the in-image part of [3, P - 3)^2
            const int xl = max(3, ix0), xh = min(P - 3, ix1), yl = max(3, iy0), yh = min(P - 3, iy1);
            const int rw = xh - xl, n = rw * (yh - yl);
            const float inv = 1.0f / (float)rw;
            for (int i = tid; i < n; i += NT) {
                const int ry = div_small(i, inv);
                const int ly = yl + ry, lx = xl + (i - ry * rw);
                const int gx = gx0 + lx, gy = gy0 + ly;
                float g;
                if (inside || (gx >= 1 && gx <= w - 2 && gy >= 1 && gy <= h - 2)) {
                    const float* r1 = &s_sm[ly * P];
                    g = pm_g2_point(r1 - P, r1, r1 + P, lx - 1, lx, lx + 1, k2inv);
                } else {   // image border: reflect-101 ring
                    g = pm_g2_point(&s_sm[(reflect101(gy - 1, h) - gy0) * P], &s_sm[ly * P], &s_sm[(reflect101(gy + 1, h) - gy0) * P], reflect101(gx - 1, w) - gx0,
                                    lx, reflect101(gx + 1, w) - gx0, k2inv);
                }
                s_f[ly * P + lx] = g;
                if (flow_out && lx >= c_lo && lx < c_hi && ly >= c_lo && ly < c_hi) flow_out[(size_t)gy * w + gx] = g;
            }
        }
        __syncthreads();
    }
    const int hr = (P - 5) >> 1;   // patches per row with 2 x 2 patches
    if (hr * hr <= NT) fed_patches<2, NT>(s_a, s_f, s_sm, s_a, P, S, steps, gx0, gy0, w, h, inside, Lnew, shrink != 0, half);
    else fed_patches<3, NT>(s_a, s_f, s_sm, s_a, P, S, steps, gx0, gy0, w, h, inside, Lnew, shrink != 0, half);
}

// ---- FED steps on register strips (the large levels) -------------------------------------------------------------
// One wave owns a strip of 64 columns x (RB + 2S) rows of Lt and of the conductivity, one column per lane, all rows in registers
// (fully unrolled, static register indices): no LDS, no barriers. Horizontal neighbours come through DP wave shifts. With
//   P[x] = (f[x] + f[x+1]) * (t[x+1] - t[x])        Q[r] = (f[r] + f[r+1]) * (t[r+1] - t[r])
// the four flux terms of nld_point are xp = P[x], xm = -P[x-1], yp = Q[r], ym = -Q[r-1] EXACTLY: float addition commutes,
// a - b == -(b - a) and (-a) * b == -(a * b) in IEEE arithmetic, and x + (-y) is x - y. So a step costs one P and one Q per point
// (11 instructions) instead of four products. Step j is valid on rows [j, R - j) and lanes [j, 64 - j): the same shrinking halo as
// the LDS kernel. Border handling (BORDER waves only): a flux across the image edge is forced to +0 — exactly the reference's
// dropped term, because the sum's first operand is never -0 (a difference of equal floats is +0 and conductivities are positive),
// so adding +-0 changes nothing — and the four corner pixels, whose step is 0 in the reference, keep their value.
__device__ __forceinline__ float dpp_from_next_lane(float v) {   // lane i <- lane i + 1 (lane 63: unspecified, a halo lane)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));   // wave_shl:1
}
__device__ __forceinline__ float dpp_from_prev_lane(float v) {   // lane i <- lane i - 1 (lane 0: unspecified, a halo lane)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));   // wave_shr:1
}

template <int S, int RB, bool BORDER, bool HALF>
__device__ __forceinline__ void nld_strip(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ Lnew, int w, int h,
                                          const NldSteps& steps, int gx0, int y0, float* __restrict__ half) {
    constexpr int R = RB + 2 * S;
    const int lane = threadIdx.x & 63;
    const int gx = gx0 + lane;
    const int ys = y0 - S;                      // image row of strip row 0
    float t[R], f[R];
    {
        // buffer loads: (descriptor, one lane-offset VGPR shared by every row, the row's byte offset in a scalar register) — plain
        // global loads would keep a 64-bit address pair per row in vector registers
        const int plane_bytes = w * h * 4;
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lt), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lf), 0, plane_bytes, 0x00020000);
        const int cx4 = 4 * (BORDER ? clampi(gx, w) : gx);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row4 = (BORDER ? clampi(ys + r, h) : ys + r) * w * 4;
            t[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, cx4, row4, 0));
            f[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, cx4, row4, 0));
        }
    }
    const bool flux_x_inside = gx >= 0 && gx + 1 <= w - 1;   // the edge between columns gx and gx + 1
    const bool edge_col = gx == 0 || gx == w - 1;
#pragma unroll
    for (int j = 1; j <= S; j++) {
        const float tau = steps.v[j - 1];
        float qprev = (f[j - 1] + f[j]) * (t[j] - t[j - 1]);
        if (BORDER && !(ys + j - 1 >= 0 && ys + j <= h - 1)) qprev = 0.0f;
#pragma unroll
        for (int r = j; r < R - j; r++) {
            const float tc = t[r];
            const float d = dpp_from_next_lane(tc) - tc;
            float P = (f[r] + dpp_from_next_lane(f[r])) * d;
            if (BORDER && !flux_x_inside) P = 0.0f;
            float q = (f[r] + f[r + 1]) * (t[r + 1] - tc);   // t[r + 1] is still this step's input
            if (BORDER && !(ys + r >= 0 && ys + r + 1 <= h - 1)) q = 0.0f;
            float sr = P - dpp_from_prev_lane(P);             // xp + xm
            sr = sr + q;                                      // + yp
            sr = sr - qprev;                                  // + ym
            float out = tc + sr * tau;
            if (BORDER && edge_col && (ys + r == 0 || ys + r == h - 1)) out = tc;
            qprev = q;
            t[r] = out;
        }
    }
    if (lane >= S && lane < 64 - S && gx < w) {
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(Lnew, 0, w * h * 4, 0x00020000);
#pragma unroll
        for (int r = S; r < S + RB; r++)
            if (!BORDER || ys + r < h) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t[r]), rn, 4 * gx, (ys + r) * w * 4, 0);
    }
    if (HALF) {
        // the last level of an octave: the next octave's start image, ((a + b) + (c + d)) * 0.25 as half_sample_kernel forms it, from the
        // rows in registers (bands start on even rows, strips on even columns: pairs never straddle waves). A template parameter, so that the
        // plain instantiations keep their register allocation.
        static_assert((RB & 1) == 0, "row pairs must stay inside a band");
        const int hw = w >> 1;
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(half, 0, hw * (h >> 1) * 4, 0x00020000);
        const bool col_ok = lane >= S && lane < 64 - S && !(gx & 1) && gx + 1 < w;
#pragma unroll
        for (int r = S; r < S + RB; r += 2) {
            const float top = t[r] + dpp_from_next_lane(t[r]);
            const float bot = t[r + 1] + dpp_from_next_lane(t[r + 1]);
            if (col_ok && (!BORDER || ys + r + 1 < h))
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (top + bot) * 0.25f), rh, 2 * gx, ((ys + r) >> 1) * hw * 4, 0);
        }
    }
}

template <int S, int RB, bool HALF = false>
#ifndef APDS_STRIP_WAVES
#define APDS_STRIP_WAVES 4
#endif
#ifndef APDS_STRIP_RB
#define APDS_STRIP_RB 16
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(APDS_STRIP_WAVES, 8))) void nld_strip_kernel(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ Lnew, int w, int h,
                                                         NldSteps steps, int strips, int nwaves, size_t bstride, float* __restrict__ half, ForkSignal sig) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_FORK_SIGNAL(sig);
    APDS_BOFS(Lt);
    APDS_BOFS(Lf);
    APDS_BOFS(Lnew);
    if (HALF) APDS_BOFS(half);
    constexpr int VW = 64 - 2 * S;              // columns a wave finishes
    // wave-uniform by construction; readfirstlane tells the compiler, so that row bases and row conditions live in scalar registers
    const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (id >= nwaves) return;                   // the kernel has no barriers
    const int band = __builtin_amdgcn_readfirstlane(id / strips);   // (the division itself runs on the vector ALU)
    const int strip = id - band * strips;
    const int gx0 = strip * VW - S, y0 = band * RB;
    const bool border = gx0 < 0 || gx0 + 64 > w || y0 - S < 0 || y0 + RB + S > h;
    if (border) nld_strip<S, RB, true, HALF>(Lt, Lf, Lnew, w, h, steps, gx0, y0, half);
    else nld_strip<S, RB, false, HALF>(Lt, Lf, Lnew, w, h, steps, gx0, y0, half);
}

// ---- resize(INTER_AREA) by exactly 2: mean of 2x2 ---------------------------------------------------------
// One thread = two neighbouring output pixels of HS_ROWS consecutive rows: 16-byte loads, all of a thread's loads issued before the first
// sum (one output per thread with 8-byte loads ran at 1.5 TB/s on the 4096^2 -> 2048^2 step, which sits on the level chain's critical path).
static constexpr int HS_ROWS = 4;
__global__ __launch_bounds__(256) void half_sample_kernel(const float* __restrict__ src, int sw, float* __restrict__ dst, int dw, int dh, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(src);
    APDS_BOFS(dst);
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int y0 = blockIdx.y * HS_ROWS;
    if (x >= dw) return;
    // 16-byte loads need rows that start on a 16-byte boundary; 8-byte stores an even destination width
    const bool wide = x + 1 < dw && !(sw & 3) && !(dw & 1) && !((reinterpret_cast<uintptr_t>(src) & 15) | (reinterpret_cast<uintptr_t>(dst) & 7));
    if (wide) {
        float4 a[HS_ROWS], b[HS_ROWS];
#pragma unroll
        for (int r = 0; r < HS_ROWS; r++) {
            const int y = min(y0 + r, dh - 1);
            a[r] = *reinterpret_cast<const float4*>(&src[(size_t)(2 * y) * sw + 2 * x]);
            b[r] = *reinterpret_cast<const float4*>(&src[(size_t)(2 * y + 1) * sw + 2 * x]);
        }
#pragma unroll
        for (int r = 0; r < HS_ROWS; r++) {
            const int y = y0 + r;
            if (y >= dh) break;
            float2 o;
            o.x = ((a[r].x + a[r].y) + (b[r].x + b[r].y)) * 0.25f;
            o.y = ((a[r].z + a[r].w) + (b[r].z + b[r].w)) * 0.25f;
            *reinterpret_cast<float2*>(&dst[(size_t)y * dw + x]) = o;
        }
        return;
    }
    for (int r = 0; r < HS_ROWS; r++) {
        const int y = y0 + r;
        if (y >= dh) break;
        for (int xx = x; xx < min(x + 2, dw); xx++) {
            const float* p0 = &src[(size_t)(2 * y) * sw + 2 * xx];
            const float* p1 = &src[(size_t)(2 * y + 1) * sw + 2 * xx];
            dst[(size_t)y * dw + xx] = ((p0[0] + p0[1]) + (p1[0] + p1[1])) * 0.25f;
        }
    }
}

// general INTER_AREA (odd source sizes): per destination pixel, up to 4 taps per axis from host-built tables
__global__ void area_resize_kernel(const float* __restrict__ src, int sw, float* __restrict__ dst, int dw, int dh, const int* __restrict__ xofs,
                                   const float* __restrict__ xw, const int* __restrict__ xcnt, const int* __restrict__ yofs,
                                   const float* __restrict__ yw, const int* __restrict__ ycnt, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(src);
    APDS_BOFS(dst);
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    float sum = 0.0f;
    const int ny = ycnt[y], nx = xcnt[x];
    for (int j = 0; j < ny; j++) {
        const float* s = &src[(size_t)yofs[y * 4 + j] * sw];
        float row = 0.0f;
        for (int k = 0; k < nx; k++) row += s[xofs[x * 4 + k]] * xw[x * 4 + k];
        if (j == 0) sum = yw[y * 4 + j] * row;
        else sum += yw[y * 4 + j] * row;
    }
    dst[(size_t)y * dw + x] = sum;
}

// ---- a1.5 + a1.6 fused: Lsmooth -> Lx, Ly (dilated Scharr pair at scale s) -> Lxx, Lxy, Lyy -> Ldet -> 3x3 extrema, one pass ----
// The reference applies sepFilter2D twice (derivatives of derivatives), each with BORDER_REFLECT_101 on ITS input. Fused,
// that means: the Lx/Ly values a tile needs within s pixels beyond the image are the values AT the reflected positions
// (Lx(reflect(p)), not a stencil evaluated on a reflected Lsmooth). So: Lsmooth tile with a 2s+1 halo (reflect on load),
// then Lx/Ly on the tile + (s+1) ring evaluated at the reflected coordinate of every ring position, then the second
// derivatives and the determinant on the tile + 1 ring (kept in LDS), then the strict 3x3 maxima of the tile above the
// threshold and inside the level's border -> keypoint mask + candidate list (block-aggregated append). Lsmooth is read once,
// Ldet is not read back for the extrema test, and two launches per level disappear.
static constexpr int DH = 32;             // tile height (the width is the kernel's TW parameter)
// (candidates of a tile: strict 3x3 maxima are never adjacent, so at most a quarter of the tile pixels — the room behind the determinant plane)

// S = sigma_size as a compile-time constant (2, 3, 4 are all the reference's AKAZE parameters produce): constant LDS strides turn
// the index divisions into multiplies and let the tile loads be issued together; S = 0 takes it at run time.
template <int S, int NT, int TW>
__device__ __forceinline__ void doh_tile_generic(const float* __restrict__ Lsmooth, float2* __restrict__ Lxy, float* __restrict__ Ldet, int w, int h, int s_rt,
                                                 float kside, float kmid, float sq, int border, float thr, uint8_t* __restrict__ mask,
                                                 uint32_t* __restrict__ list, int* __restrict__ list_count) {
    extern __shared__ float smem[];
    __shared__ int s_n, s_base;
    const int s = S ? S : s_rt;
    const int SW = TW + 4 * s + 2, SH = DH + 4 * s + 2;   // Lsmooth tile, halo 2s + 1
    const int MW = TW + 2 * s + 2, MH = DH + 2 * s + 2;   // first derivatives, halo s + 1
    constexpr int EW = TW + 2, EH = DH + 2;               // determinant, halo 1
    float* s_src = smem;
    float* s_mx = s_src + SW * SH;
    float* s_my = s_mx + MW * MH;
    float* s_det = s_src;                                 // reuses the Lsmooth tile once the first derivatives exist
    uint32_t* s_cand = reinterpret_cast<uint32_t*>(s_src + EW * EH);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * DH;
    const int ox = x0 - 2 * s - 1, oy = y0 - 2 * s - 1;   // global coordinate of s_src[0]
    if (threadIdx.x == 0) s_n = 0;
    // the tile with its 2s+1 halo inside the image (all but the outermost tiles, block-uniform): no reflected coordinates, no tests
    const bool inside = ox >= 0 && oy >= 0 && ox + SW <= w && oy + SH <= h;
    if constexpr (S > 0) {
        constexpr int CSW = TW + 4 * S + 2, CSH = DH + 4 * S + 2, NL = (CSW * CSH + NT - 1) / NT;
        float v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {   // all of the tile's loads in flight before the first LDS store
            const int i = min((int)threadIdx.x + k * NT, CSW * CSH - 1);
            const int ly = i / CSW, lx = i - ly * CSW;
            v[k] = inside ? Lsmooth[(size_t)(oy + ly) * w + (ox + lx)] : Lsmooth[(size_t)reflect101(oy + ly, h) * w + reflect101(ox + lx, w)];
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = threadIdx.x + k * NT;
            if (i < CSW * CSH) s_src[i] = v[k];
        }
    } else {
        for (int i = threadIdx.x; i < SW * SH; i += NT) {
            const int ly = i / SW, lx = i - ly * SW;
            s_src[i] = Lsmooth[(size_t)reflect101(oy + ly, h) * w + reflect101(ox + lx, w)];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MW * MH; i += NT) {
        const int my = i / MW, mx = i - my * MW;
        // the first-derivative value this ring position stands for lives at the reflected coordinate
        const int cx = inside ? mx + s : reflect101(x0 - s - 1 + mx, w) - ox, cy = inside ? my + s : reflect101(y0 - s - 1 + my, h) - oy;
        const float* r0 = &s_src[(cy - s) * SW + cx];
        const float* r1 = &s_src[cy * SW + cx];
        const float* r2 = &s_src[(cy + s) * SW + cx];
        const float rd0 = r0[s] - r0[-s], rd1 = r1[s] - r1[-s], rd2 = r2[s] - r2[-s];
        float ax = kmid * rd1;
        ax += kside * (rd0 + rd2);
        float rs0 = kmid * r0[0];
        rs0 += kside * (r0[-s] + r0[s]);
        float rs2 = kmid * r2[0];
        rs2 += kside * (r2[-s] + r2[s]);
        s_mx[i] = ax;
        s_my[i] = rs2 - rs0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < EW * EH; i += NT) {
        const int ey = i / EW, ex = i - ey * EW;
        const int gx = x0 - 1 + ex, gy = y0 - 1 + ey;
        if (!inside && (gx < 0 || gy < 0 || gx >= w || gy >= h)) continue;   // never compared: tested pixels are >= border away from the edge
        const int c = (ey + s) * MW + ex + s;
        const float* x0r = &s_mx[c - s * MW];
        const float* x1r = &s_mx[c];
        const float* x2r = &s_mx[c + s * MW];
        const float rd0 = x0r[s] - x0r[-s], rd1 = x1r[s] - x1r[-s], rd2 = x2r[s] - x2r[-s];
        float lxx = kmid * rd1;
        lxx += kside * (rd0 + rd2);
        float rsx0 = kmid * x0r[0];
        rsx0 += kside * (x0r[-s] + x0r[s]);
        float rsx2 = kmid * x2r[0];
        rsx2 += kside * (x2r[-s] + x2r[s]);
        const float lxy = rsx2 - rsx0;
        const float* y0r = &s_my[c - s * MW];
        const float* y2r = &s_my[c + s * MW];
        float rsy0 = kmid * y0r[0];
        rsy0 += kside * (y0r[-s] + y0r[s]);
        float rsy2 = kmid * y2r[0];
        rsy2 += kside * (y2r[-s] + y2r[s]);
        const float lyy = rsy2 - rsy0;
        const float det = (lxx * lyy - lxy * lxy) * sq;
        s_det[i] = det;
        if (ex >= 1 && ex <= TW && ey >= 1 && ey <= DH) {   // the tile itself
            const size_t o = (size_t)gy * w + gx;
            Lxy[o] = make_float2(s_mx[c], s_my[c]);   // interleaved: orientation and M-LDB gather both with one 8-byte load
            Ldet[o] = det;
        }
    }
    __syncthreads();
    if (border < 0) return;   // level too small for any extremum (block-uniform)
    const unsigned span_x = (unsigned)(w - 2 * border), span_y = (unsigned)(h - 2 * border);   // (positive: checked by the launcher)
    for (int i = threadIdx.x; i < TW * DH; i += NT) {
        const int ly = i / TW, lx = i - ly * TW;
        const int gx = x0 + lx, gy = y0 + ly;
        const float* p = &s_det[(ly + 1) * EW + lx + 1];
        const float v = p[0];
        // all nine tests evaluated (no short-circuit branches: a strict maximum is rare, the branches were most of this loop);
        // "reject if v <= neighbour" exactly as written in the reference, hence the negated comparisons
        // "v <= x for some neighbour x" is "v <= the largest neighbour": fmaxf ignores a NaN operand exactly as the comparison v <= NaN
        // is false, and !(v <= m) keeps a NaN v as the reference's chain of comparisons does (three v_max3_f32 instead of eight compares)
        const float m = fmaxf(fmaxf(fmaxf(fmaxf(p[-EW - 1], p[-EW]), p[-EW + 1]), fmaxf(p[-1], p[1])), fmaxf(fmaxf(p[EW - 1], p[EW]), p[EW + 1]));
        const bool keep = ((unsigned)(gx - border) < span_x) & ((unsigned)(gy - border) < span_y) & !(v <= thr) & !(v <= m);
        if (keep) {
            mask[(size_t)gy * w + gx] = 1;
            s_cand[atomicAdd(&s_n, 1)] = (uint32_t)gx | ((uint32_t)gy << 16);
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(list_count, n);   // list order is irrelevant (only used to enumerate candidates)
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) list[s_base + i] = s_cand[i];
}

template <int S, int NT, int TW>
__global__ __launch_bounds__(NT) void doh_fused_kernel(const float* __restrict__ Lsmooth, float2* __restrict__ Lxy, float* __restrict__ Ldet, int w, int h,
                                                       int s_rt, float kside, float kmid, float sq, int border, float thr, uint8_t* __restrict__ mask,
                                                       uint32_t* __restrict__ list, int* __restrict__ list_count, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_BOFS(Lsmooth);
    APDS_BOFS(Lxy);
    APDS_BOFS(Ldet);
    APDS_BOFS(mask);
    APDS_BOFS(list);
    APDS_BOFS(list_count);
    doh_tile_generic<S, NT, TW>(Lsmooth, Lxy, Ldet, w, h, s_rt, kside, kmid, sq, border, thr, mask, list, list_count);
}

// (Measured and not adopted, round 2: "column runs" — a thread owns a column of a stage and walks K consecutive rows, so that the row
// terms rd(r) = L[r][x+s] - L[r][x-s], rs(r) = kmid L[r][x] + kside (L[r][x-s] + L[r][x+s]) are computed once per row and shared by
// the outputs that use the row as their upper, middle or lower one. Bit-identical, ~12 % fewer instructions, and slower: K = 10 on
// 576 threads ran one block per CU (4096^2 extraction 2.26 ms against 1.93), K = 6 on 1024 threads at 64 registers 1.98 - 2.12 ms.
// Two horizontally adjacent outputs per item with packed-f32 arithmetic (v_pk_add_f32 / v_pk_mul_f32, pairs read with one LDS
// instruction) was tried again on the 64 x 32 tiles: bit-identical, 1.824 against 1.782 ms — v_pk_add / v_pk_mul issue at half the rate
// of their scalar forms (tools/valu_calib.py), so they save index arithmetic only, and the unaligned pair reads cost more than that.)

// ---- host launchers -------------------------------------------------------------------------------------
// grid of a persistent tile kernel with two 1024-thread blocks per CU: a multiple of 8 (one slice per XCD), at most 2 x 256 blocks
static int persistent_grid(int ntiles) {
    return std::min((ntiles + 7) & ~7, 512);
}
void launch_gray(const void* img, int rows, int cols, int channels, size_t stride, float* out, hipStream_t s, const Batch& b) {
    hipLaunchKernelGGL(gray_kernel, dim3(ceil_div(cols, 256), rows, b.n), dim3(256), 0, s, static_cast<const uint8_t*>(img), rows, cols, channels, stride, out,
                       b.img_stride, b.stride);
}

void launch_gauss(const float* src, float* dst, int w, int h, const GaussTaps& taps, int radius, hipStream_t s, const Batch& b) {
    dim3 grid(ceil_div(w, TW), ceil_div(h, TH), b.n);
    if (radius == 4) hipLaunchKernelGGL((gauss_kernel<4>), grid, dim3(256), 0, s, src, dst, w, h, taps, b.stride);
    else hipLaunchKernelGGL((gauss_kernel<2>), grid, dim3(256), 0, s, src, dst, w, h, taps, b.stride);
}

static size_t deriv_lds_bytes(int s) { return (size_t)((TH + 2 * s) * (TW + 2 * s) + 2 * (TH + 2 * s) * TW) * sizeof(float); }

void launch_smooth_flow(const float* src, float* smooth, float* flow, int w, int h, const GaussTaps& taps, const float* kptr, hipStream_t s, const Batch& b) {
    const int tiles_x = ceil_div(w, FW), tiles_y = ceil_div(h, FH), ntiles = tiles_x * tiles_y;
    // tiles [1, txi) x [1, tyi) lie inside the image with their 3-pixel halo: register strips; the frame around them: LDS tiles
    const int strip_mode = config().sf_strip;
    const int txi = w >= FW + 67 ? (w - 67) / FW + 1 : 1, tyi = h >= FH + 35 ? (h - 35) / FH + 1 : 1;
    // the strips pay once the launch has enough pixels to be throughput-bound: a batch counts as a whole
    const bool strips_on = strip_mode && txi > 1 && tyi > 1 && (size_t)w * h < ((size_t)1 << 29) && ((size_t)w * h * b.n >= ((size_t)1 << 21) || strip_mode == 2);
    if (strips_on) {
        const int rx0 = FW, ry0 = FH, rx1 = txi * FW, ry1 = tyi * FH;
        const int strips = ceil_div(rx1 - rx0, SF_VW), nwaves = strips * ceil_div(ry1 - ry0, SF_RB);
        hipLaunchKernelGGL(smooth_flow_strip_kernel, dim3(ceil_div(nwaves, 4), 1, b.n), dim3(256), 0, s, src, smooth, flow, w, h, taps, kptr, rx0, ry0, rx1, ry1,
                           strips, nwaves, b.stride);
    }
    if (strips_on) {   // the frame: one tile per block
        const int n_frame = tiles_x + (tyi - 1) * (1 + tiles_x - txi) + (tiles_y - tyi) * tiles_x;
        hipLaunchKernelGGL(smooth_flow_kernel, dim3(n_frame, 1, b.n), dim3(FNT), 0, s, src, smooth, flow, w, h, taps, kptr, tiles_x, ntiles, txi, tyi, b.stride);
    } else {
        hipLaunchKernelGGL(smooth_flow_kernel, dim3(persistent_grid(ntiles), 1, b.n), dim3(FNT), 0, s, src, smooth, flow, w, h, taps, kptr, tiles_x, ntiles, 0, 0,
                           b.stride);
    }
}
void launch_kcontrast(const float* smooth, float* modg_tmp, int w, int h, unsigned int* hmax_bits, int* hist, float* k_oct, int n_oct, hipStream_t s,
                      const Batch& b, bool gradient_done) {
    // hmax_bits and hist are zeroed by the caller (one clear of the per-image counter block for the whole batch)
    if (!gradient_done)    // otherwise launch_base_strips has written |grad| to modg_tmp and its maximum to hmax_bits
        hipLaunchKernelGGL((deriv_pair_kernel<2>), dim3(ceil_div(w, TW), ceil_div(h, TH), b.n), dim3(256), deriv_lds_bytes(1), s, smooth, modg_tmp,
                           (float*)nullptr, w, h, 1, 3.0f, 10.0f, (const float*)nullptr, hmax_bits, b.stride);
    // the histogram's grid shrinks with the image: 1024 blocks for one large frame, a share of that for each image of a batch
    const int hist_blocks = std::max(8, std::min(1024, ceil_div((long long)w * h, 4096)));
    // 16 sub-histograms per block (8: 1.755, 16: 1.739, 32: 1.763 ms per 4096^2 extraction, profiles/r03/half_sample_ab.txt)
    hipLaunchKernelGGL(kcontrast_hist_kernel<16>, dim3(hist_blocks, 1, b.n), dim3(256), 0, s, modg_tmp, w, h, hmax_bits, hist, b.stride);
    hipLaunchKernelGGL(kcontrast_finish_kernel, dim3(1, 1, b.n), dim3(64), 0, s, hist, hmax_bits, w, h, k_oct, n_oct, b.stride);
}
// image -> Lt[0] (and, if want_modg, |grad| of the sigma = 1 image + its interior maximum) in one pass on register strips. Returns
// false when the image is too small to pay (the launch-bound small tiles keep the separate kernels) or does not fit 32-bit offsets.
bool launch_base_strips(const void* img, int rows, int cols, int channels, size_t stride, const GaussTaps& g16, const GaussTaps& g10, float* Lt0, float* modg,
                        unsigned int* hmax_bits, bool want_modg, hipStream_t s, const Batch& b) {
    const int strip_mode = config().base_strip;
    const size_t px = (size_t)rows * cols;
    if (!strip_mode || (px * b.n < ((size_t)1 << 21) && strip_mode != 2) || px >= ((size_t)1 << 29) || (size_t)rows * stride >= ((size_t)1 << 31)) return false;
    if (channels == 4 && ((reinterpret_cast<uintptr_t>(img) | stride | b.img_stride) & 3)) return false;   // dword loads of the BGRA pixels
    const int strips = ceil_div(cols, BS_VW), nwaves = strips * ceil_div(rows, BS_RB);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(ceil_div(nwaves, 4), 1, b.n), dim3(256), 0, s, static_cast<const uint8_t*>(img), cols, rows, (int)stride, g16, g10, Lt0,
                           modg, hmax_bits, want_modg ? 1 : 0, strips, nwaves, b.img_stride, b.stride);
    };
    if (channels == 4) go(&base_strip_kernel<4>);
    else if (channels == 3) go(&base_strip_kernel<3>);
    else go(&base_strip_kernel<1>);
    return true;
}
template <int S>
static void nld_multi_launch(const float* Lt, const float* Lf, float* Lnew, int w, int h, const NldSteps& st, hipStream_t s, const Batch& b) {
    // 1024 threads per 64x32 tile: each step is ~3 short dependent LDS passes, so what matters is waves in flight per CU
    const dim3 grid(ceil_div(w, T2W), ceil_div(h, T2H), b.n);
    hipLaunchKernelGGL((nld_multi_kernel<S, 1024>), grid, dim3(1024), 0, s, Lt, Lf, Lnew, w, h, st, b.stride);
}
template <int S>
static void nld_strip_launch(const float* Lt, const float* Lf, float* Lnew, int w, int h, const NldSteps& st, hipStream_t s, const Batch& b, float* half) {
    constexpr int RB = APDS_STRIP_RB;
    const int strips = ceil_div(w, 64 - 2 * S), nwaves = strips * ceil_div(h, RB);
    const ForkSignal sig = ctx().take_fork_signal();
    if (half)
        hipLaunchKernelGGL((nld_strip_kernel<S, RB, true>), dim3(ceil_div(nwaves, 4), 1, b.n), dim3(256), 0, s, Lt, Lf, Lnew, w, h, st, strips, nwaves, b.stride, half,
                           sig);
    else
        hipLaunchKernelGGL((nld_strip_kernel<S, RB, false>), dim3(ceil_div(nwaves, 4), 1, b.n), dim3(256), 0, s, Lt, Lf, Lnew, w, h, st, strips, nwaves, b.stride,
                           (float*)nullptr, sig);
}
// half_out (optional): ask the launch to write the 2 x 2 area means of Lnew as well (the next octave's start image). Returns whether it did
// (the register-strip form can; the LDS-tile form of the small launches cannot: the caller then runs half_sample_kernel).
bool launch_nld_multi(const float* Lt, const float* Lf, float* Lnew, int w, int h, const float* step_sizes, int nsteps, hipStream_t s, const Batch& b,
                      float* half_out) {
    NldSteps st{};
    for (int i = 0; i < nsteps; i++) st.v[i] = step_sizes[i];
    // register strips for the throughput-bound launches (up to 4 steps on launches of at least 1 Mpx, a batch counted as a whole);
    // the LDS tiles keep the deeply fused launches of the small, latency-bound octaves (their unrolled strip code would not fit the
    // instruction cache)
    const int strip_mode = config().nld_strip;
    if (strip_mode && nsteps <= 4 && ((size_t)w * h * b.n >= ((size_t)1 << 20) || strip_mode == 2) && (size_t)w * h < ((size_t)1 << 29)) {   // 32-bit byte offsets
        switch (nsteps) {
            case 1: nld_strip_launch<1>(Lt, Lf, Lnew, w, h, st, s, b, half_out); return half_out != nullptr;
            case 2: nld_strip_launch<2>(Lt, Lf, Lnew, w, h, st, s, b, half_out); return half_out != nullptr;
            case 3: nld_strip_launch<3>(Lt, Lf, Lnew, w, h, st, s, b, half_out); return half_out != nullptr;
            default: nld_strip_launch<4>(Lt, Lf, Lnew, w, h, st, s, b, half_out); return half_out != nullptr;
        }
    }
    switch (nsteps) {
        case 1: nld_multi_launch<1>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 2: nld_multi_launch<2>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 3: nld_multi_launch<3>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 4: nld_multi_launch<4>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 5: nld_multi_launch<5>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 6: nld_multi_launch<6>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 7: nld_multi_launch<7>(Lt, Lf, Lnew, w, h, st, s, b); break;
        case 8: nld_multi_launch<8>(Lt, Lf, Lnew, w, h, st, s, b); break;
        default: fail(APDS_ERR_INTERNAL, "nld_multi: 1..8 steps per launch");
    }
    return false;
}
// one launch for a level: Lsmooth, conductivity (kept in LDS; written to flow_out only if the caller continues with more steps)
// and `nsteps` <= level_fused_max_steps() FED steps from `src` into `Lnew` (src, smooth, Lnew distinct planes)
int level_fused_max_steps() { return LF_MAX_STEPS; }
void launch_level_fused(const float* src, float* smooth, float* flow_out, const float* flow_in, float* Lnew, int w, int h, const GaussTaps& taps,
                        const float* kptr, const float* step_sizes, int nsteps, hipStream_t s, const Batch& b, float* half_out) {
    APDS_REQUIRE(nsteps >= 1 && nsteps <= LF_MAX_STEPS, APDS_ERR_INTERNAL, "level_fused: 1..29 steps");
    LevelSteps st{};
    st.n = nsteps;
    for (int i = 0; i < nsteps; i++) st.v[i] = step_sizes[i];
    const int P = LFT + 2 * nsteps + 6;
    const dim3 grid(ceil_div(w, LFT), ceil_div(h, LFT), b.n);
    // several tiles per CU: 512-thread blocks, so that three or four of them share a CU; otherwise all the threads one tile can use
    const bool small_blocks = nsteps <= LF_MAX_STEPS_512 && (size_t)grid.x * grid.y * grid.z >= 512;
    const int shrink = config().fed_shrink ? 1 : 0;
    const size_t lds = (size_t)(3 * P * P) * sizeof(float);
    static bool opted = false;   // above the default dynamic-LDS limit: opt in once (idempotent, so a race is harmless)
    if (!opted) {
        constexpr int PM = LFT + 2 * LF_MAX_STEPS + 6, PM5 = LFT + 2 * LF_MAX_STEPS_512 + 6;
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&level_fused_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PM * PM * 4));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&level_fused_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PM5 * PM5 * 4));
        opted = true;
    }
    const ForkSignal sig = ctx().take_fork_signal();
    if (small_blocks)
        hipLaunchKernelGGL((level_fused_kernel<512>), grid, dim3(512), lds, s, src, smooth, flow_out, flow_in, Lnew, w, h, taps, kptr, st, b.stride, shrink, half_out,
                           sig);
    else
        hipLaunchKernelGGL((level_fused_kernel<1024>), grid, dim3(1024), lds, s, src, smooth, flow_out, flow_in, Lnew, w, h, taps, kptr, st, b.stride, shrink,
                           half_out, sig);
}
void launch_half_sample(const float* src, int sw, float* dst, int dw, int dh, hipStream_t s, const Batch& b) {
    hipLaunchKernelGGL(half_sample_kernel, dim3(ceil_div(dw, 512), ceil_div(dh, HS_ROWS), b.n), dim3(256), 0, s, src, sw, dst, dw, dh, b.stride);
}
void launch_area_resize(const float* src, int sw, float* dst, int dw, int dh, const int* xofs, const float* xw, const int* xcnt, const int* yofs,
                        const float* yw, const int* ycnt, hipStream_t s, const Batch& b) {
    hipLaunchKernelGGL(area_resize_kernel, dim3(ceil_div(dw, 256), dh, b.n), dim3(256), 0, s, src, sw, dst, dw, dh, xofs, xw, xcnt, yofs, yw, ycnt, b.stride);
}
void launch_doh_fused(const float* Lsmooth, float2* Lxy, float* Ldet, int w, int h, int sc, float kside, float kmid, int border, float thr, uint8_t* mask,
                      uint32_t* list, int* list_count, hipStream_t s, const Batch& b) {
    // 64 x 32 tiles on 512 threads (halo 1.75x, four blocks per CU: the kernel is four barrier-separated phases, and more independent
    // blocks per CU hide their latencies better than the 128 x 32 / 1024-thread tiles of round 1 or 32-pixel tiles on 256 threads did:
    // 1.82 against 1.91 and 1.98 ms per 4096^2 extraction; those variants are gone)
    constexpr int tw = 64;
    const size_t lds = (size_t)((tw + 4 * sc + 2) * (DH + 4 * sc + 2) + 2 * (tw + 2 * sc + 2) * (DH + 2 * sc + 2)) * sizeof(float);
    APDS_REQUIRE((size_t)(tw + 2) * (DH + 2) + (size_t)tw * DH / 4 <= (size_t)(tw + 4 * sc + 2) * (DH + 4 * sc + 2), APDS_ERR_INTERNAL, "doh_fused: LDS aliasing needs sigma_size >= 2");
    // the extrema test of a level that is too small for its border is skipped (border < 0 in the kernel)
    const bool none = border + 1 >= h || w - 2 * border <= 0 || h - 2 * border <= 0;
    auto go = [&](auto kernel, int nt) {
        if (lds > 64 * 1024)   // above the default dynamic-LDS limit: opt in (idempotent)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipLaunchKernelGGL(kernel, dim3(ceil_div(w, tw), ceil_div(h, DH), b.n), dim3(nt), lds, s, Lsmooth, Lxy, Ldet, w, h, sc, kside, kmid,
                           (float)(sc * sc * sc * sc), none ? -1 : border, thr, mask, list, list_count, b.stride);
    };
    switch (sc) {
        case 2: go(&doh_fused_kernel<2, 512, 64>, 512); break;
        case 3: go(&doh_fused_kernel<3, 512, 64>, 512); break;
        case 4: go(&doh_fused_kernel<4, 512, 64>, 512); break;
        default: go(&doh_fused_kernel<0, 512, 64>, 512); break;
    }
}

}  // namespace apds
