// csrc/akaze_level_stream.hip — one AKAZE evolution level step as a STREAMING kernel (the largest levels).
//
// level_strip_kernel (akaze_level_strips.hip) holds a strip of 16 output rows with its whole halo in registers: 16 + 2 (S + 3) rows are
// loaded and filtered to finish 16, i.e. 1.75x (S = 3) or 1.88x (S = 4) the rows, every one of them through all stages. Here a wave owns
// the same 64 columns (one per lane, halo S + 3 on either side) but WALKS DOWN a band of rows, and every stage keeps only the last few
// rows it needs in small register rings (indexed statically: the row loop is unrolled by the ring period 8):
//
//   row u arrives     G(u)      = 5-tap Gaussian of the source row along x (DPP neighbours)                            ring 8
//                     Ls(u-2)   = the same taps down the rows G(u-4 .. u)  -> Lsmooth (stored)
//                     rd, rs    = Scharr row terms of Ls(u-2) (DPP neighbours, reflect-101 in x)                       ring 4
//                     f(u-3)    = PM-g2 conductivity from rd / rs of rows u-4, u-3, u-2 (reflect-101 in y)            ring 8
//                     fx(u-3)   = f + f(x+1);  fy(u-4) = f(u-4) + f(u-3): the pair sums every FED step uses            ring 8
//   FED step j        t_j(u-3-j) from t_(j-1) of rows u-4-j .. u-2-j (its x flux through DPP, its y fluxes from the ring), j = 1 .. S
//                     t_S(u-3-S) -> the level's new Lt (stored)
//
// so a band of R rows costs R + 2 (S + 3) row steps instead of R x 1.75. Same operations in the same order as level_strip /
// smooth_flow_kernel / nld_point (the float contract of akaze_filters.hip): bit-identical planes. Borders as there: the Gaussian
// replicates (clamped loads), the Scharr pass reflects (lane selects at x = 0 / w - 1, the other row at y = 0 / h - 1), a flux across
// the image edge is +0 and the four corner pixels keep their value. Reference: OpenCV's Create_Nonlinear_Scale_Space behind
// feature_extraction/src/lib.rs:64-79 (restated in oracle/akaze_oracle.cpp).
#include "akaze.h"
#include "config.h"

namespace apds {

namespace {

__device__ __forceinline__ int clampi_(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
__device__ __forceinline__ float lane_next(float v) {   // lane i <- lane i + 1 (wave_shl:1; lane 63 keeps garbage: a halo lane)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_prev(float v) {   // lane i <- lane i - 1 (wave_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}

struct StreamSteps {
    float v[4];
};

struct LevelStreamArgs {
    const float* src;
    float* smooth;
    float* flow_out;
    float* Lnew;
    float* half;     // the next octave's start image (2 x 2 means of Lnew, (w / 2) x (h / 2)) or null: see store_half below
    const float* kptr;
    int w, h, strips, bands, rb;
    GaussTaps taps;
    StreamSteps steps;
};

template <int S, bool XEDGE, bool YEDGE, bool FLOW_OUT, bool HALF>
__device__ __forceinline__ void level_stream_rows(const LevelStreamArgs& a, int strip, int band) {
    constexpr int H = S + 3;              // halo: S (FED) + 1 (Scharr ring) + 2 (Gaussian)
    constexpr int VW = 64 - 2 * H;
    const int lane = threadIdx.x & 63;
    const int w = a.w, h = a.h;
    const int gx = strip * VW - H + lane;
    const int y0 = band * a.rb, y1 = min(y0 + a.rb, h);
    const bool mine = lane >= H && lane < 64 - H && gx < w;   // (gx >= 0 for these lanes)
    const float k0 = a.taps.k[0], k1 = a.taps.k[1], k2 = a.taps.k[2];
    const float kc = *a.kptr;
    const float k2inv = 1.0f / (kc * kc);
    const int plane_bytes = w * h * 4;
    const __amdgpu_buffer_rsrc_t r_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src), 0, plane_bytes, 0x00020000);
    const int cx4 = 4 * (XEDGE ? clampi_(gx, w) : gx);            // replicate in x: what the Gaussian wants
    constexpr int DROP = (int)0x80000000;                          // stores of lanes without an output column fall outside every plane
    const int xo4 = mine ? 4 * gx : DROP;
    const bool at_x0 = gx == 0, at_x1 = gx == w - 1;
    const bool flux_x_inside = gx >= 0 && gx + 1 <= w - 1;        // the edge between columns gx and gx + 1
    const bool edge_col = at_x0 || at_x1;
    const int u0 = y0 - H;                                          // first source row of the walk
    const int T = (y1 - y0) + 2 * H;                                // row steps: the last one finishes output row y1 - 1
    auto load_row = [&](int u) -> float {
        const int r = YEDGE ? clampi_(u, h) : min(u, h - 1);       // replicate in y (interior bands never leave the image upwards)
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_src, cx4, __builtin_amdgcn_readfirstlane(r * w * 4), 0));
    };
    // a store of a row outside [y0, y1) goes through a zero-length descriptor at offset 0: dropped (see akaze_doh_strips.hip)
    auto store_row = [&](float* plane, float v, int row) {
        const bool ok = row >= y0 && row < y1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(plane, 0, ok ? plane_bytes : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, xo4, __builtin_amdgcn_readfirstlane(ok ? row * w * 4 : 0), 0);
    };
    // Round 4: when this level is the last of its octave the kernel also writes the next octave's start image - the 2 x 2 area mean
    // ((a + b) + (c + d)) * 0.25 of half_sample_kernel, same operands in the same order - from the rows it has in hand: the separate pass
    // over the plane (40 us on the level chain's critical path at 4096^2) is gone. A lane with an even column pairs with its right
    // neighbour (DPP), an odd row with the row before it (kept in `hprev`); columns pair inside a strip (VW and the strip origins are
    // even), rows inside a band (the launcher makes the band height even). HALF is a template parameter: the plain instantiations keep
    // the register allocation they were tuned with (as a run-time flag it changed the schedule of ALL of them: 70 instead of 103 VGPRs, 90
    // / 97 us instead of 71 / 74 for the 16-Mpx levels).
    const int hw = w >> 1;
    const int half_bytes = hw * (h >> 1) * 4;
    const int hx4 = (mine && !(gx & 1) && gx + 1 < w) ? 2 * gx : DROP;   // 4 * (gx / 2)
    float hprev = 0.0f;
    auto store_half = [&](float out, int row) {
        // rows row - 1 (hprev) and row; stored when row is odd and both lie in [y0, y1) (y0 is even, so row - 1 >= y0 whenever row > y0)
        const float top = hprev + lane_next(hprev);
        const float bot = out + lane_next(out);
        const bool ok = (row & 1) && row > y0 && row < y1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.half, 0, ok ? half_bytes : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (top + bot) * 0.25f), rs, hx4, __builtin_amdgcn_readfirstlane(ok ? (row >> 1) * hw * 4 : 0), 0);
        hprev = out;
    };
    float P[8], G[8], F[8], FX[8], FY[8], rd[4], rs[4], TT[S][2], qprev[S];
#pragma unroll
    for (int i = 0; i < 8; i++) P[i] = G[i] = F[i] = FX[i] = FY[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) rd[i] = rs[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < S; i++) TT[i][0] = TT[i][1] = qprev[i] = 0.0f;   // (t_S needs no ring: it goes straight to memory)
    float cur[8], nxt[8];
#pragma unroll
    for (int j = 0; j < 8; j++) cur[j] = load_row(u0 + j);
    for (int tb = 0; tb < T; tb += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) nxt[j] = load_row(u0 + tb + 8 + j);   // the next period's rows, in flight while this one computes
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int u = u0 + tb + j;      // source row handled now; every ring slot of row u - lag is (j - lag) & mask
            // ---- stage A: Gaussian along x of the source row (also t_0 of the FED steps)
            const float v = cur[j];
            P[j & 7] = v;
            {
                const float l1 = lane_prev(v), r1 = lane_next(v);
                const float l2 = lane_prev(l1), r2 = lane_next(r1);
                float acc = k0 * v;
                acc += k1 * (l1 + r1);
                acc += k2 * (l2 + r2);
                G[j & 7] = acc;
            }
            // ---- stage B: Gaussian down the rows -> Lsmooth(u - 2)
            float ls = k0 * G[(j - 2) & 7];
            ls += k1 * (G[(j - 3) & 7] + G[(j - 1) & 7]);
            ls += k2 * (G[(j - 4) & 7] + G[j & 7]);
            store_row(a.smooth, ls, u - 2);
            // ---- stage C: Scharr row terms of Lsmooth(u - 2)
            {
                float l = lane_prev(ls), rr = lane_next(ls);
                if (XEDGE) {                  // reflect-101 in x
                    const float l0 = l;
                    l = at_x0 ? rr : l;
                    rr = at_x1 ? l0 : rr;
                }
                rd[(j - 2) & 3] = rr - l;
                float t = 10.0f * ls;
                t += 3.0f * (l + rr);
                rs[(j - 2) & 3] = t;
            }
            // ---- stage D: conductivity of row u - 3
            {
                const int vr = u - 3;
                float rdu = rd[(j - 4) & 3], rdd = rd[(j - 2) & 3], rsu = rs[(j - 4) & 3], rsd = rs[(j - 2) & 3];
                if (YEDGE) {                  // reflect-101 in y (wave-uniform)
                    if (vr == 0) {
                        rdu = rdd;
                        rsu = rsd;
                    }
                    if (vr == h - 1) {
                        rdd = rd[(j - 4) & 3];
                        rsd = rs[(j - 4) & 3];
                    }
                }
                float ax = 10.0f * rd[(j - 3) & 3];
                ax += 3.0f * (rdu + rdd);
                const float ay = rsd - rsu;
                const float f = 1.0f / (1.0f + ((ax * ax + ay * ay) * k2inv));
                F[(j - 3) & 7] = f;
                if (FLOW_OUT) store_row(a.flow_out, f, vr);
                FX[(j - 3) & 7] = f + lane_next(f);                 // the x pair sum of the FED steps: (f[x] + f[x+1])
                FY[(j - 4) & 7] = F[(j - 4) & 7] + f;               // the y pair sum of row u - 4: (f[r] + f[r+1])
            }
            // ---- FED steps: step s finishes row u - 3 - s from t_(s-1) of that row and the next one
#pragma unroll
            for (int s = 1; s <= S; s++) {
                const int lag = 3 + s, rho = u - lag;
                const float tc = s == 1 ? P[(j - lag) & 7] : TT[s - 2][(j - lag) & 1];
                const float tn = s == 1 ? P[(j - lag + 1) & 7] : TT[s - 2][(j - lag + 1) & 1];   // row rho + 1 (step s - 1 made it in this row step)
                const float d = lane_next(tc) - tc;
                float px = FX[(j - lag) & 7] * d;
                if (XEDGE && !flux_x_inside) px = 0.0f;
                float q = FY[(j - lag) & 7] * (tn - tc);
                if (YEDGE && !(rho >= 0 && rho + 1 <= h - 1)) q = 0.0f;
                float sum = px - lane_prev(px);
                sum = sum + q;
                sum = sum - qprev[s - 1];
                float out = tc + sum * a.steps.v[s - 1];
                if (XEDGE && YEDGE && edge_col && (rho == 0 || rho == h - 1)) out = tc;
                qprev[s - 1] = q;
                if (s < S) TT[s - 1][(j - lag) & 1] = out;
                else {
                    store_row(a.Lnew, out, rho);
                    if (HALF) store_half(out, rho);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = nxt[j];
    }
}

}  // namespace

template <int S, bool FLOW_OUT, bool HALF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void level_stream_kernel(LevelStreamArgs a, size_t bstride, ForkSignal sig) {
    APDS_RAISE_WAVE_PRIORITY();
    APDS_FORK_SIGNAL(sig);
    a.src = bofs(a.src, bstride);
    a.smooth = bofs(a.smooth, bstride);
    a.Lnew = bofs(a.Lnew, bstride);
    a.kptr = bofs(a.kptr, bstride);
    if (FLOW_OUT) a.flow_out = bofs(a.flow_out, bstride);
    if (HALF) a.half = bofs(a.half, bstride);
    constexpr int H = S + 3, VW = 64 - 2 * H;
    const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (id >= a.strips * a.bands) return;   // no barriers in this kernel
    const int band = __builtin_amdgcn_readfirstlane(id / a.strips);
    const int strip = id - band * a.strips;
    const int gx0 = strip * VW - H, y0 = band * a.rb;
    const bool xedge = gx0 < 0 || gx0 + 64 > a.w;
    const bool yedge = y0 - H < 0 || y0 + a.rb + H + 16 > a.h;   // (+ 16: the unrolled loop's padding and the prefetch stay inside the image)
    if (xedge && yedge) level_stream_rows<S, true, true, FLOW_OUT, HALF>(a, strip, band);
    else if (xedge) level_stream_rows<S, true, false, FLOW_OUT, HALF>(a, strip, band);
    else if (yedge) level_stream_rows<S, false, true, FLOW_OUT, HALF>(a, strip, band);
    else level_stream_rows<S, false, false, FLOW_OUT, HALF>(a, strip, band);
}

// Lsmooth, conductivity and the level's first `nsteps` (1 .. 4) FED steps, streaming form. False: not a level for it (the caller takes
// level_strip_kernel).
bool launch_level_stream(const float* src, float* smooth, float* flow_out, float* Lnew, int w, int h, const GaussTaps& taps, const float* kptr,
                         const float* step_sizes, int nsteps, hipStream_t s, const Batch& b, float* half_out) {
    const int mode = config().level_stream;
    if (mode == 0 || nsteps < 1 || nsteps > 4 || (size_t)w * h >= ((size_t)1 << 29) || w < 64 || h < 32) return false;
    if (mode != 2 && (size_t)w * h * b.n < ((size_t)1 << 23)) return false;
    const int H = nsteps + 3, vw = 64 - 2 * H;
    const int strips = ceil_div(w, vw);
    LevelStreamArgs a{src, smooth, flow_out, Lnew, half_out, kptr, w, h, strips, 0, 0, taps, {}};
    for (int i = 0; i < nsteps; i++) a.steps.v[i] = step_sizes[i];
    auto rows_for = [&](auto kernel) { return config().level_stream_rows > 0 ? config().level_stream_rows : stream_band_rows(kernel, strips, h, b.n, 64, 16); };
    auto go = [&](auto kernel) {
        a.rb = rows_for(kernel);
        if (half_out) a.rb = (a.rb + 1) & ~1;   // row pairs of the fused half-sample stay inside a band
        a.bands = ceil_div(h, a.rb);
        hipLaunchKernelGGL(kernel, dim3(ceil_div((long long)a.strips * a.bands, 4), 1, b.n), dim3(256), 0, s, a, b.stride, ctx().take_fork_signal());
    };
    APDS_REQUIRE(!(half_out && flow_out), APDS_ERR_INTERNAL, "level_stream: a launch that finishes its level writes no conductivity plane");
    if (half_out) {
        switch (nsteps) {
            case 1: go(&level_stream_kernel<1, false, true>); break;
            case 2: go(&level_stream_kernel<2, false, true>); break;
            case 3: go(&level_stream_kernel<3, false, true>); break;
            default: go(&level_stream_kernel<4, false, true>); break;
        }
    } else if (flow_out) {
        a.half = nullptr;
        switch (nsteps) {
            case 1: go(&level_stream_kernel<1, true>); break;
            case 2: go(&level_stream_kernel<2, true>); break;
            case 3: go(&level_stream_kernel<3, true>); break;
            default: go(&level_stream_kernel<4, true>); break;
        }
    } else {
        switch (nsteps) {
            case 1: go(&level_stream_kernel<1, false>); break;
            case 2: go(&level_stream_kernel<2, false>); break;
            case 3: go(&level_stream_kernel<3, false>); break;
            default: go(&level_stream_kernel<4, false>); break;
        }
    }
    return true;
}

}  // namespace apds
