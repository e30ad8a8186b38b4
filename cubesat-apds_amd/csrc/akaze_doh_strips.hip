// csrc/akaze_doh_strips.hip — a1.5 + a1.6 (determinant of the Hessian + 3x3 extrema) as a STREAMING kernel for the large levels.
//
// Replaces, for levels of at least 64 x 64 pixels, the LDS-tile kernel doh_fused_kernel (akaze_filters.hip), whose four phases walk
// 2-D tiles with per-item index arithmetic: ~160 VALU instructions and ~38 LDS reads per pixel. Same values, bit for bit
// (OpenCV's Compute_Determinant_Hessian_Response / Find_Scale_Space_Extrema behind feature_extraction/src/lib.rs:64-79; restated in
// oracle/akaze_oracle.cpp): the filter is separable, so a wave that owns 64 columns (one per lane) and walks DOWN the rows needs its
// horizontal neighbours only - and everything vertical stays in the lane's own registers:
//
//   row terms of the smoothed image L (taps at x +- s):   rd(r) = L[r][x+s] - L[r][x-s]      rs(r) = kmid L[r][x] + kside (L[r][x-s] + L[r][x+s])
//   first derivatives at row v = r - s:                    Lx(v) = kmid rd(v) + kside (rd(v-s) + rd(v+s))     Ly(v) = rs(v+s) - rs(v-s)
//   row terms of (Lx, Ly):                                 rdx, rsx from Lx; rsy from Ly (the same two formulas)
//   second derivatives at row z = v - s:                   Lxx(z) = kmid rdx(z) + kside (rdx(z-s) + rdx(z+s))  Lxy = rsx(z+s) - rsx(z-s)  Lyy = rsy(z+s) - rsy(z-s)
//   det(z) = (Lxx Lyy - Lxy Lxy) s^4, and the strict 3x3 maximum test of row z - 1 from the last three det rows.
//
// Per row a wave does ONE global load per lane, two exchanges through a wave-private LDS line (L, then the (Lx, Ly) pair: a write and two
// reads each, at lanes x -+ s - no barrier, a wave's LDS operations execute in order), ~40 VALU instructions, and the stores of
// (Lx, Ly), det, the keypoint mask byte and the suppression-status byte of the rows that have just become final. Row terms live in rings
// of 2s + 1 registers indexed statically (the row loop is unrolled by the ring length). x +- 1 of the extrema test come through DPP.
//
// Borders. OpenCV applies sepFilter2D twice, each time reflecting (101) ITS OWN input: a first derivative needed outside the image is the
// VALUE at the reflected position. Columns: the lanes' LDS read indices are reflected once, at kernel start (idx = reflect(x -+ s) - X0),
// which gives exactly that in both passes. Rows: a band at the top or bottom edge walks the VIRTUAL rows -1-2s .. and loads row
// reflect(u); rd / rs of a virtual row are those of its mirror image, Lx of a virtual row comes out right by itself (its two outer
// taps swap places: a + b == b + a), and Ly of a virtual row needs its operands swapped (rs(v-s) - rs(v+s): the mirror's upper tap is this
// row's lower one). Interior bands carry none of this (VEDGE = false).
#include <map>
#include <mutex>

#include "akaze.h"
#include "config.h"

namespace apds {

namespace {

__device__ __forceinline__ int reflect101i(int i, int n) {
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
// (lanes 63 / 0 have no source lane and keep whatever the destination register held: they are halo lanes)
__device__ __forceinline__ float lane_next(float v) {   // lane i <- lane i + 1 (wave_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_prev(float v) {   // lane i <- lane i - 1 (wave_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}

// maxNum of finite-or-NaN floats as ONE instruction each: fmaxf() makes the compiler quiet every operand first (a v_max_f32 x, x per
// input), which doubled the extrema test. v_max_f32 / v_max3_f32 return the other operand(s) when one is a NaN, as fmaxf does.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr int DS_CAND = 192;       // candidate slots per wave before a flush (a row adds at most 32: strict maxima are never adjacent)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct DohStripArgs {
    const float* Lsmooth;
    float2* Lxy;
    float* Ldet;
    uint8_t* mask;
    uint8_t* status;
    int w, h, border;
    float kside, kmid, sq, thr;
    int strips, bands, rb;
    int dense_det;   // 1: the determinant of every pixel is stored (debug-plane reads); 0: only around the candidates (see the store below)
};

// a wave's collected candidates -> the level's list (one atomic per flush; the list's order does not matter)
// Buffer operations, not plain pointers: through bofs()'s integer arithmetic the pointers are generic, and a flat_* operation anywhere in
// the row loop (even on this rare path) makes every wait of the loop a vmcnt(0).
__device__ __forceinline__ void flush_candidates(uint32_t* __restrict__ list, int* __restrict__ list_count, const uint32_t* __restrict__ cand, int n, int lane) {
    const __amdgpu_buffer_rsrc_t r_cnt = __builtin_amdgcn_make_buffer_rsrc(list_count, 0, 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_list = __builtin_amdgcn_make_buffer_rsrc(list, 0, 0x7ffffffc, 0x00020000);
    int base = 0;
    if (lane == 0) base = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(n, r_cnt, 0, 0, 0);
    base = __builtin_amdgcn_readfirstlane(base);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < DS_CAND / 64; k++)
        if (lane + 64 * k < n) __builtin_amdgcn_raw_buffer_store_b32((int)cand[lane + 64 * k], r_list, 4 * (base + lane + 64 * k), 0, 0);
    __builtin_amdgcn_wave_barrier();
}

template <int S, bool VEDGE>
__device__ __forceinline__ void doh_strip_rows(const DohStripArgs& a, uint32_t* __restrict__ list, int* __restrict__ list_count, int strip, int band,
                                               float* __restrict__ xl, float2* __restrict__ xxy, uint32_t* __restrict__ cand) {
    constexpr int P = 2 * S + 1;            // ring length = rows per unrolled period
    constexpr int VW = 64 - 4 * S - 2;      // columns a wave finishes
    const int lane = threadIdx.x & 63;
    const int w = a.w, h = a.h;
    const int X0 = strip * VW - P;          // image column of lane 0
    const int x = X0 + lane;
    const int y0 = band * a.rb, y1 = min(y0 + a.rb, h);
    // lanes P .. 63 - P hold this wave's output columns; x >= 0 there by construction
    const bool out_lane = lane >= P && lane < 64 - P && x < w;
    // LDS read indices of the taps at x -+ s, reflected at the image edge once for every row; lanes that are not image columns (or whose
    // tap lies outside the wave) read something harmless: nothing derived from them is ever stored
    const int im = min(max(reflect101i(x - S, w) - X0, 0), 63), ip = min(max(reflect101i(x + S, w) - X0, 0), 63);
    const float kside = a.kside, kmid = a.kmid, sq = a.sq, thr = a.thr;
    const int border = a.border;
    const bool x_tested = border >= 0 && out_lane && (unsigned)(x - border) < (unsigned)(w - 2 * border);
    const int plane4 = w * h * 4;
    const __amdgpu_buffer_rsrc_t r_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Lsmooth), 0, plane4, 0x00020000);
    const int x4 = 4 * x;                   // (negative for the first strip's left halo lanes: out of range, the load returns 0)
    // Stores are UNCONDITIONAL instructions (the compiler can then count them and wait for the prefetched rows only, not for every store in
    // flight): a lane that owns no output column stores at an offset beyond every plane, which the buffer range check drops, and a row that
    // is not (yet / any more) a row of this band stores through a zero-length descriptor at row offset 0.
    constexpr int DROP = (int)0x80000000;
    const int xo4 = out_lane ? x4 : DROP, xo8 = out_lane ? 2 * x4 : DROP, xo1 = out_lane ? x : DROP;
    const int u0 = y0 - 1 - 2 * S;          // first (virtual) row of the smoothed image this band needs
    const int T = (y1 - y0) + 4 * S + 2;    // rows to walk: the last one finishes the extrema test of row y1 - 1
    auto row_offset = [&](int u) -> int {
        const int r = VEDGE ? reflect101i(u, h) : u;
        return __builtin_amdgcn_readfirstlane(r * w * 4);
    };
    float rd[P], rs[P], rdx[P], rsx[P], rsy[P];
#pragma unroll
    for (int j = 0; j < P; j++) rd[j] = rs[j] = rdx[j] = rsx[j] = rsy[j] = 0.0f;
    float d1 = 0.0f, d0 = 0.0f, hm0 = 0.0f, hs1 = 0.0f, hm1 = 0.0f;   // det rows z - 1 and z - 2; neighbour maxima of rows z - 2 and z - 1
    int ncand = 0;                                          // wave-uniform
    float cur[P], nxt[P];
#pragma unroll
    for (int j = 0; j < P; j++) cur[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_src, x4, row_offset(u0 + j), 0));
    for (int tb = 0; tb < T; tb += P) {
#pragma unroll
        for (int j = 0; j < P; j++)   // the next period's rows: in flight while this period computes (rows past the band are loaded and never used)
            nxt[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_src, x4, row_offset(VEDGE ? u0 + tb + P + j : min(u0 + tb + P + j, h - 1)), 0));
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int jA = (j + 1) % P;          // ring slot of the row 2s ago
            const int jB = (j + S + 1) % P;      // ring slot of the row s ago
            const int u = u0 + tb + j;           // (virtual) row of L handled now
            const int v = u - S;                 // row of (Lx, Ly) finished now
            const int z = v - S;                 // row of det finished now
            const int y = z - 1;                 // row whose extrema test is finished now
            // ---- stage 1: row terms of L
            const float Lc = cur[j];
            xl[lane] = Lc;
            __builtin_amdgcn_wave_barrier();
            const float Lm = xl[im], Lp = xl[ip];
            __builtin_amdgcn_wave_barrier();
            rd[j] = Lp - Lm;
            {
                float t = kmid * Lc;
                t += kside * (Lm + Lp);
                rs[j] = t;
            }
            float lx = kmid * rd[jB];
            lx += kside * (rd[jA] + rd[j]);
            float ly;
            if (VEDGE && (unsigned)v >= (unsigned)h) ly = rs[jA] - rs[j];   // a virtual row stands for its mirror image: taps swapped
            else ly = rs[j] - rs[jA];
            {
                const bool ok = v >= y0 && v < y1;   // (record count AND row offset zero for a dropped row: out of range under either form of the check)
                const __amdgpu_buffer_rsrc_t r_xy = __builtin_amdgcn_make_buffer_rsrc(a.Lxy, 0, ok ? 2 * plane4 : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, lx), __builtin_bit_cast(unsigned, ly)}, r_xy, xo8,
                                                      __builtin_amdgcn_readfirstlane(ok ? v * w * 8 : 0), 0);
            }
            // ---- stage 2: row terms of (Lx, Ly)
            xxy[lane] = make_float2(lx, ly);
            __builtin_amdgcn_wave_barrier();
            const float2 dm = xxy[im], dp = xxy[ip];
            __builtin_amdgcn_wave_barrier();
            rdx[j] = dp.x - dm.x;
            {
                float t = kmid * lx;
                t += kside * (dm.x + dp.x);
                rsx[j] = t;
                float q = kmid * ly;
                q += kside * (dm.y + dp.y);
                rsy[j] = q;
            }
            float lxx = kmid * rdx[jB];
            lxx += kside * (rdx[jA] + rdx[j]);
            const float lxy = rsx[j] - rsx[jA];
            const float lyy = rsy[j] - rsy[jA];
            const float d2 = (lxx * lyy - lxy * lxy) * sq;
            // The determinant is read again only AT candidates (their response; the suppression compares responses of candidates) and in
            // the 3 x 3 block around one (sub-pixel fit): with dense_det = 0 nothing is stored here, and a row with candidates stores the
            // three rows of the candidates' columns and of the columns beside them further down (4 of the stage's 18 bytes per pixel less).
            if (a.dense_det) {
                const bool ok = z >= y0 && z < y1;
                const __amdgpu_buffer_rsrc_t r_det = __builtin_amdgcn_make_buffer_rsrc(a.Ldet, 0, ok ? plane4 : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d2), r_det, xo4, __builtin_amdgcn_readfirstlane(ok ? z * w * 4 : 0), 0);
            }
            // ---- extrema of row y = z - 1: strictly above the threshold and above its eight neighbours (rows z - 2, z - 1, z)
            const float hs2 = vmax(lane_prev(d2), lane_next(d2));
            const float hm2 = vmax(hs2, d2);
            {
                // "v <= thr or v <= some neighbour" is "v <= max(thr, neighbours)"; a NaN neighbour drops out of the maximum exactly as its
                // comparison would be false
                const float m = vmax(vmax3(hm0, hs1, hm2), thr);
                const bool y_band = y >= y0 && y < y1;
                const bool y_tested = y_band && (unsigned)(y - border) < (unsigned)(h - 2 * border);
                // "reject if v <= neighbour", as the reference writes it: the negated comparisons keep its NaN behaviour
                const bool keep = x_tested & y_tested & !(d1 <= m);
                const int npx = y_band ? w * h : 0;
                const __amdgpu_buffer_rsrc_t r_mask = __builtin_amdgcn_make_buffer_rsrc(a.mask, 0, npx, 0x00020000);
                const __amdgpu_buffer_rsrc_t r_stat = __builtin_amdgcn_make_buffer_rsrc(a.status, 0, npx, 0x00020000);
                const int o = __builtin_amdgcn_readfirstlane(y_band ? y * w : 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(keep ? 1 : 0), r_mask, xo1, o, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, r_stat, xo1, o, 0);
                const unsigned long long b = __ballot(keep);
                if (b) {   // wave-uniform, rare
                    if (!a.dense_det) {
                        // rows y - 1, y, y + 1 (= z - 2, z - 1, z; all inside the image: candidates keep `border` >= 1 away from its edges) of
                        // the candidates' columns and their two neighbours: every lane stores its own column's three values
                        const unsigned long long near = b | (b << 1) | (b >> 1);
                        // (x4, not xo4: the column beside a candidate in the wave's first / last output column is a halo lane, whose determinant is
                        // as valid as the extrema test that just used it; the other lanes store past the buffer, i.e. nothing)
                        const int vo = ((near >> lane) & 1) ? x4 : DROP;
                        const __amdgpu_buffer_rsrc_t r_det = __builtin_amdgcn_make_buffer_rsrc(a.Ldet, 0, plane4, 0x00020000);
                        const int row = __builtin_amdgcn_readfirstlane((y - 1) * w * 4);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d0), r_det, vo, row, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d1), r_det, vo, row + w * 4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d2), r_det, vo, row + 2 * w * 4, 0);
                    }
                    if (keep) cand[ncand + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0))] = (uint32_t)x | ((uint32_t)y << 16);
                    ncand += __builtin_popcountll(b);
                    if (ncand > DS_CAND - 32) {
                        flush_candidates(list, list_count, cand, ncand, lane);
                        ncand = 0;
                    }
                }
            }
            hm0 = hm1;
            hs1 = hs2;
            hm1 = hm2;
            d0 = d1;
            d1 = d2;
        }
#pragma unroll
        for (int j = 0; j < P; j++) cur[j] = nxt[j];
    }
    if (ncand) flush_candidates(list, list_count, cand, ncand, lane);
}

}  // namespace

// (outside the anonymous namespace: the profilers' kernel names stay readable)
template <int S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void doh_strip_kernel(DohStripArgs a, uint32_t* __restrict__ list, int* __restrict__ list_count, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    a.Lsmooth = bofs(a.Lsmooth, bstride);
    a.Lxy = bofs(a.Lxy, bstride);
    a.Ldet = bofs(a.Ldet, bstride);
    a.mask = bofs(a.mask, bstride);
    a.status = bofs(a.status, bstride);
    list = bofs(list, bstride);
    list_count = bofs(list_count, bstride);
    __shared__ float s_l[4][64];
    __shared__ float2 s_xy[4][64];
    __shared__ uint32_t s_cand[4][DS_CAND];
    const int wv = threadIdx.x >> 6;
    // wave-uniform by construction; readfirstlane tells the compiler, so that row offsets and row conditions live in scalar registers
    const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (id >= a.strips * a.bands) return;   // no block-level barriers in this kernel
    const int band = __builtin_amdgcn_readfirstlane(id / a.strips);
    const int strip = id - band * a.strips;
    constexpr int P = 2 * S + 1;
    const int y0 = band * a.rb;
    // a band whose walk (with the unrolled loop's padding and the prefetch) stays inside the image needs no reflected rows
    const bool vedge = y0 - 1 - 2 * S < 0 || y0 + a.rb + 2 * S + 1 + P >= a.h;
    if (vedge) doh_strip_rows<S, true>(a, list, list_count, strip, band, s_l[wv], s_xy[wv], s_cand[wv]);
    else doh_strip_rows<S, false>(a, list, list_count, strip, band, s_l[wv], s_xy[wv], s_cand[wv]);
}

long long stream_wave_slots(const void* kernel, int dynamic_lds) {
    static std::mutex m;
    static std::map<std::pair<int, const void*>, long long> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev * 1024 + dynamic_lds / 1024;   // (cache key: device and LDS request in KiB)
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find({dev, kernel});
    if (it != cache.end()) return it->second;
    int blocks_per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, 256, (size_t)dynamic_lds) != hipSuccess || blocks_per_cu < 1) {
        (void)hipGetLastError();
        blocks_per_cu = 4;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev / 1024) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 256;
    }
    const long long slots = (long long)blocks_per_cu * cus * 4;
    cache[{dev, kernel}] = slots;
    return slots;
}

// true: the level's (Lx, Ly), det, keypoint mask, suppression status (every pixel of the level: no zero fill needed) and candidate list are
// on their way on `s`. false: not a level for this kernel (the caller takes doh_fused_kernel and clears mask / status itself).
bool doh_strips_eligible(int w, int h, int sc, int batch) {
    const int mode = config().doh_strip;
    if (mode == 0 || sc < 2 || sc > 4 || w < 64 || h < 64) return false;
    // 8 Mpx and more (the first octave of a 4096^2 frame): below that a level has too few 64-column strips to fill the chip with bands of
    // a useful height (2048^2: 41 strips; bands of 16 rows spend half their walk on the 4 s + 2 warm-up rows) and the LDS tiles are quicker
    return mode == 2 || (long long)w * h * batch >= (1ll << 23);
}

bool launch_doh_strips(const float* Lsmooth, float2* Lxy, float* Ldet, int w, int h, int sc, float kside, float kmid, int border, float thr, uint8_t* mask,
                       uint8_t* status, uint32_t* list, int* list_count, hipStream_t s, const Batch& b, bool dense_det) {
    if (!doh_strips_eligible(w, h, sc, b.n)) return false;
    const int vw = 64 - 4 * sc - 2;
    const int strips = ceil_div(w, vw);
    // band height (APDS_DOH_STRIP_ROWS: test hook): about 64 rows - a walk of 64 + 4 s + 2 rows - stretched so that the launch's waves fill the
    // resident wave slots a whole number of times (akaze.h: stream_band_rows)
    const int rb_env = config().doh_strip_rows;
    // (An occupancy cap - unused dynamic LDS, so that the kernel's long-lived waves leave registers to the level chain it runs beside - was
    // measured at 4 / 3 / 2 blocks per CU: 1.71 - 1.76 ms per 4096^2 extraction in every setting, profiles/r03/level_ab.txt. No cap.)
    constexpr int lds_pad = 0;
    auto rows_for = [&](auto kernel) { return rb_env > 0 ? rb_env : stream_band_rows(kernel, strips, h, b.n, 64, 16, lds_pad); };
    const int rb = sc == 2 ? rows_for(&doh_strip_kernel<2>) : sc == 3 ? rows_for(&doh_strip_kernel<3>) : rows_for(&doh_strip_kernel<4>);
    const bool none = border + 1 >= h || w - 2 * border <= 0 || h - 2 * border <= 0;
    DohStripArgs a{Lsmooth, Lxy, Ldet, mask, status, w, h, none ? -1 : border, kside, kmid, (float)(sc * sc * sc * sc), thr, strips,
                   ceil_div(h, rb), rb, dense_det ? 1 : 0};
    const dim3 grid(ceil_div((long long)a.strips * a.bands, 4), 1, b.n);
    switch (sc) {
        case 2: hipLaunchKernelGGL(doh_strip_kernel<2>, grid, dim3(256), lds_pad, s, a, list, list_count, b.stride); break;
        case 3: hipLaunchKernelGGL(doh_strip_kernel<3>, grid, dim3(256), lds_pad, s, a, list, list_count, b.stride); break;
        default: hipLaunchKernelGGL(doh_strip_kernel<4>, grid, dim3(256), lds_pad, s, a, list, list_count, b.stride); break;
    }
    return true;
}

}  // namespace apds
