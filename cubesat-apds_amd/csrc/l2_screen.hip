// csrc/l2_screen.hip — float-descriptor L2 top-2 as a BF16 MFMA screen + exact f32 re-rank (BASELINE config 3, fast mode).
//
// l2_match.hip computes every one of the Q x N distances with f32 MFMA (157 TFLOP/s peak). The bf16 matrix pipe is 16 times
// faster, and a top-2 search only needs exact arithmetic for the handful of rows that can still be among the two nearest. So:
//
//   pass A  S(q,t) = |q|^2 + |t|^2 - 2 q~.t~ with q~, t~ the operands rounded to bf16, products exact and accumulation in f32
//           inside v_mfma_f32_16x16x32_bf16; fused running top-2 per query -> s2(q), the second smallest screen value. Run over a
//           SAMPLE of the train rows (the first 1/16, at least 8192): its s2 is an upper bound of the whole set's, which is all the
//           proof below needs, and it costs 1/16 of a pass;
//   pass B  the products of ALL rows; every row with S(q,t) <= s2(q) + 2 eps(q) is appended to a candidate list;
//   pass C  the candidates are re-ranked with EXACTLY the arithmetic of l2_match.hip (the same binary32 fmaf chain over k, the same
//           |t|^2 - 2 q.t and |q|^2 + . and max(., 0)), keys (distance bits << 32 | row) reduced by 64-bit atomic min.
//
// Bound (F = the value l2_match.hip computes, S = the screen value): with u = 2^-8 the unit roundoff of bf16,
//   |q_i t_i - q~_i t~_i| <= (2u + u^2) |q_i| |t_i|, summed and with Cauchy-Schwarz: |q.t - q~.t~| <= (2u + u^2) |q| |t|;
//   the f32 accumulations on either side and the few f32 operations around them add at most 2^-13 (|q|^2 + |t|^2) (K = 128 terms
//   at 2^-24 relative each is 2^-17 |q| |t|; the slack is generous on purpose). Hence |F - S| <= eps(q) with
//   eps(q) = 2 (2u + u^2) |q| Tmax + 2^-13 (|q|^2 + Tmax^2),   Tmax = the largest row norm of the train set.
// Two rows (of the sample, hence of the set) have S <= s2, hence F <= s2 + eps, so the second smallest F over the whole set is
// <= s2 + eps; every row of the exact top-2 (ties included) has F <= that, hence S <= s2 + 2 eps: it is a candidate. (A looser s2
// only admits more candidates: a sample of 1/12 of the rows admits ~2 * 12 * 2.5 = 60 per query on BASELINE config 3, ~5 with the
// exact s2: 25 ms more re-rank against 190 ms less screening.) The re-rank therefore returns the keys l2_match.hip returns, bit for bit
// (tests/test_l2_match_gpu.py compares the two modes directly).
//
// Screen kernel (CDNA4): block = 8 waves = 384 queries x a stream of 128-row train tiles. A wave keeps its 48 queries (three 16-column
// blocks) as MFMA B operands in registers for the whole kernel (bf16: 48 VGPRs); train tiles are staged in LDS (bf16, pre-scaled by
// -2, row pitch 272 B so that the 16 rows of a ds_read_b128 group fall in different banks), double buffered, one barrier per tile;
// each A read feeds three MFMAs. The accumulator starts from |t|^2 of its four rows, so an accumulator IS the ranking value
// |t|^2 - 2 q~.t~ and the epilogue is two min and one compare per 16 x 16 block; like in the other matchers a lane owns one query
// column, so the running top-2 (or the threshold) lives in the lane and insertions / appends are rare.
// (Round 4, built, verified and removed - commit a7e97b9: the same two passes on v_mfma_f32_32x32x16_bf16 - 64 resident queries per wave,
// twice the math per issued MFMA, a third more per LDS byte. 144 - 149 VGPRs = three waves per SIMD, so blocks of twelve waves: keys
// identical, 252 ms per screen instead of 209 (0.43 against 0.51 of the bf16 peak; profiles/r04/l2_screen_mfma32_ab.txt). The
// sixteen-value minimum tree per accumulator and one wave less per SIMD cost more than the wider instruction returns.)
// (Round 4, from the Hamming matcher, also measured here: tiles by LDS-DMA into an unpadded XOR-swizzled image instead of register staging -
// keys identical, 108 instead of 122 VGPRs, 231 ms per step instead of 225.5 (0.506 against 0.519): this kernel's gaps are not where the
// staging instructions are. The wave-priority change is kept: 0.514 -> 0.519. profiles/r04/l2_prio_ab.txt, l2_glds.txt.)
#include <cmath>

#include "config.h"
#include "kernels.h"

namespace apds {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int SC_TM = 128;            // train rows per tile
#ifndef APDS_SC_NC
#define APDS_SC_NC 3
#endif
static constexpr int SC_NC = APDS_SC_NC;     // 16-query column blocks per wave (each A read from LDS feeds SC_NC MFMAs)
static constexpr int SC_Q = 8 * 16 * SC_NC;  // queries per block (8 waves)
static constexpr int SC_D = 128;             // descriptor length (the screen is built for it)
static constexpr int SC_PITCH = 272;         // bytes per staged train row (256 + 16)
static constexpr uint64_t SC_EMPTY = ~0ull;

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t x = __float_as_uint(f);
    if ((x & 0x7F800000u) == 0x7F800000u) return (uint16_t)(x >> 16) | ((x & 0xFFFFu) ? 0x40 : 0);   // inf / nan
    x += 0x7FFFu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

// rows of f32 -> bf16 (scaled by `scale`, a power of two: exact), and |row|^2 in f32 exactly as row_norms_kernel of l2_match.hip
__global__ void to_bf16_rows_kernel(const float* __restrict__ x, long long n, float scale, uint16_t* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n * SC_D) return;
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    uint2 o;
    o.x = (uint32_t)f32_to_bf16_rne(v.x * scale) | ((uint32_t)f32_to_bf16_rne(v.y * scale) << 16);
    o.y = (uint32_t)f32_to_bf16_rne(v.z * scale) | ((uint32_t)f32_to_bf16_rne(v.w * scale) << 16);
    *reinterpret_cast<uint2*>(out + i) = o;
}

__global__ void max_norm_kernel(const float* __restrict__ norms_sq, long long n, unsigned int* __restrict__ out_bits) {
    APDS_RAISE_WAVE_PRIORITY();
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, norms_sq[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));   // non-negative floats order like their bits
}

struct ScTop2 {
    float d0, d1;
    uint32_t i0, i1;
};
__device__ __forceinline__ void sc_insert(ScTop2& b, float d, uint32_t idx) {
    if (d < b.d1) {
        if (d < b.d0) {
            b.d1 = b.d0;
            b.i1 = b.i0;
            b.d0 = d;
            b.i0 = idx;
        } else {
            b.d1 = d;
            b.i1 = idx;
        }
    }
}
__device__ __forceinline__ uint64_t sc_key(float d, uint32_t idx) { return idx == 0xFFFFFFFFu ? SC_EMPTY : ((uint64_t)__float_as_uint(d) << 32) | idx; }

// PASS 0: running top-2 of the screen value per query -> out[split][nq][2] keys (d^2 = max(|q|^2 + u, 0) bits << 32 | row).
// PASS 1: rows with u <= theta[q] are appended to the block's own region of cand (query << 32 | row; region = cand_cap entries per
//         block, position from an LDS counter: one global counter for all blocks serialised at ~12 ns per append, 0.65 s for the
//         54 M candidates of config 3); the block's count goes to cand_count[block] (entries past the region are dropped and show
//         in the count: the host checks it).
template <int PASS, bool PRIO>
__global__ __launch_bounds__(512) void l2_screen_kernel(const uint16_t* __restrict__ train_bf, const float* __restrict__ tnorm, int n_train,
                                                        const uint16_t* __restrict__ query_bf, const float* __restrict__ qnorm, int nq, int tiles_per_split,
                                                        uint32_t index_base, uint64_t* __restrict__ out, const float* __restrict__ theta,
                                                        unsigned long long* __restrict__ cand, unsigned long long cand_cap,
                                                        unsigned long long* __restrict__ cand_count) {
    if (!PRIO) APDS_RAISE_WAVE_PRIORITY();
    // 16-byte aligned: the static LDS word below would otherwise push this array to offset 4 and turn every ds_read_b128 /
    // ds_write_b128 of the tiles into misaligned accesses (measured: 209 -> 1185 ms per pass)
    extern __shared__ __attribute__((aligned(128))) unsigned char sc_lds[];
    auto tile_lds = [&](int buf) { return sc_lds + buf * (SC_TM * SC_PITCH); };
    auto norm_lds = [&](int buf) { return reinterpret_cast<float*>(sc_lds + 2 * SC_TM * SC_PITCH) + buf * SC_TM; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * SC_Q + wave * 16 * SC_NC;         // this wave's queries
    const int n_tiles = (n_train + SC_TM - 1) / SC_TM;
    const int tile_begin = blockIdx.y * tiles_per_split, tile_end = min(n_tiles, tile_begin + tiles_per_split);
    const int block_id = blockIdx.y * gridDim.x + blockIdx.x;
    __shared__ unsigned int s_cand_n;
    if (PASS == 1 && tid == 0) s_cand_n = 0;
    if (tile_begin >= tile_end) {
        if (PASS == 1 && tid == 0) cand_count[block_id] = 0;
        return;
    }
    const int col = lane & 15, kq = lane >> 4;                     // accumulator column / k chunk (operands) / row group (accumulators)

    // B operands: query (q0 + 16 c + col), k = 32 s + 8 kq .. + 7, for c = 0, 1 and the four k steps s
    bf16x8 B[SC_NC][4];
    float qq[SC_NC], th[SC_NC];
#pragma unroll
    for (int c = 0; c < SC_NC; c++) {
        const int qi = min(q0 + 16 * c + col, nq - 1);
#pragma unroll
        for (int s = 0; s < 4; s++) B[c][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(query_bf + (size_t)qi * SC_D + 32 * s + 8 * kq));
        qq[c] = qnorm[qi];
        th[c] = PASS == 1 ? theta[qi] : 0.f;
    }
    ScTop2 best[SC_NC];
#pragma unroll
    for (int c = 0; c < SC_NC; c++) {
        best[c].d0 = best[c].d1 = INFINITY;
        best[c].i0 = best[c].i1 = 0xFFFFFFFFu;
    }

    // staging: the tile is 128 rows x 256 B = 2048 pieces of 16 B, four per thread; rows past the end re-read the last row and get
    // a norm of +inf, so they never rank
    uint4 pre0, pre1, pre2, pre3;   // (separate variables: as an array indexed inside the lambdas they went to scratch)
    float pre_norm = INFINITY;
    const int pr = tid >> 4, pg = tid & 15;   // piece p of this thread: row 32 p + pr, 16-byte group pg
    auto load_tile = [&](int tile) {
        const int r0 = tile * SC_TM + pr;
        pre0 = *reinterpret_cast<const uint4*>(train_bf + (size_t)min(r0, n_train - 1) * SC_D + 8 * pg);
        pre1 = *reinterpret_cast<const uint4*>(train_bf + (size_t)min(r0 + 32, n_train - 1) * SC_D + 8 * pg);
        pre2 = *reinterpret_cast<const uint4*>(train_bf + (size_t)min(r0 + 64, n_train - 1) * SC_D + 8 * pg);
        pre3 = *reinterpret_cast<const uint4*>(train_bf + (size_t)min(r0 + 96, n_train - 1) * SC_D + 8 * pg);
        if (tid < SC_TM) {
            const int row = tile * SC_TM + tid;
            pre_norm = row < n_train ? tnorm[row] : INFINITY;
        }
    };
    auto commit = [&](int buf) {
        unsigned char* d = tile_lds(buf) + pr * SC_PITCH + 16 * pg;
        *reinterpret_cast<uint4*>(d) = pre0;
        *reinterpret_cast<uint4*>(d + 32 * SC_PITCH) = pre1;
        *reinterpret_cast<uint4*>(d + 64 * SC_PITCH) = pre2;
        *reinterpret_cast<uint4*>(d + 96 * SC_PITCH) = pre3;
        if (tid < SC_TM) norm_lds(buf)[tid] = pre_norm;
    };
    load_tile(tile_begin);
    commit(0);
    __syncthreads();

    for (int tile = tile_begin; tile < tile_end; tile++) {
        const int buf = (tile - tile_begin) & 1;
        const bool more = tile + 1 < tile_end;
        if (more) load_tile(tile + 1);
        const unsigned char* T = tile_lds(buf);
        const float* Nn = norm_lds(buf);
#pragma unroll 2
        for (int rb = 0; rb < 8; rb++) {                           // 16-row blocks of the tile
            const f32x4 init = *reinterpret_cast<const f32x4*>(Nn + rb * 16 + 4 * kq);   // |t|^2 of this lane's four rows
            f32x4 acc[SC_NC];
#pragma unroll
            for (int c = 0; c < SC_NC; c++) acc[c] = init;
            const unsigned char* arow = T + (rb * 16 + col) * SC_PITCH + 16 * kq;
            // (round 4, from the Hamming matcher: a wave about to feed the matrix pipe goes ahead of the SIMD's waves that are in their epilogue)
            if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bf16x8 A = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arow + 64 * s));
#pragma unroll
                for (int c = 0; c < SC_NC; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B[c][s], acc[c], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            // epilogue: acc = |t|^2 - 2 q~.t~ for rows 4 kq + j of the block, query column `col` of each of the wave's query blocks
            float mn[SC_NC];
            bool any_hit = false;
#pragma unroll
            for (int c = 0; c < SC_NC; c++) {
                mn[c] = fminf(fminf(acc[c][0], acc[c][1]), fminf(acc[c][2], acc[c][3]));
                any_hit |= PASS == 0 ? mn[c] < best[c].d1 : mn[c] <= th[c];
            }
            const uint32_t row0 = (uint32_t)(tile * SC_TM + rb * 16 + 4 * kq);
            if (__any(any_hit)) {
                if (PASS == 0) {
#pragma unroll
                    for (int c = 0; c < SC_NC; c++)
#pragma unroll
                        for (int j = 0; j < 4; j++) sc_insert(best[c], acc[c][j], row0 + j + index_base);
                } else {
#pragma unroll
                    for (int c = 0; c < SC_NC; c++) {
                        const int qi = q0 + 16 * c + col;
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (acc[c][j] <= th[c] && qi < nq && (int)(row0 + j) < n_train) {
                                const unsigned int pos = atomicAdd(&s_cand_n, 1u);
                                if (pos < cand_cap) cand[(size_t)block_id * cand_cap + pos] = ((unsigned long long)(uint32_t)qi << 32) | (row0 + j + index_base);
                            }
                    }
                }
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();   // the next tile is staged; everybody is done with this one
    }
    if (PASS == 1 && tid == 0) cand_count[block_id] = s_cand_n;   // (after the loop's closing barrier: every append has happened)
    if (PASS == 0) {
        // a query column lives in four lanes (kq = 0..3, different rows): fold them with shuffles, lanes 0..15 write
#pragma unroll
        for (int c = 0; c < SC_NC; c++) {
            ScTop2 b = best[c];
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const float od0 = __shfl_xor(b.d0, off), od1 = __shfl_xor(b.d1, off);
                const uint32_t oi0 = (uint32_t)__shfl_xor((int)b.i0, off), oi1 = (uint32_t)__shfl_xor((int)b.i1, off);
                // merge two sorted pairs; equal values: lower row first
                ScTop2 m = b;
                auto ins = [&](float d, uint32_t i) {
                    if (i == 0xFFFFFFFFu) return;
                    if (d < m.d0 || (d == m.d0 && i < m.i0)) {
                        m.d1 = m.d0;
                        m.i1 = m.i0;
                        m.d0 = d;
                        m.i0 = i;
                    } else if ((d < m.d1 || (d == m.d1 && i < m.i1)) && i != m.i0) {
                        m.d1 = d;
                        m.i1 = i;
                    }
                };
                ins(od0, oi0);
                ins(od1, oi1);
                b = m;
            }
            const int qi = q0 + 16 * c + col;
            if (kq == 0 && qi < nq) {
                uint64_t* o = out + ((size_t)blockIdx.y * nq + qi) * 2;
                o[0] = sc_key(fmaxf(qq[c] + b.d0, 0.f), b.i0);
                o[1] = sc_key(fmaxf(qq[c] + b.d1, 0.f), b.i1);
            }
        }
    }
}

// theta[q] (in units of u = d^2 - |q|^2): second screen distance + 2 eps(q) - |q|^2
__global__ void screen_theta_kernel(const uint64_t* __restrict__ top2, const float* __restrict__ qnorm, int nq, const unsigned int* __restrict__ tmax_sq_bits,
                                    float* __restrict__ theta) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint64_t k2 = top2[(size_t)i * 2 + 1];
    if (k2 == SC_EMPTY) {   // fewer than two rows: everything is a candidate
        theta[i] = INFINITY;
        return;
    }
    const float s2 = __uint_as_float((uint32_t)(k2 >> 32));
    const float qn = qnorm[i], tm = __uint_as_float(*tmax_sq_bits);
    const float u = 0.00390625f;   // 2^-8
    const float eps = 2.0f * (2.0f * u + u * u) * sqrtf(qn) * sqrtf(tm) + (qn + tm) * 0.0001220703125f;   // + 2^-13 (|q|^2 + Tmax^2)
    // s2 is max(|q|^2 + u2, 0): if the clamp was active the true value is <= 0, and using 0 only loosens the threshold
    theta[i] = (s2 + 2.0f * eps) * 1.0000002f - qn;
}

// exact keys of the candidates: the binary32 arithmetic of l2_topk_kernel (fmaf chain over k ascending from 0, u = fmaf(-2, dot, |t|^2),
// d^2 = max(|q|^2 + u, 0)); STEP 0: atomic min into best[q]; STEP 1: atomic min into second[q] of the keys != best[q]
// grid.y = the screen kernel's blocks (one candidate region each), grid.x strides over the region's entries
template <int STEP>
__global__ void screen_rerank_kernel(const unsigned long long* __restrict__ cand, const unsigned long long* __restrict__ counts, unsigned long long region,
                                     const float* __restrict__ q, const float* __restrict__ qnorm, const float* __restrict__ t, const float* __restrict__ tnorm,
                                     uint32_t index_base, unsigned long long* __restrict__ keys, unsigned long long* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const unsigned long long n = min(counts[blockIdx.y], region);
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long i = (unsigned long long)blockIdx.y * region + e;
    const unsigned long long c = cand[i];
    const uint32_t qi = (uint32_t)(c >> 32), row = (uint32_t)c;
    unsigned long long key;
    if (STEP == 0) {
        const float4* a = reinterpret_cast<const float4*>(t + (size_t)(row - index_base) * SC_D);
        const float4* b = reinterpret_cast<const float4*>(q + (size_t)qi * SC_D);
        float dot = 0.0f;
#pragma unroll 8
        for (int k = 0; k < SC_D / 4; k++) {
            const float4 x = a[k], y = b[k];
            dot = __builtin_fmaf(x.x, y.x, dot);
            dot = __builtin_fmaf(x.y, y.y, dot);
            dot = __builtin_fmaf(x.z, y.z, dot);
            dot = __builtin_fmaf(x.w, y.w, dot);
        }
        const float u = __builtin_fmaf(-2.0f, dot, tnorm[row - index_base]);
        const float d2 = fmaxf(qnorm[qi] + u, 0.f);
        key = ((unsigned long long)__float_as_uint(d2) << 32) | row;
        keys[i] = key;
        atomicMin(&out[(size_t)qi * 2], key);
    } else {
        key = keys[i];
        if (key != out[(size_t)qi * 2]) atomicMin(&out[(size_t)qi * 2 + 1], key);
    }
    }
}

// returns false if the screen could not be used (shape, alignment, candidate overflow): the caller falls back to the f32 kernel
bool l2_topk_screen_device(const float* q, int nq, const float* t, long long nt, int dim, uint32_t index_base, int k, uint64_t* out, hipStream_t s,
                           double* candidates_per_query) {
    if (dim != SC_D || k != 2 || nt < 2 || nq < 1) return false;
    if (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(t)) & 15) != 0) return false;
    ThreadCtx& c = ctx();
    float* qn = c.alloc_n<float>(nq);
    float* tn = c.alloc_n<float>(nt);
    uint16_t* qb = c.alloc_n<uint16_t>((size_t)nq * SC_D);
    uint16_t* tb = c.alloc_n<uint16_t>((size_t)nt * SC_D);
    unsigned int* tmax = c.alloc_n<unsigned int>(4);
    HIP_CHECK(hipMemsetAsync(tmax, 0, 16, s));
    l2_row_norms_device(q, nq, dim, qn, s);
    l2_row_norms_device(t, nt, dim, tn, s);
    hipLaunchKernelGGL(to_bf16_rows_kernel, dim3(ceil_div((long long)nq * SC_D / 4, 256)), dim3(256), 0, s, q, (long long)nq, 1.0f, qb);
    hipLaunchKernelGGL(to_bf16_rows_kernel, dim3(ceil_div(nt * SC_D / 4, 256)), dim3(256), 0, s, t, nt, -2.0f, tb);
    hipLaunchKernelGGL(max_norm_kernel, dim3(256), dim3(256), 0, s, (const float*)tn, nt, tmax);
    const int q_tiles = ceil_div(nq, SC_Q), t_tiles = ceil_div(nt, SC_TM);
    int splits = std::max(1, std::min(t_tiles, ceil_div(256 * 2, q_tiles)));
    const int tiles_per_split = ceil_div(t_tiles, splits);
    splits = ceil_div(t_tiles, tiles_per_split);
    // pass A runs over a sample of the rows: the first n_sample (whole tiles)
    // (a threshold taken from a fraction f of the rows admits ~2 / f rows of the whole set, times ~2.5 for the 2 eps margin: pass A
    // costs f of a pass, the re-rank ~0.36 ms per candidate per query at config 3 => the sum is flat around f = 1/8 .. 1/16)
    const int sample_div = config().l2_sample_div;
    const long long n_sample = std::min<long long>(nt, std::max<long long>(8192, (nt / sample_div + SC_TM - 1) / SC_TM * SC_TM));
    const int s_tiles = ceil_div(n_sample, SC_TM);
    int s_splits = std::max(1, std::min(s_tiles, ceil_div(256 * 2, q_tiles)));
    const int s_tiles_per_split = ceil_div(s_tiles, s_splits);
    s_splits = ceil_div(s_tiles, s_tiles_per_split);
    uint64_t* parts = c.alloc_n<uint64_t>((size_t)s_splits * nq * 2);
    uint64_t* top2 = s_splits == 1 ? parts : c.alloc_n<uint64_t>((size_t)nq * 2);
    float* theta = c.alloc_n<float>(nq);
    // candidate regions: one per block of pass B, `region` entries each (256 queries x 256 candidates: 4x what config 3 needs)
    const int n_blocks = q_tiles * splits;
    const unsigned long long region = (unsigned long long)SC_Q * 192;
    if ((unsigned long long)n_blocks * region > (1ull << 29)) return false;
    unsigned long long* cand = c.alloc_n<unsigned long long>((size_t)n_blocks * region);
    unsigned long long* keys = c.alloc_n<unsigned long long>((size_t)n_blocks * region);
    unsigned long long* counts = c.alloc_n<unsigned long long>(n_blocks);
    const size_t lds = (size_t)2 * SC_TM * SC_PITCH + 2 * SC_TM * sizeof(float);
    const bool prio = config().l2_prio != 0;
    for (const void* kf : {reinterpret_cast<const void*>(&l2_screen_kernel<0, false>), reinterpret_cast<const void*>(&l2_screen_kernel<1, false>),
                           reinterpret_cast<const void*>(&l2_screen_kernel<0, true>), reinterpret_cast<const void*>(&l2_screen_kernel<1, true>)})
        HIP_CHECK(hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        KernelTimer timer("l2_screen", s);
        auto k0 = prio ? &l2_screen_kernel<0, true> : &l2_screen_kernel<0, false>;
        hipLaunchKernelGGL(k0, dim3(q_tiles, s_splits), dim3(512), lds, s, (const uint16_t*)tb, (const float*)tn, (int)n_sample,
                           (const uint16_t*)qb, (const float*)qn, nq, s_tiles_per_split, index_base, parts, (const float*)nullptr, (unsigned long long*)nullptr, 0ull,
                           (unsigned long long*)nullptr);
    }
    if (s_splits > 1) merge_topk_device(parts, s_splits, nq, 2, top2, s);
    hipLaunchKernelGGL(screen_theta_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, (const uint64_t*)top2, (const float*)qn, nq, (const unsigned int*)tmax, theta);
    {
        KernelTimer timer("l2_screen", s);
        auto k1 = prio ? &l2_screen_kernel<1, true> : &l2_screen_kernel<1, false>;
        hipLaunchKernelGGL(k1, dim3(q_tiles, splits), dim3(512), lds, s, (const uint16_t*)tb, (const float*)tn, (int)nt, (const uint16_t*)qb,
                           (const float*)qn, nq, tiles_per_split, index_base, (uint64_t*)nullptr, (const float*)theta, cand, region, counts);
    }
    std::vector<unsigned long long> hc(n_blocks);
    HIP_CHECK(hipMemcpyAsync(hc.data(), counts, (size_t)n_blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    unsigned long long n_cand = 0, largest = 0;
    for (unsigned long long v : hc) {
        n_cand += v;
        largest = std::max(largest, v);
    }
    if (candidates_per_query) *candidates_per_query = (double)n_cand / nq;
    if (largest > region) return false;   // pathological input (near-duplicate rows everywhere): the exact kernel takes over
    HIP_CHECK(hipMemsetAsync(out, 0xFF, (size_t)nq * 2 * sizeof(uint64_t), s));
    if (n_cand) {
        const dim3 grid((unsigned int)std::max<unsigned long long>(1, std::min<unsigned long long>(64, (largest + 255) / 256)), n_blocks);
        KernelTimer timer("l2_rerank", s);
        hipLaunchKernelGGL((screen_rerank_kernel<0>), grid, dim3(256), 0, s, (const unsigned long long*)cand, (const unsigned long long*)counts, region, q,
                           (const float*)qn, t, (const float*)tn, index_base, keys, reinterpret_cast<unsigned long long*>(out));
        hipLaunchKernelGGL((screen_rerank_kernel<1>), grid, dim3(256), 0, s, (const unsigned long long*)cand, (const unsigned long long*)counts, region, q,
                           (const float*)qn, t, (const float*)tn, index_base, keys, reinterpret_cast<unsigned long long*>(out));
    }
    HIP_CHECK(hipGetLastError());
    return true;
}

}  // namespace apds
