// csrc/l2_match.hip — brute-force L2 k-NN of float descriptors as a distance GEMM on the matrix cores.
//
// BASELINE.json config 3 / north_star: "MFMA only for the float-descriptor all-pairs L2 distance GEMM". The reference
// itself matches Hamming only (feature_extraction/src/lib.rs:101,121), so this has no reference call site; semantics
// follow cv::BFMatcher(NORM_L2).knnMatch: distance = sqrt(sum (q_i - t_i)^2), ties to the lower train index.
//
//   d^2(q,t) = |q|^2 + |t|^2 - 2 q.t      -> the Q x N dot products are one GEMM, fused with a running top-k so the
//                                            distance matrix never leaves the registers.
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, 157 TFLOP/s peak on MI355X).
// Orientation: train rows are the M dimension, queries the N dimension, so that an accumulator LANE owns ONE query
// column per 16x16 tile (col = lane & 15) and 4 train rows: the running top-2 of a query lives in the lanes that own
// it, with no cross-lane traffic until the end (same trick as the Hamming kernel). Kernel structure: see l2_topk_kernel.
#include <cmath>

#include "kernels.h"

namespace apds {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int L2_TM = 128, L2_TN = 128, L2_KMAX = 128;
static constexpr uint64_t L2_EMPTY = ~0ull;

__global__ void row_norms_kernel(const float* __restrict__ x, long long n, int dim, float* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float v = x[row * dim + i];
        s += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
}

struct Top2 {
    float d0, d1;
    uint32_t i0, i1;
};

__device__ __forceinline__ void top2_insert(Top2& b, float d, uint32_t idx) {
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

__device__ __forceinline__ uint64_t l2_key(float d, uint32_t idx) {
    return idx == 0xFFFFFFFFu ? L2_EMPTY : ((uint64_t)__float_as_uint(d) << 32) | idx;   // d >= 0: bit pattern order == value order
}

// grid: x = query tiles (128 queries), y = splits of the train tiles. out: [split][nq][K] keys
//
// Block = 8 waves = two per SIMD. The block's 128 queries are resident in LDS for the whole kernel; a block tile is 128 train
// rows, of which wave w owns rows [16w, 16w+16): it loads them, stages them in its own LDS slice, multiplies them against
// all 128 queries (eight 16x16 accumulators, v_mfma_f32_16x16x4_f32) and screens the results. Nothing in the tile loop is
// shared between waves, so there is no barrier in it: the two waves of a SIMD drift apart and one wave's non-MFMA work
// (top-2 screening, LDS commit of the next tile, pipeline fill/drain) runs under the other's MFMAs.
// LDS rows keep the k axis as four classes [k%4 == 0 | 1 | 2 | 3] (a 16x16x4 operand lane needs k = 4*step + (lane >> 4)),
// so four consecutive steps of one lane are 16 contiguous bytes: one ds_read_b128 per operand per four steps; row pitch
// KP+4 floats: the 16 lanes of a b128 group cover all 64 banks.
// FULL_ROWS: dim == KP (no zero padding of the k axis, rows 16-byte aligned). KP: k extent staged in LDS (64 or 128).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, bool FULL_ROWS, int KP>
__global__ __launch_bounds__(512) void l2_topk_kernel(const float* __restrict__ train, const float* __restrict__ tnorm, int n_train,
                                                      const float* __restrict__ queries, const float* __restrict__ qnorm, int nq, int dim,
                                                      int tiles_per_split, uint32_t index_base, uint64_t* __restrict__ out) {
    extern __shared__ float l2_lds[];
    constexpr int ST = KP + 4;                       // LDS row pitch in floats (multiple of 4: 16-byte aligned rows)
    constexpr int Q4 = KP >> 2;                      // floats per k class
    float* sQ = l2_lds;                              // 128 x ST
    float* sT = sQ + L2_TN * ST;                     // 8 waves x 16 rows x ST
    float* sTT = sT + L2_TM * ST;                    // 8 x 16 train norms (+inf for rows past the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * L2_TN;
    const int n_tiles = (n_train + L2_TM - 1) / L2_TM;
    const int tile_begin = blockIdx.y * tiles_per_split, tile_end = min(n_tiles, tile_begin + tiles_per_split);
    if (tile_begin >= tile_end) return;

    // queries -> LDS (zero padded to KP columns, rows past nq replicate the last query)
    for (int i = tid; i < L2_TN * KP; i += 512) {
        const int r = i / KP, c = i - r * KP;
        const int qi = min(q0 + r, nq - 1);
        sQ[r * ST + (c & 3) * Q4 + (c >> 2)] = c < dim ? queries[(size_t)qi * dim + c] : 0.f;
    }
    // a wave stages its 16 rows as (row, 4-column group) pieces: 16 * KP / 4 float4 / 64 lanes = KP / 16 pieces per lane,
    // exactly one per MFMA group of the K loop
    constexpr int gshift = KP == 128 ? 5 : 4;        // float4 groups per row = KP / 4 = 32 or 16
    constexpr int groups_per_row = 1 << gshift;
    constexpr int pieces = (16 << gshift) >> 6;      // 8 for KP = 128, 4 for KP = 64
    constexpr int groups = KP >> 4;                  // K-loop groups of 4 k-steps (16 k values)
    static_assert(pieces == groups, "one staged piece per MFMA group");
    float* wT = sT + wave * 16 * ST;
    float* wTT = sTT + wave * 16;
    float4 pre[pieces];
    float pre_norm = INFINITY;                       // lane < 16: the norm of its row of the tile being staged
    bool pre_valid = false;                          // ... and whether that row exists (resolved at the LDS write)
    // rows past the end re-read the last row: their norm is staged as +inf, so they are never selected, and an
    // unconditional load keeps the loads free of branches (a branch around a load made the compiler wait for it at once)
    auto load_piece = [&](int p, int tile) {
        const int piece = p * 64 + lane;
        const int r = piece >> gshift, g = piece & (groups_per_row - 1);
        const int row = min(tile * L2_TM + wave * 16 + r, n_train - 1), c = g * 4;
        const float* src = train + (size_t)row * dim + c;
        float4 v;
        if (FULL_ROWS) v = *reinterpret_cast<const float4*>(src);
        else {
            v.x = c < dim ? src[0] : 0.f;
            v.y = c + 1 < dim ? src[1] : 0.f;
            v.z = c + 2 < dim ? src[2] : 0.f;
            v.w = c + 3 < dim ? src[3] : 0.f;
        }
        pre[p] = v;
    };
    auto load_norm = [&](int tile) {
        if (lane < 16) {
            const int row = tile * L2_TM + wave * 16 + lane;
            pre_norm = tnorm[min(row, n_train - 1)];
            pre_valid = row < n_train;
        }
    };
    auto commit = [&]() {   // wave-local: a wave's LDS operations complete in order, no barrier needed
#pragma unroll
        for (int p = 0; p < pieces; p++) {
            const int piece = p * 64 + lane;
            const int r = piece >> gshift, g = piece & (groups_per_row - 1);
            float* d = &wT[r * ST + g];              // k = 4g + j -> class j, slot g
            d[0] = pre[p].x;
            d[Q4] = pre[p].y;
            d[2 * Q4] = pre[p].z;
            d[3 * Q4] = pre[p].w;
        }
        if (lane < 16) wTT[lane] = pre_valid ? pre_norm : INFINITY;
    };

    Top2 best[8];
    float qq[8];
#pragma unroll
    for (int n = 0; n < 8; n++) {
        best[n].d0 = best[n].d1 = INFINITY;
        best[n].i0 = best[n].i1 = 0xFFFFFFFFu;
        qq[n] = qnorm[min(q0 + n * 16 + (lane & 15), nq - 1)];
    }
    const int kc = lane >> 4;                        // this lane's k class
    const float* aBase = &wT[(lane & 15) * ST + kc * Q4];
    const float* bBase = &sQ[(lane & 15) * ST + kc * Q4];

#pragma unroll
    for (int p = 0; p < pieces; p++) load_piece(p, tile_begin);
    load_norm(tile_begin);
    commit();
    __syncthreads();                                 // publishes sQ (the only data shared between waves)

    for (int tile = tile_begin; tile < tile_end; tile++) {
        const bool more = tile + 1 < tile_end;
        if (more) load_norm(tile + 1);
        // K loop, fully unrolled: operand registers ping-pong between two sets; the nine ds_read_b128s of the next group are
        // issued before the 32 MFMAs of the current group (sched_barrier pins that order)
        f32x4 acc[8];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        float4 opA[2], opB[2][8];
#define L2_LOAD(S, G)                                                                                        \
    opA[S] = *reinterpret_cast<const float4*>(aBase + (G) * 4);                                              \
    _Pragma("unroll") for (int n = 0; n < 8; n++) opB[S][n] = *reinterpret_cast<const float4*>(bBase + n * 16 * ST + (G) * 4);
#define L2_STEP(S, c) \
    _Pragma("unroll") for (int n = 0; n < 8; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(opA[S].c, opB[S][n].c, acc[n], 0, 0, 0);
        L2_LOAD(0, 0)
#pragma unroll
        for (int g = 0; g < groups; g++) {
            const int cur = g & 1;
            if (g + 1 < groups) {
                if (cur) { L2_LOAD(0, g + 1) } else { L2_LOAD(1, g + 1) }
            }
                // all global loads of the next tile go out with the first group: the commit at the end of this tile consumes them,
            // and every later group that carried a load cost a fixed ~300 cycles (1 / 2 / 4 / 8 groups with loads:
            // 86.9 / 85.8 / 83.9 / 81.6 % of the MFMA peak)
            if (more && g == 0) {
#pragma unroll
                for (int p = 0; p < pieces; p++) load_piece(p, tile + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) {   // the first step starts the accumulators from an inline zero instead of register writes
#pragma unroll
                for (int n = 0; n < 8; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(opA[0].x, opB[0][n].x, zero, 0, 0, 0);
                L2_STEP(0, y) L2_STEP(0, z) L2_STEP(0, w)
            } else if (cur) {
                L2_STEP(1, x) L2_STEP(1, y) L2_STEP(1, z) L2_STEP(1, w)
            } else {
                L2_STEP(0, x) L2_STEP(0, y) L2_STEP(0, z) L2_STEP(0, w)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef L2_STEP
#undef L2_LOAD
        // epilogue: candidates are ranked on u = |t|^2 - 2 q.t (|q|^2 is constant per query and added once at the end).
        // A lane holds rows 4*(lane>>4) + r of the wave's 16 for its query column of each of the 8 column blocks:
        // screen all 32 values with min chains, insert only when some lane has a hit
        const float4 t4 = *reinterpret_cast<const float4*>(&wTT[4 * kc]);
        float vals[8][4];
        bool hit = false;
#pragma unroll
        for (int n = 0; n < 8; n++) {
            vals[n][0] = __builtin_fmaf(-2.0f, acc[n][0], t4.x);
            vals[n][1] = __builtin_fmaf(-2.0f, acc[n][1], t4.y);
            vals[n][2] = __builtin_fmaf(-2.0f, acc[n][2], t4.z);
            vals[n][3] = __builtin_fmaf(-2.0f, acc[n][3], t4.w);
            const float mn = fminf(fminf(vals[n][0], vals[n][1]), fminf(vals[n][2], vals[n][3]));
            hit |= mn < best[n].d1;
        }
        if (__any(hit)) {
            const uint32_t row0 = (uint32_t)(tile * L2_TM + wave * 16 + 4 * kc) + index_base;
#pragma unroll
            for (int n = 0; n < 8; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) top2_insert(best[n], vals[n][r], row0 + r);
        }
        if (more) commit();
    }
    // d^2 = max(|q|^2 + u, 0): same value as ranking on d^2 directly, the add is monotone
#pragma unroll
    for (int n = 0; n < 8; n++) {
        best[n].d0 = fmaxf(qq[n] + best[n].d0, 0.f);
        best[n].d1 = fmaxf(qq[n] + best[n].d1, 0.f);
    }
    // merge the 32 partial lists of every query (8 waves x 4 row classes) through LDS, source-major so that the final scan
    // reads conflict-free; the LDS is reused from its start, hence the barrier (slower waves may still read sQ)
    __syncthreads();
    uint64_t* cand = reinterpret_cast<uint64_t*>(l2_lds);      // [32 sources][2][128 queries] = 64 KB
    {
        const int src = wave * 4 + kc;
#pragma unroll
        for (int n = 0; n < 8; n++) {
            const int ql = n * 16 + (lane & 15);
            cand[(src * 2 + 0) * L2_TN + ql] = l2_key(best[n].d0, best[n].i0);
            cand[(src * 2 + 1) * L2_TN + ql] = l2_key(best[n].d1, best[n].i1);
        }
    }
    __syncthreads();
    if (tid < L2_TN && q0 + tid < nq) {
        uint64_t b0 = L2_EMPTY, b1 = L2_EMPTY;
        for (int j = 0; j < 64; j++) {
            const uint64_t key = cand[j * L2_TN + tid];
            if (key < b0) {
                b1 = b0;
                b0 = key;
            } else if (key < b1) {
                b1 = key;
            }
        }
        uint64_t* o = out + ((size_t)blockIdx.y * nq + q0 + tid) * K;
        o[0] = b0;
        if (K == 2) o[1] = b1;
    }
}

void l2_row_norms_device(const float* x, long long n, int dim, float* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(row_norms_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, x, n, dim, out);
}

void l2_topk_device(const float* q, int nq, const float* t, long long nt, int dim, uint32_t index_base, int k, uint64_t* out, hipStream_t s) {
    APDS_REQUIRE(k == 1 || k == 2, APDS_ERR_ASSERT, "L2 top-k supports k in {1,2}");
    APDS_REQUIRE(dim >= 1 && dim <= L2_KMAX, APDS_ERR_ASSERT, "float descriptor length must be 1..128");
    APDS_REQUIRE(nt < (1ll << 31), APDS_ERR_ASSERT, "train set too large for one call; shard it");
    if (nq <= 0) return;
    if (nt <= 0) {
        HIP_CHECK(hipMemsetAsync(out, 0xFF, (size_t)nq * k * 8, s));
        return;
    }
    ThreadCtx& c = ctx();
    float* qn = c.alloc_n<float>(nq);
    float* tn = c.alloc_n<float>(nt);
    hipLaunchKernelGGL(row_norms_kernel, dim3(ceil_div(nq, 4)), dim3(256), 0, s, q, (long long)nq, dim, qn);
    hipLaunchKernelGGL(row_norms_kernel, dim3(ceil_div(nt, 4)), dim3(256), 0, s, t, nt, dim, tn);
    const int kp = dim <= 64 ? 64 : 128;
    const int q_tiles = ceil_div(nq, L2_TN), t_tiles = ceil_div(nt, L2_TM);
    // enough blocks for every CU, but keep each block streaming many train tiles
    int splits = std::max(1, std::min(t_tiles, ceil_div(256 * 2, q_tiles)));
    const int tiles_per_split = ceil_div(t_tiles, splits);
    splits = ceil_div(t_tiles, tiles_per_split);
    uint64_t* parts = splits == 1 ? out : c.alloc_n<uint64_t>((size_t)splits * nq * k);
    const size_t lds = (size_t)(2 * 128 * (kp + 4) + 128) * sizeof(float);
    {
        KernelTimer timer("l2_topk", s);
        const bool full = dim == kp && (reinterpret_cast<uintptr_t>(t) & 15) == 0;
        // the 128-wide tiles need more than the default 64 KB of dynamic LDS: opt in per instantiation (cheap, idempotent)
#define L2_LAUNCH(KK, FF, PP)                                                                                                                            \
    do {                                                                                                                                                 \
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_topk_kernel<KK, FF, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)); \
        hipLaunchKernelGGL((l2_topk_kernel<KK, FF, PP>), dim3(q_tiles, splits), dim3(512), lds, s, t, (const float*)tn, (int)nt, q, (const float*)qn, nq, \
                           dim, tiles_per_split, index_base, parts);                                                                                     \
    } while (0)
#define L2_LAUNCH_K(KK)                       \
    do {                                      \
        if (kp == 128) {                      \
            if (full) L2_LAUNCH(KK, true, 128);  \
            else L2_LAUNCH(KK, false, 128);      \
        } else {                              \
            if (full) L2_LAUNCH(KK, true, 64);   \
            else L2_LAUNCH(KK, false, 64);       \
        }                                     \
    } while (0)
        if (k == 2) L2_LAUNCH_K(2);
        else L2_LAUNCH_K(1);
#undef L2_LAUNCH_K
#undef L2_LAUNCH
    }
    HIP_CHECK(hipGetLastError());
    if (splits > 1) merge_topk_device(parts, splits, nq, k, out, s);
}

}  // namespace apds

using namespace apds;

extern "C" {

int apds_dev_l2_topk_ex(const void* q, int nq, const void* t, int64_t nt, int dim, uint32_t index_base, int k, int mode, void* out_keys, void* stream,
                        int* mode_used, double* candidates_per_query) {
    return guarded([&] {
        APDS_REQUIRE(nq >= 0 && nt >= 0, APDS_ERR_ASSERT, "negative row count");
        APDS_REQUIRE(mode == APDS_L2_EXACT || mode == APDS_L2_SCREEN, APDS_ERR_BAD_ARG, "mode must be APDS_L2_EXACT or APDS_L2_SCREEN");
        ctx().ws_reset();
        hipStream_t s = pick_stream(stream);
        if (mode_used) *mode_used = APDS_L2_EXACT;
        if (candidates_per_query) *candidates_per_query = 0;
        if (mode == APDS_L2_SCREEN && l2_topk_screen_device(static_cast<const float*>(q), nq, static_cast<const float*>(t), nt, dim, index_base, k,
                                                            static_cast<uint64_t*>(out_keys), s, candidates_per_query)) {
            if (mode_used) *mode_used = APDS_L2_SCREEN;
            return;
        }
        ctx().ws_reset();   // the screen does not apply (shape, alignment, candidate overflow): the exact kernel
        l2_topk_device(static_cast<const float*>(q), nq, static_cast<const float*>(t), nt, dim, index_base, k, static_cast<uint64_t*>(out_keys), s);
    });
}

int apds_dev_l2_topk(const void* q, int nq, const void* t, int64_t nt, int dim, uint32_t index_base, int k, void* out_keys, void* stream) {
    return guarded([&] {
        APDS_REQUIRE(nq >= 0 && nt >= 0, APDS_ERR_ASSERT, "negative row count");
        ctx().ws_reset();
        l2_topk_device(static_cast<const float*>(q), nq, static_cast<const float*>(t), nt, dim, index_base, k, static_cast<uint64_t*>(out_keys),
                       pick_stream(stream));
    });
}

int apds_l2_knn_match(const float* q, int nq, const float* t, int nt, int dim, int k, int32_t* idx, float* dist) {
    return guarded([&] {
        APDS_REQUIRE(nq >= 0 && nt >= 0 && k >= 1, APDS_ERR_ASSERT, "bad sizes");
        APDS_REQUIRE(k <= 2, APDS_ERR_ASSERT, "k > 2 is not implemented");
        APDS_REQUIRE((q || !nq) && (t || !nt) && idx && dist, APDS_ERR_BAD_ARG, "null argument");
        if (!nq) return;
        if (!nt) {
            for (long long i = 0; i < (long long)nq * k; i++) idx[i] = -1, dist[i] = INFINITY;
            return;
        }
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        float* dq = c.alloc_n<float>((size_t)nq * dim);
        float* dt = c.alloc_n<float>((size_t)nt * dim);
        uint64_t* keys = c.alloc_n<uint64_t>((size_t)nq * k);
        HIP_CHECK(hipMemcpyAsync(dq, q, (size_t)nq * dim * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(dt, t, (size_t)nt * dim * 4, hipMemcpyHostToDevice, s));
        l2_topk_device(dq, nq, dt, nt, dim, 0, k, keys, s);
        std::vector<uint64_t> h((size_t)nq * k);
        HIP_CHECK(hipMemcpyAsync(h.data(), keys, h.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < h.size(); i++) {
            if (h[i] == ~0ull) idx[i] = -1, dist[i] = INFINITY;
            else {
                union {
                    uint32_t u;
                    float f;
                } cv;
                cv.u = (uint32_t)(h[i] >> 32);
                const float d2 = cv.f;
                idx[i] = (int32_t)(uint32_t)h[i];
                dist[i] = sqrtf(d2);   // BFMatcher(NORM_L2) reports the distance, not its square
            }
        }
    });
}

}  // extern "C"
