// csrc/l2_match.hip — brute-force L2 k-NN of float descriptors as a distance GEMM on the matrix cores.
//
// BASELINE.json config 3 / north_star: "MFMA only for the float-descriptor all-pairs L2 distance GEMM". The reference
// itself matches Hamming only (feature_extraction/src/lib.rs:101,121), so this has no reference call site; semantics
// follow cv::BFMatcher(NORM_L2).knnMatch: distance = sqrt(sum (q_i - t_i)^2), ties to the lower train index.
//
//   d^2(q,t) = |q|^2 + |t|^2 - 2 q.t      -> the Q x N dot products are one GEMM, fused with a running top-k so the
//                                            distance matrix never leaves the registers.
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, 157 TFLOP/s peak on MI355X).
// Orientation: train rows are the M dimension, queries the N dimension, so that an accumulator LANE owns ONE query
// column (col = lane & 31) and 16 train rows per 32x32 tile: the running top-2 of a query lives in the lane that
// owns it, with no cross-lane traffic until the end (same trick as the Hamming kernel).
// Block = 4 waves, tile = 128 train rows x 128 queries x K (<= 128) resident in LDS. A 32x32x2 operand lane needs
// k = 2*step + (lane >> 5): rows are stored with the k axis split into [even k | odd k], so four consecutive steps of
// one lane are 16 contiguous bytes -> one ds_read_b128 per operand per four steps (row pitch K+4 floats: the 16 lanes
// of a b128 group cover all 64 banks), issued one group ahead of the MFMAs that consume them. The next train tile is
// prefetched into registers under the MFMAs.
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
// FULL_ROWS: dim == KP (no zero padding of the k axis, rows 16-byte aligned). KP: k extent staged in LDS (64 or 128).
template <int K, bool FULL_ROWS, int KP>
__global__ __launch_bounds__(256) void l2_topk_kernel(const float* __restrict__ train, const float* __restrict__ tnorm, int n_train,
                                                      const float* __restrict__ queries, const float* __restrict__ qnorm, int nq, int dim,
                                                      int tiles_per_split, uint32_t index_base, uint64_t* __restrict__ out) {
    extern __shared__ float l2_lds[];
    constexpr int kp = KP;
    constexpr int ST = KP + 4;                       // LDS row pitch in floats (multiple of 4: 16-byte aligned rows)
    constexpr int half = KP >> 1;                    // columns [0, half): even k, [half, kp): odd k
    float* sQ = l2_lds;                              // 128 x ST
    float* sT = sQ + L2_TN * ST;                     // 128 x ST
    float* sTT = sT + L2_TM * ST;                    // 128 train norms (+inf for rows past the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int q0 = blockIdx.x * L2_TN;
    const int n_tiles = (n_train + L2_TM - 1) / L2_TM;
    const int tile_begin = blockIdx.y * tiles_per_split, tile_end = min(n_tiles, tile_begin + tiles_per_split);
    if (tile_begin >= tile_end) return;

    // queries -> LDS (zero padded to kp columns, rows past nq replicate the last query)
    for (int i = tid; i < L2_TN * kp; i += 256) {
        const int r = i / kp, c = i - r * kp;
        const int qi = min(q0 + r, nq - 1);
        sQ[r * ST + (c & 1) * half + (c >> 1)] = c < dim ? queries[(size_t)qi * dim + c] : 0.f;
    }
    // each thread stages a fixed set of (row, 4-column group) pieces of a train tile: 128 * KP / 4 float4 / 256 threads = KP / 8
    // pieces, i.e. exactly one per MFMA group of the K loop
    constexpr int gshift = KP == 128 ? 5 : 4;        // float4 groups per row = KP / 4 = 32 or 16
    constexpr int groups_per_row = 1 << gshift;
    constexpr int pieces = (L2_TM << gshift) >> 8;   // 16 for KP = 128, 8 for KP = 64
    float4 pre[pieces];
    float pre_norm = INFINITY;                       // this thread's row norm of the prefetched tile (threads < 128)
    bool pre_valid = false;                          // ... and whether that row exists (resolved at commit: no use of the load here)
    // rows past the end re-read the last row: their norm is staged as +inf, so they are never selected, and an
    // unconditional load keeps the loads free of branches (a branch around a load made the compiler wait for it at once)
    auto prefetch_piece = [&](int p, int tile) {
        const int piece = p * 256 + tid;
        const int r = piece >> gshift, g = piece & (groups_per_row - 1);
        const int row = min(tile * L2_TM + r, n_train - 1), c = g * 4;
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
    auto prefetch_norm = [&](int tile) {
        if (tid < L2_TM) {
            const int row = tile * L2_TM + tid;
            pre_norm = tnorm[min(row, n_train - 1)];
            pre_valid = row < n_train;
        }
    };
    auto commit = [&](int tile) {
#pragma unroll
        for (int p = 0; p < pieces; p++) {
            const int piece = p * 256 + tid;
            const int r = piece >> gshift, g = piece & (groups_per_row - 1);
            float* d = &sT[r * ST + g * 2];          // k = 4g..4g+3 -> even slots 2g, 2g+1 and odd slots half+2g, half+2g+1
            *reinterpret_cast<float2*>(d) = make_float2(pre[p].x, pre[p].z);
            *reinterpret_cast<float2*>(d + half) = make_float2(pre[p].y, pre[p].w);
        }
        if (tid < L2_TM) sTT[tid] = pre_valid ? pre_norm : INFINITY;
    };

    Top2 best[2];
    float qq[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        best[n].d0 = best[n].d1 = INFINITY;
        best[n].i0 = best[n].i1 = 0xFFFFFFFFu;
        qq[n] = qnorm[min(q0 + wc * 64 + n * 32 + (lane & 31), nq - 1)];
    }
    const int kk = lane >> 5;                        // this lane's k parity within a k-step of 2
    const float* aBase = &sT[(wr * 64 + (lane & 31)) * ST + kk * half];
    const float* bBase = &sQ[(wc * 64 + (lane & 31)) * ST + kk * half];
    constexpr int groups = KP >> 3;                  // 4 k-steps (8 k values) per group; == pieces

#pragma unroll
    for (int p = 0; p < pieces; p++) prefetch_piece(p, tile_begin);
    prefetch_norm(tile_begin);
    for (int tile = tile_begin; tile < tile_end; tile++) {
        __syncthreads();                             // everyone is done reading sT of the previous tile
        commit(tile);
        __syncthreads();
        const bool more = tile + 1 < tile_end;
        if (more) prefetch_norm(tile + 1);
        // K loop, fully unrolled: operand registers ping-pong between two sets; the ds_read_b128s of the next group and ONE
        // global load of the next train tile are issued before the 16 MFMAs of the current group (sched_barrier pins that
        // order), so LDS latency and the issue cost of the vector-memory instructions sit under ~1000 cycles of MFMA
        // (all 16 loads in front of the loop kept the matrix pipe idle for ~8 % of the tile).
        f32x16 acc[2][2];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float4 opA[2][2], opB[2][2];                 // [set][row block]
#define L2_LOAD(S, G)                                                                  \
    opA[S][0] = *reinterpret_cast<const float4*>(aBase + (G) * 4);                     \
    opA[S][1] = *reinterpret_cast<const float4*>(aBase + 32 * ST + (G) * 4);           \
    opB[S][0] = *reinterpret_cast<const float4*>(bBase + (G) * 4);                     \
    opB[S][1] = *reinterpret_cast<const float4*>(bBase + 32 * ST + (G) * 4);
#define L2_STEP(S, c)                                                                                      \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[S][0].c, opB[S][0].c, acc[0][0], 0, 0, 0);      \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[S][0].c, opB[S][1].c, acc[0][1], 0, 0, 0);      \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[S][1].c, opB[S][0].c, acc[1][0], 0, 0, 0);      \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[S][1].c, opB[S][1].c, acc[1][1], 0, 0, 0);
        L2_LOAD(0, 0)
#pragma unroll
        for (int g = 0; g < groups; g++) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < groups) {
                if (nxt) { L2_LOAD(1, g + 1) } else { L2_LOAD(0, g + 1) }
            }
            if (more) prefetch_piece(g, tile + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) {   // the first step starts the accumulators from an inline zero instead of 64 register writes
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][0].x, opB[0][0].x, zero, 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][0].x, opB[0][1].x, zero, 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][1].x, opB[0][0].x, zero, 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][1].x, opB[0][1].x, zero, 0, 0, 0);
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
        // epilogue: candidates are ranked on u = |t|^2 - 2 q.t (|q|^2 is constant per query and added once at the end);
        // the lane's 32 train rows per n are screened with one min chain and inserted only on a hit
        float4 tt[2][4];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int rb = 0; rb < 4; rb++) tt[m][rb] = *reinterpret_cast<const float4*>(&sTT[wr * 64 + m * 32 + 8 * rb + 4 * (lane >> 5)]);
#pragma unroll
        for (int n = 0; n < 2; n++) {
            float vals[2][16];
            float mn = INFINITY;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float4 t4 = tt[m][r >> 2];
                    const float tr = (r & 3) == 0 ? t4.x : ((r & 3) == 1 ? t4.y : ((r & 3) == 2 ? t4.z : t4.w));
                    const float v = __builtin_fmaf(-2.0f, acc[m][n][r], tr);
                    vals[m][r] = v;
                    mn = fminf(mn, v);
                }
            if (__any(mn < best[n].d1)) {
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int rl = wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        top2_insert(best[n], vals[m][r], (uint32_t)(tile * L2_TM + rl) + index_base);
                    }
            }
        }
    }
    // d^2 = max(|q|^2 + u, 0): same value as ranking on d^2 directly, the add is monotone
#pragma unroll
    for (int n = 0; n < 2; n++) {
        best[n].d0 = fmaxf(qq[n] + best[n].d0, 0.f);
        best[n].d1 = fmaxf(qq[n] + best[n].d1, 0.f);
    }
    // merge the 4 partial lists of every query (2 lane halves x 2 row-waves) through LDS (reusing sT)
    __syncthreads();
    uint64_t* cand = reinterpret_cast<uint64_t*>(sT);           // [128 queries][4 sources][2]
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int ql = wc * 64 + n * 32 + (lane & 31);
        const int srcslot = wr * 2 + (lane >> 5);
        cand[(ql * 4 + srcslot) * 2 + 0] = l2_key(best[n].d0, best[n].i0);
        cand[(ql * 4 + srcslot) * 2 + 1] = l2_key(best[n].d1, best[n].i1);
    }
    __syncthreads();
    if (tid < L2_TN && q0 + tid < nq) {
        uint64_t b0 = L2_EMPTY, b1 = L2_EMPTY;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t key = cand[tid * 8 + j];
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
        hipLaunchKernelGGL((l2_topk_kernel<KK, FF, PP>), dim3(q_tiles, splits), dim3(256), lds, s, t, (const float*)tn, (int)nt, q, (const float*)qn, nq, \
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
