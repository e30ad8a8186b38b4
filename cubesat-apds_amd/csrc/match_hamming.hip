// csrc/match_hamming.hip — Hamming brute-force top-k on gfx950 (integer VALU + scalar broadcast).
//
// Replaces BFMatcher(NORM_HAMMING).knnMatch / match(crossCheck) behind
// /root/reference/feature_extraction/src/lib.rs:94-126.
//
// Mapping (CDNA4): one LANE owns T query descriptors (16 dwords each, in VGPRs for the whole kernel);
// one WAVE walks a contiguous chunk of train rows. A train row is wave-uniform, so it is fetched with
// scalar loads (s_load_dwordx16: one 64-byte line per row) and fed to the VALU as an SGPR operand:
//     v_xor_b32  t, s_row[j], v_q[j]      v_bcnt_u32_b32  acc, t, acc
// = 30 VALU lane-ops per (query,row) pair on the fast path (15 dwords; the 16th holds 6 descriptor bits and is only
// needed to finish a candidate hit; the algorithmic figure used for the roofline stays 32), no LDS and no cross-lane traffic on the hot path. The running
// top-k lives per lane; the popcount chain starts at -threshold so "distance < current k-th best" is the
// sign bit, a whole group of pairs is screened with one v_min3/v_cmp, and the insertion code runs only for
// the rare groups that contain a hit. Train rows are split into chunks over blockIdx.y so the grid fills
// 256 CUs; per-chunk candidates are merged by a second tiny kernel. Keys are (distance << 32 | index):
// unsigned 64-bit min reproduces BFMatcher's order (distance, then lower train index).
#include <atomic>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "config.h"
#include "kernels.h"

namespace apds {

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr uint64_t EMPTY_KEY = ~0ull;
static constexpr int INF_THR = 1 << 20;          // > any Hamming distance of a 512-bit row
static constexpr uint32_t NO_INDEX = 0xFFFFFFFFu;
// Per-chunk candidate lists. A work item (one wave: 64*T queries x one row chunk) leaves one RECORD: T*K 64-bit presence masks
// (bit = lane; slot s = t*K + k) followed by the present keys only, slot-major, lanes ascending. After the threshold pre-pass
// ~5 of 6 slots are empty (the chunk held nothing better than the query's starting threshold), so records are mostly header:
// the lists cost ~0.8 bytes per slot instead of 4 (and instead of 8 as 64-bit keys). Keys are 32 bits: distance (<= 512,
// 10 bits) << 22 | row offset inside the chunk (chunks hold at most 2^22 rows); same order as the 64-bit
// (distance << 32 | global row) keys they expand to in the merge. Records sit at a fixed pitch (capacity for all slots
// present); untouched bytes cost no traffic.
static constexpr int PART_ROW_BITS = 22;
template <int T, int K>
struct PartRecord {
    static constexpr int SLOTS = T * K;
    static constexpr int HEADER = 2 * SLOTS;             // u32 words of masks
    static constexpr int PITCH = HEADER + 64 * SLOTS;    // u32 words per record
};

// acc + popcount(x) in ONE VALU op. hipcc otherwise splits the accumulate into v_bcnt(x,0) + v_add3 (5 ops per
// two dwords instead of 4), so the accumulate form is spelled out.
__device__ __forceinline__ int bcnt_acc(uint32_t x, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// Lower bound of the distances of one train row to the lane's T queries: popcount over the first 15 dwords (480 bits)
// on top of start[t] = -threshold. M-LDB uses 486 bits, so dword 15 holds at most 6 set bits of the XOR: if the bound
// is already >= threshold the pair cannot be a hit, and dword 15 is only looked at in the rare hit path.
// 30 VALU ops per pair instead of 32.
//
// Issue order (measured, profiles/r02/valu_calib.log): on a gfx950 SIMD a VOP2 v_xor_b32 takes 2 issue cycles and the VOP3
// v_bcnt_u32_b32 takes 4, but one wave issues at most one VALU op per 4-cycle window, so an xor only costs 2 when the xor of
// ANOTHER wave shares its window. With the waves of a SIMD running the plain stream xor, bcnt, xor, bcnt, ... in step that
// pairing rarely happens (3.83 cycles per op, 38.5 T lane-ops/s chip-wide); an `s_nop 0` between each xor and the bcnt that
// consumes it takes the wave out of step with its neighbours and the pairs settle at 2 + 4 cycles (2.9 per op, 50.7 T lane-ops/s,
// 99 % of the 6-cycle pair). A nop after every op, or the xor issued one or two steps ahead, does not: only this placement.
#ifndef APDS_PAIR_NOP
#define APDS_PAIR_NOP 1
#endif
template <int T>
__device__ __forceinline__ void row_distances(const u32x16 row, const uint32_t (&q)[T][16], const int (&start)[T], int (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; t++) {
        int a = start[t];
#pragma unroll
        for (int j = 0; j < 15; j++) {
#if APDS_PAIR_NOP
            const uint32_t x = q[t][j] ^ row[j];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 0");
            __builtin_amdgcn_sched_barrier(0);
            a = bcnt_acc(x, a);
            __builtin_amdgcn_sched_barrier(0);
#else
            a = bcnt_acc(q[t][j] ^ row[j], a);
#endif
        }
        acc[t] = a;
    }
}

// Paged scans (k > 16, see hamming_topk_device) look for the best K keys ABOVE a per-query floor key (the last key of the page before):
// FLOOR adds that one comparison to the rare hit path - the rows at or below the floor are the query's best, so they all reach it, but
// there are only 16 per page of them.
struct KeyFloor {
    uint32_t dist, index;   // the floor key (distance << 32 | global row index), split
    uint32_t base;          // global index of row 0 of the scanned array
};

template <int T, int K, bool FLOOR = false>
__device__ __forceinline__ void insert_hits(const int (&acc)[T], const int (&nthr_old)[T], const uint32_t (&q)[T][16], uint32_t row15, uint32_t r,
                                            int (&bd)[T][K], uint32_t (&bi)[T][K], const KeyFloor (&fl)[T]) {
#pragma unroll
    for (int t = 0; t < T; t++) {
        if (acc[t] < 0) {   // the 480-bit bound is below the threshold this pair was screened with: finish the distance
            const int d = acc[t] - nthr_old[t] + __popc(q[t][15] ^ row15);
            if (FLOOR && !((uint32_t)d > fl[t].dist || ((uint32_t)d == fl[t].dist && r + fl[t].base > fl[t].index))) continue;
            if (d < bd[t][K - 1]) {
                // insertion with strict '<': a later row never moves ahead of an equal earlier one
                bool placed = false;
#pragma unroll
                for (int j = K - 1; j > 0; j--) {
                    if (!placed) {
                        if (bd[t][j - 1] > d) {
                            bd[t][j] = bd[t][j - 1];
                            bi[t][j] = bi[t][j - 1];
                        } else {
                            bd[t][j] = d;
                            bi[t][j] = r;
                            placed = true;
                        }
                    }
                }
                if (!placed) {
                    bd[t][0] = d;
                    bi[t][0] = r;
                }
            }
        }
    }
}

// One work item = (64*T queries per wave, 4 waves) x (one chunk of train rows).
template <int T, int K, bool FLOOR = false>
__device__ __forceinline__ void hamming_topk_item(const u32x16* __restrict__ train, int n_train, const u32x4* __restrict__ queries, int nq,
                                                  int rows_per_chunk, const int* __restrict__ init_thr, uint32_t* __restrict__ out, int chunk,
                                                  int qblock, const uint64_t* __restrict__ floor_keys = nullptr, uint32_t floor_base = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int qbase = (qblock * 4 + wave) * (64 * T);
    if (qbase >= nq) return;   // wave-uniform
    const int row0 = chunk * rows_per_chunk;
    const int row1 = min(n_train, row0 + rows_per_chunk);
    uint32_t q[T][16];
    int bd[T][K];
    uint32_t bi[T][K];
    int nthr[T];
    KeyFloor fl[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int qi = min(qbase + t * 64 + lane, nq - 1);
        if (FLOOR) {   // the last key of the query's previous page (all ones when that page was not full: nothing is above it)
            const uint64_t f = floor_keys[(size_t)qi * K + (K - 1)];
            fl[t].dist = (uint32_t)(f >> 32);
            fl[t].index = (uint32_t)f;
            fl[t].base = floor_base;
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const u32x4 x = queries[(size_t)qi * 4 + v];
            q[t][4 * v + 0] = x.x;
            q[t][4 * v + 1] = x.y;
            q[t][4 * v + 2] = x.z;
            q[t][4 * v + 3] = x.w;
        }
        const int thr0 = init_thr ? init_thr[qi] : INF_THR;
#pragma unroll
        for (int k = 0; k < K; k++) {
            bd[t][k] = thr0;
            bi[t][k] = NO_INDEX;
        }
        nthr[t] = -thr0;
    }

    // Screen two rows against the lane's T queries; run the insertion code only if some lane has a hit.
    auto pair_step = [&](const u32x16& a0, const u32x16& a1, int r) {
        int acc0[T], acc1[T];
        row_distances<T>(a0, q, nthr, acc0);
        row_distances<T>(a1, q, nthr, acc1);
        int m = acc0[0];
#pragma unroll
        for (int t = 1; t < T; t++) m = min(m, acc0[t]);
#pragma unroll
        for (int t = 0; t < T; t++) m = min(m, acc1[t]);
        if (__any(m < 0)) {
            int nthr_old[T];
#pragma unroll
            for (int t = 0; t < T; t++) nthr_old[t] = nthr[t];
            insert_hits<T, K, FLOOR>(acc0, nthr_old, q, a0[15], (uint32_t)r, bd, bi, fl);
            insert_hits<T, K, FLOOR>(acc1, nthr_old, q, a1[15], (uint32_t)(r + 1), bd, bi, fl);
#pragma unroll
            for (int t = 0; t < T; t++) nthr[t] = -bd[t][K - 1];
        }
    };

    int r = row0;
    // Main loop: four rows per trip as two ping-pong pairs (A, B). The scalar loads of one pair are issued
    // before the 2*T*32 VALU ops of the other pair, so their latency sits under compute; SMEM returns out
    // of order, hence the only wait is lgkmcnt(0) and it always lands after a full compute phase.
    if (r + 3 < row1) {
        u32x16 a0 = train[r], a1 = train[r + 1];
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        for (; r + 3 < row1; r += 4) {
            const u32x16 b0 = train[r + 2], b1 = train[r + 3];
            __builtin_amdgcn_sched_barrier(0);
            pair_step(a0, a1, r);
            __builtin_amdgcn_s_waitcnt(0xC07F);   // pair B has landed (issued one compute phase ago)
            const int rn = min(r + 4, row1 - 1), rm = min(r + 5, row1 - 1);
            a0 = train[rn];
            a1 = train[rm];
            __builtin_amdgcn_sched_barrier(0);
            pair_step(b0, b1, r + 2);
            __builtin_amdgcn_s_waitcnt(0xC07F);   // pair A has landed
        }
    }
    for (; r < row1; r++) {   // tail: at most three rows
        const u32x16 a0 = train[r];
        int acc0[T];
        row_distances<T>(a0, q, nthr, acc0);
        insert_hits<T, K, FLOOR>(acc0, nthr, q, a0[15], (uint32_t)r, bd, bi, fl);
#pragma unroll
        for (int t = 0; t < T; t++) nthr[t] = -bd[t][K - 1];
    }

    // one record per (chunk, wave tile): presence masks + the present keys, compacted with ballot / popcount prefix sums
    using Rec = PartRecord<T, K>;
    const int n_wtiles = (nq + 64 * T - 1) / (64 * T);
    uint32_t* rec = out + ((size_t)chunk * n_wtiles + (qblock * 4 + wave)) * Rec::PITCH;
    const uint64_t lt = (1ull << lane) - 1;
    int base = 0;
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int qi = qbase + t * 64 + lane;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool present = qi < nq && bi[t][k] != NO_INDEX;
            const uint64_t m = __ballot(present);
            if (present) rec[Rec::HEADER + base + __popcll(m & lt)] = ((uint32_t)bd[t][k] << PART_ROW_BITS) | (bi[t][k] - (uint32_t)row0);
            if (lane == 0) {
                rec[2 * (t * K + k)] = (uint32_t)m;
                rec[2 * (t * K + k) + 1] = (uint32_t)(m >> 32);
            }
            base += __popcll(m);
        }
    }
}

// grid: 1-D, 8 * ceil(chunks/8) * ceil(nq / (256*T)) blocks. block = 256 threads = 4 waves, each wave its own 64*T queries.
template <int T, int K>
__global__ __launch_bounds__(256) void hamming_topk_kernel(const u32x16* __restrict__ train, int n_train,
                                                           const u32x4* __restrict__ queries, int nq, int rows_per_chunk,
                                                           const int* __restrict__ init_thr, uint32_t* __restrict__ out, int qtile_blocks,
                                                           int n_chunks, int xcd_aware) {
    // 1-D grid, XCD-aware by default: workgroups are dealt round-robin over the 8 XCDs, so all query tiles of one row
    // chunk are given the same (id % 8): the chunk is then streamed into ONE XCD's L2 instead of all eight (13x less
    // L2->fabric traffic at equal speed, once the chunk count is a multiple of 8 so that no XCD gets extra chunks).
    // APDS_MATCH_XCD=0 restores the plain (chunk-major) order. Placement only affects speed, never results.
    const int nqb = qtile_blocks;
    int chunk, qblock;
    if (xcd_aware) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        chunk = (slot / nqb) * 8 + xcd;
        qblock = slot - (slot / nqb) * nqb;
    } else {
        chunk = blockIdx.x / nqb;
        qblock = blockIdx.x - chunk * nqb;
    }
    if (chunk >= n_chunks) return;   // block-uniform
    hamming_topk_item<T, K>(train, n_train, queries, nq, rows_per_chunk, init_thr, out, chunk, qblock);
}

// One page of a paged scan: the K best keys above floor_keys[query][K - 1] (one query per lane, plain chunk-major grid).
template <int K>
__global__ __launch_bounds__(256) void hamming_topk_page_kernel(const u32x16* __restrict__ train, int n_train, const u32x4* __restrict__ queries, int nq,
                                                                int rows_per_chunk, const int* __restrict__ init_thr, uint32_t* __restrict__ out,
                                                                int qtile_blocks, int n_chunks, const uint64_t* __restrict__ floor_keys,
                                                                uint32_t floor_base) {
    const int chunk = blockIdx.x / qtile_blocks, qblock = blockIdx.x - chunk * qtile_blocks;
    if (chunk >= n_chunks) return;   // block-uniform
    hamming_topk_item<1, K, true>(train, n_train, queries, nq, rows_per_chunk, init_thr, out, chunk, qblock, floor_keys, floor_base);
}

// (A persistent work-queue form of this kernel - resident workgroups pulling items from an atomic counter - was kept through round 2 behind
// APDS_MATCH_PERSIST; the plain grid was as fast in every sweep (profiles/r01/match_persistent_sweep.log): deleted in round 3.)

// The same for any k (the sharded matcher above 16 neighbours, and k not a power of two): every list is ascending and all keys are distinct
// (a key carries its global row), so output j is the smallest key above output j - 1: each lane walks a cursor per list. parts <= 64.
__global__ void merge_topk_any_kernel(const uint64_t* __restrict__ parts_keys, int parts, int nq, int k, uint64_t* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    uint16_t cur[64];   // cursor of every list
    for (int p = 0; p < parts; p++) cur[p] = 0;
    for (int j = 0; j < k; j++) {
        uint64_t best = EMPTY_KEY;
        int arg = -1;
        for (int p = 0; p < parts; p++) {
            if (cur[p] >= k) continue;
            const uint64_t v = parts_keys[((size_t)p * nq + qi) * k + cur[p]];
            if (v < best) best = v, arg = p;
        }
        out[(size_t)qi * k + j] = best;
        if (arg >= 0) cur[arg]++;
    }
}

// merge `parts` sorted candidate lists per query into the k smallest keys
template <int K>
__global__ void merge_topk_kernel(const uint64_t* __restrict__ parts_keys, int parts, int nq, uint64_t* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    uint64_t best[K];
#pragma unroll
    for (int k = 0; k < K; k++) best[k] = EMPTY_KEY;
    auto push = [&](uint64_t key) {
        if (key < best[K - 1]) {
            bool placed = false;
#pragma unroll
            for (int j = K - 1; j > 0; j--) {
                if (!placed) {
                    if (best[j - 1] > key) best[j] = best[j - 1];
                    else {
                        best[j] = key;
                        placed = true;
                    }
                }
            }
            if (!placed) best[0] = key;
        }
    };
    constexpr int U = K <= 2 ? 8 : 2;   // independent loads in flight per lane
    int p = 0;
    for (; p + U <= parts; p += U) {
        uint64_t v[U][K];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < K; k++) v[u][k] = parts_keys[((size_t)(p + u) * nq + qi) * K + k];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < K; k++) push(v[u][k]);
    }
    for (; p < parts; p++)
#pragma unroll
        for (int k = 0; k < K; k++) push(parts_keys[((size_t)p * nq + qi) * K + k]);
#pragma unroll
    for (int k = 0; k < K; k++) out[(size_t)qi * K + k] = best[k];
}

// Merge of the per-chunk records of hamming_topk_kernel. One BLOCK per query tile (lane = the T queries it owned in the match
// kernel): its four waves take every fourth chunk each, four chunks per trip with all of a trip's loads issued before the first
// insertion (a wave walking all chunks one by one was a chain of 360 dependent memory round trips: 0.23 ms for 138 waves on an
// otherwise idle GPU), then wave 0 folds the other waves' lists into its own through LDS. A present key expands to
// (distance << 32 | row offset + p * rows_per_chunk + row_base); keys are unique, so the K smallest do not depend on the order of
// insertion. `extra` (nq x K 64-bit keys, may be null) is one more already-expanded sorted list (the sample pass).
template <int K>
__device__ __forceinline__ void topk_insert(uint64_t (&best)[K], uint64_t key) {
    if (key < best[K - 1]) {
        bool placed = false;
#pragma unroll
        for (int j = K - 1; j > 0; j--) {
            if (!placed) {
                if (best[j - 1] > key) best[j] = best[j - 1];
                else {
                    best[j] = key;
                    placed = true;
                }
            }
        }
        if (!placed) best[0] = key;
    }
}

template <int T, int K>
__global__ __launch_bounds__(256) void merge_records_kernel(const uint32_t* __restrict__ recs, int parts, int rows_per_chunk, uint32_t row_base,
                                                            const uint64_t* __restrict__ extra, int nq, uint64_t* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    using Rec = PartRecord<T, K>;
    constexpr int G = T * K <= 8 ? 4 : 1;       // chunks per trip (the wide records of large k: one)
    __shared__ uint64_t s_best[3][T * K][64];   // lists of waves 1..3
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_wtiles = (nq + 64 * T - 1) / (64 * T);
    const int wtile = blockIdx.x;
    uint64_t best[T][K];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int qi = wtile * 64 * T + t * 64 + lane;
#pragma unroll
        for (int k = 0; k < K; k++) best[t][k] = (wave == 0 && extra && qi < nq) ? extra[(size_t)qi * K + k] : EMPTY_KEY;   // a sorted list: a valid start
    }
    const uint64_t lt = (1ull << lane) - 1;
    for (int p0 = wave; p0 < parts; p0 += 4 * G) {
        uint32_t keys[G][Rec::SLOTS];
#pragma unroll
        for (int g = 0; g < G; g++) {           // all loads of the trip first
            const int p = p0 + 4 * g;
            const bool live = p < parts;        // wave-uniform
            const uint32_t* rec = recs + ((size_t)(live ? p : p0) * n_wtiles + wtile) * Rec::PITCH;
            int base = 0;
#pragma unroll
            for (int s = 0; s < Rec::SLOTS; s++) {
                const uint64_t m = live ? ((uint64_t)rec[2 * s] | ((uint64_t)rec[2 * s + 1] << 32)) : 0ull;   // wave-uniform
                const bool present = (m >> lane) & 1;
                keys[g][s] = present ? rec[Rec::HEADER + base + __popcll(m & lt)] : 0xFFFFFFFFu;
                base += __popcll(m);
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t chunk_base = (uint32_t)(p0 + 4 * g) * (uint32_t)rows_per_chunk + row_base;
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const uint32_t part = keys[g][t * K + k];
                    if (part == 0xFFFFFFFFu) continue;   // distance 1023 cannot occur: "absent"
                    topk_insert<K>(best[t], ((uint64_t)(part >> PART_ROW_BITS) << 32) |
                                                (uint64_t)(uint32_t)((part & ((1u << PART_ROW_BITS) - 1)) + chunk_base));
                }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int k = 0; k < K; k++) s_best[wave - 1][t * K + k][lane] = best[t][k];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; w++)
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint64_t key = s_best[w][t * K + k][lane];
                if (key != EMPTY_KEY) topk_insert<K>(best[t], key);
            }
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int qi = wtile * 64 * T + t * 64 + lane;
        if (qi < nq)
#pragma unroll
            for (int k = 0; k < K; k++) out[(size_t)qi * K + k] = best[t][k];
    }
}

__global__ void take_first_columns_kernel(const uint64_t* __restrict__ in, int nq, int kin, int kout, uint64_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nq * kout) return;
    const int q = (int)(i / kout), c = (int)(i - (long long)q * kout);
    out[i] = in[(size_t)q * kin + c];
}

// second-best distance of a sample of train rows -> initial thresholds for the full scan
__global__ void thr_from_keys_kernel(const uint64_t* __restrict__ keys, int nq, int K, int* __restrict__ thr) {
    APDS_RAISE_WAVE_PRIORITY();
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const uint64_t key = keys[(size_t)qi * K + (K - 1)];
    thr[qi] = key == EMPTY_KEY ? INF_THR : (int)(key >> 32);
}

__global__ void pack_rows_kernel(const uint8_t* __restrict__ src, long long n, int desc_bytes, long long src_stride,
                                 uint32_t* __restrict__ dst) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one dword of one row
    if (i >= n * 16) return;
    const long long row = i >> 4;
    const int w = (int)(i & 15);
    const uint8_t* p = src + row * src_stride + w * 4;
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++)
        if (w * 4 + b < desc_bytes) v |= (uint32_t)p[b] << (8 * b);
    dst[i] = v;
}

__global__ void ratio_flag_kernel(const uint64_t* __restrict__ keys, int nq, int K, float fs, uint8_t* __restrict__ flags) {
    APDS_RAISE_WAVE_PRIORITY();
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const uint64_t k0 = keys[(size_t)qi * K], k1 = keys[(size_t)qi * K + 1];
    const float d0 = (float)(uint32_t)(k0 >> 32), d1 = (float)(uint32_t)(k1 >> 32);
    flags[qi] = (k0 != EMPTY_KEY && k1 != EMPTY_KEY && d0 < d1 * fs) ? 1 : 0;
}

__global__ void crosscheck_scatter_kernel(const uint64_t* __restrict__ train_best, long long n_train, unsigned long long* __restrict__ best_per_query) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_train) return;
    const uint64_t key = train_best[i];
    if (key == EMPTY_KEY) return;
    const uint32_t qidx = (uint32_t)key;
    const uint64_t cand = (key & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)i;
    atomicMin(&best_per_query[qidx], (unsigned long long)cand);
}

__global__ void nonempty_flag_kernel(const uint64_t* __restrict__ keys, int n, uint8_t* __restrict__ flags) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = keys[i] != EMPTY_KEY;
}

// ---- ordered compaction: flags -> exclusive positions (3 small kernels, no host round trip) -------------
static constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(SCAN_BLOCK) void scan_block_counts_kernel(const uint8_t* __restrict__ flags, int n, int* __restrict__ block_counts) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int f = i < n ? (flags[i] != 0) : 0;
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; w++) s += wsum[w];
        block_counts[blockIdx.x] = s;
    }
}

// single block: exclusive scan of block_counts in place, total to *total
__global__ __launch_bounds__(1024) void scan_offsets_kernel(int* __restrict__ block_counts, int nblocks, int* __restrict__ total) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        const int incl = buf[threadIdx.x];
        if (i < nblocks) block_counts[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__device__ __forceinline__ int block_exclusive_pos(int f, int block_offset) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int k = 0; k < w; k++) before += wsum[k];
    return block_offset + before + __popcll(b & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(SCAN_BLOCK) void emit_ratio_matches_kernel(const uint64_t* __restrict__ keys, int nq, int K, const uint8_t* __restrict__ flags,
                                                                        const int* __restrict__ block_offsets, apds_dmatch* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int f = i < nq ? (flags[i] != 0) : 0;
    const int pos = block_exclusive_pos(f, block_offsets[blockIdx.x]);
    if (f) {
        const uint64_t k0 = keys[(size_t)i * K];
        apds_dmatch m;
        m.query_idx = i;
        m.train_idx = (int32_t)(uint32_t)k0;
        m.img_idx = 0;
        m.distance = (float)(uint32_t)(k0 >> 32);
        out[pos] = m;
    }
}

// ---- register-only VALU microbenchmark (denominator of the popcount roofline) ---------------------------
// One instruction kind per MODE, eight independent chains per lane, nothing but that instruction in the loop body (the loop
// counter is scalar). Modes 0-3 and 8-10 are 32-bit integer ops, 4-7 and 11 are FP32 ops issued by the same waves on the same
// SIMDs, so one run shows whether the integer ops the match kernel is made of issue at the FP32 rate or at half of it.
//   0 v_xor_b32(sgpr, vgpr) + v_bcnt_u32_b32 accumulate: the match kernel's inner pair (two lane-ops per pair)
//   1 v_xor_b32 (sgpr operand)   2 v_bcnt_u32_b32 accumulate   3 v_add_u32 (sgpr operand)
//   4 v_fma_f32   5 v_add_f32   6 v_pk_fma_f32 (two FMAs per lane per instruction)   7 v_pk_add_f32 (two adds)
//   8 v_xor_b32 (vgpr, vgpr)   9 v_bfi_b32 (VOP3, three vgprs)   10 v_and_b32 (vgpr, vgpr)   11 v_mul_f32
// Modes 12-17 replay the ISSUE PATTERN of the match kernel's inner loop (one train row in 15 SGPRs against 4 queries of 15 dwords
// in VGPRs = 60 xor + 60 bcnt per row) in different instruction orders, to find the order the SIMD issues fastest:
//   12 query-sequential, one dependent chain per query (xor t,s_j,q_cj ; bcnt a_c,t,a_c for j = 0..14, then the next query): the
//      order hipcc emits for row_distances()        13 dword-major (the four queries' chains interleaved round-robin)
//   14 = 13 with the xor of step i+1 issued before the bcnt of step i (two temporaries)      15 = 13 with the row first copied to
//   VGPRs (15 v_mov per row, not counted) so the xor has no SGPR operand      16 = 12 skewed like 14      17 = 13 with two xors
//   ahead (three temporaries)
// Every wave also leaves its s_memtime span, so the host can state cycles per wave-instruction per SIMD without assuming a clock.
typedef float f32x2 __attribute__((ext_vector_type(2)));
static constexpr int VALU_MODES = 28;
static constexpr int VALU_CHAINS = 8, VALU_UNROLL = 16;

template <int MODE>
__global__ __launch_bounds__(256) void valu_peak_kernel(uint32_t* __restrict__ sink, unsigned long long* __restrict__ spans, int iters,
                                                        const u32x16* __restrict__ rowp) {
    extern __shared__ uint32_t occupancy_pad[];   // dynamic LDS request only bounds the workgroups per CU
    if (MODE >= 12) {
        uint32_t q4[4][15];
        int a[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            a[c] = c;
#pragma unroll
            for (int j = 0; j < 15; j++) q4[c][j] = threadIdx.x * 2654435761u + (c * 16 + j) * 40503u + 1u;
        }
        // the row is reloaded every iteration (wave-uniform address: s_load_dwordx16, as in the match kernel), one row ahead; the
        // xor is plain C and the accumulate is bcnt_acc(), as in row_distances(); sched_barrier(0) pins the order under test
        // (two adjacent inline-asm VALU ops make hipcc insert an s_nop between them, which the match kernel's loop does not have)
        u32x16 row = rowp[blockIdx.x & 7];
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
            const u32x16 next = rowp[(blockIdx.x + it + 1) & 7];
#define APDS_X(tmp, j, c) { tmp = row[j] ^ q4[c][j]; __builtin_amdgcn_sched_barrier(0); }
#define APDS_XV(tmp, j, c) { tmp = rv[j] ^ q4[c][j]; __builtin_amdgcn_sched_barrier(0); }
#define APDS_B(tmp, c) { a[c] = bcnt_acc(tmp, a[c]); __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 12) {
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) { uint32_t t; APDS_X(t, j, c); APDS_B(t, c); }
            } else if (MODE == 18) {      // 12 with an s_nop between every xor and the bcnt that consumes it
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) { uint32_t t; APDS_X(t, j, c); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); APDS_B(t, c); }
            } else if (MODE == 19) {      // 13 with the s_nop
#pragma unroll
                for (int j = 0; j < 15; j++)
#pragma unroll
                    for (int c = 0; c < 4; c++) { uint32_t t; APDS_X(t, j, c); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); APDS_B(t, c); }
            } else if (MODE == 20) {      // 12 with an s_nop after EVERY instruction
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) {
                        uint32_t t;
                        APDS_X(t, j, c); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0);
                        APDS_B(t, c); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0);
                    }
            } else if (MODE == 22) {      // bcnt only (the xor hoisted: 15 xors, then 60 bcnt with an s_nop after each) - is bcnt 4 cycles whatever the phase?
                uint32_t t[15];
#pragma unroll
                for (int j = 0; j < 15; j++) APDS_X(t[j], j, 0);
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) { APDS_B(t[j], c); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); }
            } else if (MODE == 23) {      // X X nop B B
#pragma unroll
                for (int c = 0; c < 4; c += 2)
#pragma unroll
                    for (int j = 0; j < 15; j++) {
                        uint32_t t0, t1;
                        APDS_X(t0, j, c); APDS_X(t1, j, c + 1); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0);
                        APDS_B(t0, c); APDS_B(t1, c + 1);
                    }
            } else if (MODE == 24) {      // X s_nop 1 B
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) { uint32_t t; APDS_X(t, j, c); asm volatile("s_nop 1"); __builtin_amdgcn_sched_barrier(0); APDS_B(t, c); }
            } else if (MODE == 25) {      // B nop X (the nop after the bcnt instead of before it): X0, then [B nop X] ...
                uint32_t t[60];
                APDS_X(t[0], 0, 0);
#pragma unroll
                for (int i = 0; i < 60; i++) {
                    APDS_B(t[i], i / 15);
                    asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0);
                    if (i + 1 < 60) APDS_X(t[i + 1], (i + 1) % 15, (i + 1) / 15);
                }
            } else if (MODE == 26) {      // X nop B where the nop is an s_sleep-free scalar ALU op (s_add on a dummy) instead of s_nop
                int dummy = it;
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) {
                        uint32_t t;
                        APDS_X(t, j, c);
                        asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy));
                        __builtin_amdgcn_sched_barrier(0);
                        APDS_B(t, c);
                    }
                if (dummy == 0x7ffffff0) a[0]++;
            } else if (MODE == 27) {      // X nop B with a second independent pair stream interleaved: X0 X1 nop B0 B1 on two queries at a time, dword-major
#pragma unroll
                for (int j = 0; j < 15; j++) {
                    uint32_t t0, t1, t2, t3;
                    APDS_X(t0, j, 0); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); APDS_B(t0, 0);
                    APDS_X(t1, j, 1); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); APDS_B(t1, 1);
                    APDS_X(t2, j, 2); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); APDS_B(t2, 2);
                    APDS_X(t3, j, 3); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); APDS_B(t3, 3);
                }
            } else if (MODE == 21) {      // 12 with ONE SGPR for the whole row (row[0]) instead of fifteen
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 15; j++) { uint32_t t; t = row[0] ^ q4[c][j]; __builtin_amdgcn_sched_barrier(0); APDS_B(t, c); }
            } else if (MODE == 13) {
#pragma unroll
                for (int j = 0; j < 15; j++)
#pragma unroll
                    for (int c = 0; c < 4; c++) { uint32_t t; APDS_X(t, j, c); APDS_B(t, c); }
            } else if (MODE == 14 || MODE == 17) {
                constexpr int AHEAD = MODE == 17 ? 2 : 1;
                uint32_t t[60];
#pragma unroll
                for (int i = 0; i < 60 + AHEAD; i++) {
                    if (i < 60) APDS_X(t[i], i >> 2, i & 3);
                    if (i >= AHEAD) APDS_B(t[i - AHEAD], (i - AHEAD) & 3);
                }
            } else if (MODE == 15) {
                uint32_t rv[15];
#pragma unroll
                for (int j = 0; j < 15; j++) { asm volatile("v_mov_b32 %0, %1" : "=v"(rv[j]) : "s"(row[j])); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 15; j++)
#pragma unroll
                    for (int c = 0; c < 4; c++) { uint32_t t; APDS_XV(t, j, c); APDS_B(t, c); }
            } else {   // 16: query-sequential, skewed by one
                uint32_t t[60];
#pragma unroll
                for (int i = 0; i < 61; i++) {
                    if (i < 60) APDS_X(t[i], i % 15, i / 15);
                    if (i >= 1) APDS_B(t[i - 1], (i - 1) / 15);
                }
            }
#undef APDS_X
#undef APDS_XV
#undef APDS_B
            row = next;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if ((a[0] + a[1] + a[2] + a[3]) == 0x7fffffff) sink[0] = 1;
        if ((threadIdx.x & 63) == 0) spans[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
        return;
    }
    uint32_t q[VALU_UNROLL];
#pragma unroll
    for (int j = 0; j < VALU_UNROLL; j++) q[j] = threadIdx.x * 2654435761u + j * 40503u + 1u;
    uint32_t acc[VALU_CHAINS];
    float facc[VALU_CHAINS];
    f32x2 pacc[VALU_CHAINS];
#pragma unroll
    for (int c = 0; c < VALU_CHAINS; c++) {
        acc[c] = c + threadIdx.x;
        facc[c] = 1.0f + 0.001f * (float)(c + (threadIdx.x & 7));
        pacc[c] = f32x2{facc[c], facc[c] * 0.5f};
    }
    const float fm = 0.99999f, fa = 1e-6f;
    const f32x2 pm = {0.99999f, 0.99998f}, pa = {1e-6f, 2e-6f};
    uint32_t s = blockIdx.x * 97u + 1u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < VALU_UNROLL; j++) {
#pragma unroll
            for (int c = 0; c < VALU_CHAINS; c++) {
                if (MODE == 0) {
                    uint32_t x;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(q[j]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[c]) : "v"(x));
                } else if (MODE == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[c]) : "s"(s));
                else if (MODE == 2) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[c]) : "v"(q[j]));
                else if (MODE == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[c]) : "s"(s));
                else if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(facc[c]) : "v"(fm), "v"(fa));
                else if (MODE == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(facc[c]) : "v"(fa));
                else if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pacc[c]) : "v"(pm), "v"(pa));
                else if (MODE == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pacc[c]) : "v"(pa));
                else if (MODE == 8) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[c]) : "v"(q[j]));
                else if (MODE == 9) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(q[j]), "v"(q[(j + 1) & (VALU_UNROLL - 1)]));
                else if (MODE == 10) asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[c]) : "v"(q[j]));
                else asm volatile("v_mul_f32 %0, %1, %0" : "+v"(facc[c]) : "v"(fm));
            }
        }
        s = s * 1664525u + 1013904223u;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t fold = 0;
#pragma unroll
    for (int c = 0; c < VALU_CHAINS; c++) fold += acc[c] + __float_as_uint(facc[c]) + __float_as_uint(pacc[c].x) + __float_as_uint(pacc[c].y);
    if (fold == 0x7fffffffu) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) spans[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- host launchers -------------------------------------------------------------------------------------
struct ChunkPlan {
    int T, chunks, rows_per_chunk, qtiles_blocks;
};


// Occupancy cap of the main scan (bytes of unused dynamic LDS per workgroup; 0 = none), process-wide: APDS_MATCH_LDS_CAP at first
// use, then apds_dev_match_lds_cap(). See launch_topk.
// dynamic-LDS bytes the most recent scan launch of the process was made with (test hook: the cap reaches every kernel variant)
std::atomic<int>& last_scan_launch_lds() {
    static std::atomic<int> v{-1};
    return v;
}

std::atomic<int>& match_lds_cap() {
    static std::atomic<int> cap{config().match_lds_cap};
    return cap;
}
// The same cap for the scans launched by ONE host thread (the streamed pipeline's match worker, when its starvation watch decides to
// cap): -1 = the process-wide value applies.
static thread_local int tl_scan_cap = -1;
void set_thread_scan_cap(int bytes) { tl_scan_cap = bytes; }

// Work items are (64*T queries) x (rows_per_chunk train rows) per wave. Items are kept small enough that the
// grid is many dispatch rounds deep (the block scheduler then balances the tail), but not so small that the
// per-chunk candidate lists dominate the merge.
static ChunkPlan plan_chunks(int nq, long long n_train, bool sample_pass = false, bool one_query_per_lane = false) {
    ChunkPlan p;
    // (the sweeps behind these constants: profiles/r01/match_work_item_sweep.log, match_probe.log; they were environment knobs until round 3)
    constexpr int target_waves = 256 * 4 * 4 * 24;   // waves per launch: many dispatch rounds deep, so the block scheduler balances the tail
    const int min_rows = sample_pass ? 256 : 1024;   // shortest row chunk of a work item
    p.T = nq >= 64 * 4 * 64 ? 4 : (nq >= 64 * 2 * 64 ? 2 : 1);
    // the threshold pre-pass covers few rows: smaller items (T = 1, short chunks) keep all CUs busy
    // (round 3, with the pre-pass running beside the previous frame's main scan: T = 2 / 4 for it measured again, no difference)
    if (sample_pass) p.T = 1;
    if (one_query_per_lane) p.T = 1;
    const int waves_q = ceil_div(nq, 64 * p.T);
    p.qtiles_blocks = ceil_div(waves_q, 4);
    long long chunks = ceil_div(target_waves, waves_q);
    const long long max_chunks = std::max<long long>(1, n_train / min_rows);
    chunks = std::min<long long>(std::max<long long>(chunks, 1), std::min<long long>(max_chunks, 65535));
    // chunk c runs on XCD (c % 8) when the XCD-aware placement is on: keep the chunk count a multiple of 8 so that every
    // XCD gets the same number of rows (3 extra chunks on 3 XCDs was a 2 % tail)
    chunks = std::max<long long>(chunks, ceil_div(n_train, 1ll << PART_ROW_BITS));   // 22-bit row offsets in the per-chunk keys
    if (chunks >= 8) chunks = (chunks + 7) & ~7ll;
    long long rpc = (n_train + chunks - 1) / chunks;
    rpc = (rpc + 3) & ~3ll;
    p.rows_per_chunk = (int)rpc;
    p.chunks = (int)((n_train + rpc - 1) / rpc);
    return p;
}

template <int K>
static void launch_topk(const void* q, int nq, const void* t, long long nt, const int* init_thr, uint32_t* parts, const ChunkPlan& p, hipStream_t s,
                        const char* timer_name = "hamming_topk", const uint64_t* floor_keys = nullptr, uint32_t floor_base = 0) {
    constexpr int xcd = 1;   // XCD-aware chunk placement (profiles/r01/match_xcd_placement_ab.log)
    const u32x16* tr = static_cast<const u32x16*>(t);
    const u32x4* qq = static_cast<const u32x4*>(q);
    // Occupancy cap: the kernels use no LDS, so an (unused) dynamic LDS request of `cap` bytes per workgroup bounds the
    // workgroups resident per CU (160 KB / cap). Two waves per SIMD already issue at full VALU rate; capping there leaves
    // registers, wave slots and the rest of the LDS free, so the short kernels of the other pipeline stages are dispatched
    // at once instead of waiting for a match wave to retire. Applies to every variant of the scan (all T, all K).
    const size_t cap = (size_t)std::max(0, tl_scan_cap >= 0 ? tl_scan_cap : match_lds_cap().load(std::memory_order_relaxed));
    last_scan_launch_lds().store((int)cap, std::memory_order_relaxed);
    if constexpr (K == 16) {
        if (floor_keys) {   // one page of a paged scan
            dim3 grid((unsigned)(p.chunks * p.qtiles_blocks)), block(256);
            KernelTimer timer(timer_name, s);
            hipLaunchKernelGGL((hamming_topk_page_kernel<K>), grid, block, cap, s, tr, (int)nt, qq, nq, p.rows_per_chunk, init_thr, parts,
                               p.qtiles_blocks, p.chunks, floor_keys, floor_base);
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    if (K > 2) {   // larger k keeps K (distance, index) pairs per query in registers: one query per lane
        dim3 grid((unsigned)(ceil_div(p.chunks, 8) * 8 * p.qtiles_blocks)), block(256);
        KernelTimer timer(timer_name, s);
        hipLaunchKernelGGL((hamming_topk_kernel<1, K>), grid, block, cap, s, tr, (int)nt, qq, nq, p.rows_per_chunk, init_thr, parts, p.qtiles_blocks,
                           p.chunks, 0);
        HIP_CHECK(hipGetLastError());
        return;
    }
    dim3 grid((unsigned)(ceil_div(p.chunks, 8) * 8 * p.qtiles_blocks)), block(256);
    KernelTimer timer(timer_name, s);
    switch (p.T) {
        case 4: hipLaunchKernelGGL((hamming_topk_kernel<4, K>), grid, block, cap, s, tr, (int)nt, qq, nq, p.rows_per_chunk, init_thr, parts, p.qtiles_blocks, p.chunks, xcd); break;
        case 2: hipLaunchKernelGGL((hamming_topk_kernel<2, K>), grid, block, cap, s, tr, (int)nt, qq, nq, p.rows_per_chunk, init_thr, parts, p.qtiles_blocks, p.chunks, xcd); break;
        default: hipLaunchKernelGGL((hamming_topk_kernel<1, K>), grid, block, cap, s, tr, (int)nt, qq, nq, p.rows_per_chunk, init_thr, parts, p.qtiles_blocks, p.chunks, xcd); break;
    }
    HIP_CHECK(hipGetLastError());
}

template <int K>
static size_t record_words(int nq, const ChunkPlan& p) {   // u32 words of the record buffer of one launch
    const size_t wtiles = (size_t)ceil_div(nq, 64 * p.T);
    const size_t pitch = p.T == 4 ? PartRecord<4, K>::PITCH : (p.T == 2 ? PartRecord<2, K>::PITCH : PartRecord<1, K>::PITCH);
    return (size_t)p.chunks * wtiles * pitch;
}

template <int K>
static void merge_records_launch(const uint32_t* recs, const ChunkPlan& p, uint32_t row_base, const uint64_t* extra, int nq, uint64_t* out, hipStream_t s) {
    const dim3 grid(ceil_div(nq, 64 * p.T)), block(256);   // one block per query tile
    switch (p.T) {
        case 4: hipLaunchKernelGGL((merge_records_kernel<4, K>), grid, block, 0, s, recs, p.chunks, p.rows_per_chunk, row_base, extra, nq, out); break;
        case 2: hipLaunchKernelGGL((merge_records_kernel<2, K>), grid, block, 0, s, recs, p.chunks, p.rows_per_chunk, row_base, extra, nq, out); break;
        default: hipLaunchKernelGGL((merge_records_kernel<1, K>), grid, block, 0, s, recs, p.chunks, p.rows_per_chunk, row_base, extra, nq, out); break;
    }
}

template <int K>
static void merge_launch(const uint64_t* parts, int nparts, int nq, uint64_t* out, hipStream_t s) {
    hipLaunchKernelGGL((merge_topk_kernel<K>), dim3(ceil_div(nq, 256)), dim3(256), 0, s, parts, nparts, nq, out);
}

template <int K>
static void topk_device_k(const void* q, int nq, const void* t, long long nt, uint32_t index_base, uint64_t* out, hipStream_t s) {
    ThreadCtx& c = ctx();
    // Phase 0 (only for large scans): exact top-k over the first `sample` rows gives per-query thresholds that
    // every chunk starts from, so the rare-hit fast path is reached immediately. The sample rows have the
    // lowest indices, hence a later row at equal distance never outranks them: strict '<' stays exact.
    // (1/16 of the rows, at most APDS_MATCH_SAMPLE = 16384, from 32768 rows up: a 125k-row shard of an 8-GPU run still gets one)
    const int sample_rows = config().match_sample;
    long long sample = 0;
    if (sample_rows > 0 && nt >= 32768) sample = std::min<long long>(sample_rows, (nt / 16) & ~1023ll);
    const int* thr = nullptr;
    uint64_t* sample_keys = nullptr;
    if (sample) {
        ChunkPlan sp = plan_chunks(nq, sample, true, K > 2);
        uint32_t* sparts = c.alloc_n<uint32_t>(record_words<K>(nq, sp));
        sample_keys = c.alloc_n<uint64_t>((size_t)nq * K);
        launch_topk<K>(q, nq, t, sample, nullptr, sparts, sp, s, "hamming_topk_sample");
        merge_records_launch<K>(sparts, sp, index_base, nullptr, nq, sample_keys, s);
        int* thr_buf = c.alloc_n<int>(nq);
        hipLaunchKernelGGL(thr_from_keys_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, sample_keys, nq, K, thr_buf);
        thr = thr_buf;
    }
    const char* rest = static_cast<const char*>(t) + (size_t)sample * 64;
    const long long nrest = nt - sample;
    ChunkPlan p = plan_chunks(nq, nrest, false, K > 2);
    uint32_t* parts = c.alloc_n<uint32_t>(record_words<K>(nq, p));   // [chunks][wave tiles] records
    launch_topk<K>(q, nq, rest, nrest, thr, parts, p, s);
    // the sample pass's result joins the merge as one more (already expanded) list
    merge_records_launch<K>(parts, p, index_base + (uint32_t)sample, sample_keys, nq, out, s);
    HIP_CHECK(hipGetLastError());
}

// k > 16 (BFMatcher::knnMatch takes any k, lib.rs:94-103): pages of 16. Page j is the plain K = 16 scan restricted to the keys above the
// last key of page j - 1 (keys are unique, so "above the floor" removes exactly the rows already reported); every page is a full pass over
// the train rows, with its own threshold pre-pass under the same floor. The scratch buffers are shared by the pages (stream order).
__global__ void gather_pages_kernel(const uint64_t* __restrict__ pages, int nq, int n_pages, int k, uint64_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nq * k) return;
    const int q = (int)(i / k), c = (int)(i - (long long)q * k);
    const int page = c >> 4;
    out[i] = page < n_pages ? pages[((size_t)page * nq + q) * 16 + (c & 15)] : EMPTY_KEY;
}

static long long split_sample_rows(long long nt);

static void topk_paged_device(const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, uint64_t* out, hipStream_t s) {
    constexpr int K = 16;
    ThreadCtx& c = ctx();
    const int n_pages = (int)std::min<long long>(ceil_div(k, K), ceil_div(nt, K));   // pages past the last train row would be empty
    const long long sample = split_sample_rows(nt);
    const ChunkPlan p = plan_chunks(nq, nt - sample, false, true);
    ChunkPlan sp{};
    uint32_t* sparts = nullptr;
    uint64_t* sample_keys = nullptr;
    int* thr = nullptr;
    if (sample) {
        sp = plan_chunks(nq, sample, true, true);
        sparts = c.alloc_n<uint32_t>(record_words<K>(nq, sp));
        sample_keys = c.alloc_n<uint64_t>((size_t)nq * K);
        thr = c.alloc_n<int>(nq);
    }
    uint32_t* parts = c.alloc_n<uint32_t>(record_words<K>(nq, p));
    uint64_t* pages = c.alloc_n<uint64_t>((size_t)n_pages * nq * K);
    const char* rest = static_cast<const char*>(t) + (size_t)sample * 64;
    for (int j = 0; j < n_pages; j++) {
        const uint64_t* floor_keys = j ? pages + (size_t)(j - 1) * nq * K : nullptr;
        if (sample) {
            launch_topk<K>(q, nq, t, sample, nullptr, sparts, sp, s, "hamming_topk_sample", floor_keys, index_base);
            merge_records_launch<K>(sparts, sp, index_base, nullptr, nq, sample_keys, s);
            hipLaunchKernelGGL(thr_from_keys_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, (const uint64_t*)sample_keys, nq, K, thr);
        }
        launch_topk<K>(q, nq, rest, nt - sample, thr, parts, p, s, "hamming_topk", floor_keys, index_base + (uint32_t)sample);
        merge_records_launch<K>(parts, p, index_base + (uint32_t)sample, sample_keys, nq, pages + (size_t)j * nq * K, s);
    }
    hipLaunchKernelGGL(gather_pages_kernel, dim3((unsigned)ceil_div((long long)nq * k, 256)), dim3(256), 0, s, (const uint64_t*)pages, nq, n_pages, k, out);
    HIP_CHECK(hipGetLastError());
}

// ---- the same scan in three separately launched steps, for pipelines that overlap consecutive frames ---------------------------------
// topk_device_k runs threshold pre-pass -> main scan -> merge back to back on one stream: per frame ~0.6 ms of pre-pass, ~0.2 ms of
// merge and four dependent-launch gaps sit between two main scans. With the intermediate buffers owned by a per-frame state object
// (instead of the calling thread's workspace, which the next call reuses), frame i + 1's pre-pass can run on a second stream while
// frame i's main scan is on the GPU and frame i - 1's merge on a third: the main scans then follow each other directly.
struct TopkSplitState {
    int device = 0;
    char* buf = nullptr;
    size_t cap = 0;
    // layout of the current frame (offsets into buf), filled by the pre-pass
    size_t off_sparts = 0, off_sample_keys = 0, off_thr = 0, off_parts = 0;
    int nq = 0, k = 0;
    long long nt = 0, sample = 0;
    ChunkPlan sp{}, p{};
    // the matrix-core form (hamming_mfma.hip): expanded queries, the split lists, and the expanded train rows - the caller's resident copy
    // (topk_split_use_train) when it is one of exactly these rows, this state's own otherwise (expanded by every pre-pass)
    bool mfma = false;
    HmPlan hp{};
    const HmTrain* shared_train = nullptr;
    uint32_t index_base_mfma = 0;
    size_t off_q4 = 0, off_qp = 0, off_t4 = 0, off_tp = 0, off_top2 = 0;
    const void* t4 = nullptr;
    const float* tp = nullptr;
};

static void split_reserve(TopkSplitState& st, size_t need) {
    if (need <= st.cap) return;
    // grow-only; a frame still using the old buffer is finished first (rare: the first frames of a run)
    HIP_CHECK(hipDeviceSynchronize());
    if (st.buf) (void)hipFree(st.buf);
    st.buf = nullptr;
    st.cap = 0;
    const size_t want = need + need / 4;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.buf), want));
    st.cap = want;
}

static void split_prepass_mfma(TopkSplitState& st, const void* q, int nq, const void* t, long long nt, int k, hipStream_t s) {
    st.mfma = true;
    st.nq = nq;
    st.k = k;
    st.nt = nt;
    st.hp = hm_plan(nq, nt);
    const bool shared = st.shared_train && st.shared_train->src == t && st.shared_train->n == nt;
    size_t need = 0;
    auto take = [&](size_t bytes) {
        const size_t o = need;
        need += (bytes + 255) & ~(size_t)255;
        return o;
    };
    st.off_q4 = take((size_t)nq * 256);
    st.off_qp = take((size_t)nq * 4);
    st.off_parts = take((size_t)st.hp.splits * nq * 16);
    st.off_top2 = take((size_t)nq * 16);
    if (!shared) {
        st.off_t4 = take((size_t)hm_padded_rows(nt) * 256);
        st.off_tp = take((size_t)hm_padded_rows(nt) * 4);
    }
    split_reserve(st, need);
    KernelTimer timer("hamming_topk_sample", s);
    hm_expand_device(q, nq, true, st.buf + st.off_q4, reinterpret_cast<float*>(st.buf + st.off_qp), s);
    if (shared) {
        st.t4 = st.shared_train->rows;
        st.tp = st.shared_train->pc;
    } else {
        hm_expand_device(t, nt, false, st.buf + st.off_t4, reinterpret_cast<float*>(st.buf + st.off_tp), s);
        st.t4 = st.buf + st.off_t4;
        st.tp = reinterpret_cast<const float*>(st.buf + st.off_tp);
    }
    HIP_CHECK(hipGetLastError());
}
static void split_scan_mfma(TopkSplitState& st, hipStream_t s) {
    hm_scan_device(st.buf + st.off_q4, reinterpret_cast<const float*>(st.buf + st.off_qp), st.nq, st.t4, st.tp, st.nt, st.hp, st.index_base_mfma,
                   reinterpret_cast<uint64_t*>(st.buf + st.off_parts), s);
}
static void split_merge_mfma(TopkSplitState& st, uint64_t* out, hipStream_t s) {
    const uint64_t* parts = reinterpret_cast<const uint64_t*>(st.buf + st.off_parts);
    uint64_t* top2 = st.k == 2 ? out : reinterpret_cast<uint64_t*>(st.buf + st.off_top2);
    if (st.hp.splits > 1) merge_topk_device(parts, st.hp.splits, st.nq, 2, top2, s);
    else if (st.k == 2) HIP_CHECK(hipMemcpyAsync(out, parts, (size_t)st.nq * 16, hipMemcpyDeviceToDevice, s));
    else top2 = const_cast<uint64_t*>(parts);
    if (st.k == 1) take_first_columns_device(top2, st.nq, 2, 1, out, s);
    HIP_CHECK(hipGetLastError());
}

static long long split_sample_rows(long long nt) {
    const int sample_rows = config().match_sample;
    return (sample_rows > 0 && nt >= 32768) ? std::min<long long>(sample_rows, (nt / 16) & ~1023ll) : 0;
}

template <int K>
static void split_prepass_k(TopkSplitState& st, const void* q, int nq, const void* t, long long nt, uint32_t index_base, hipStream_t s) {
    st.nq = nq;
    st.k = K;
    st.nt = nt;
    st.sample = split_sample_rows(nt);
    st.p = plan_chunks(nq, nt - st.sample, false, false);
    size_t need = 0;
    auto take = [&](size_t bytes) {
        const size_t o = need;
        need += (bytes + 255) & ~(size_t)255;
        return o;
    };
    if (st.sample) {
        st.sp = plan_chunks(nq, st.sample, true, false);
        st.off_sparts = take(record_words<K>(nq, st.sp) * 4);
        st.off_sample_keys = take((size_t)nq * K * 8);
        st.off_thr = take((size_t)nq * 4);
    }
    st.off_parts = take(record_words<K>(nq, st.p) * 4);
    split_reserve(st, need);
    if (!st.sample) return;
    uint32_t* sparts = reinterpret_cast<uint32_t*>(st.buf + st.off_sparts);
    uint64_t* sample_keys = reinterpret_cast<uint64_t*>(st.buf + st.off_sample_keys);
    launch_topk<K>(q, nq, t, st.sample, nullptr, sparts, st.sp, s, "hamming_topk_sample");
    merge_records_launch<K>(sparts, st.sp, index_base, nullptr, nq, sample_keys, s);
    hipLaunchKernelGGL(thr_from_keys_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, sample_keys, nq, K, reinterpret_cast<int*>(st.buf + st.off_thr));
    HIP_CHECK(hipGetLastError());
}

template <int K>
static void split_scan_k(TopkSplitState& st, const void* q, const void* t, hipStream_t s) {
    const char* rest = static_cast<const char*>(t) + (size_t)st.sample * 64;
    launch_topk<K>(q, st.nq, rest, st.nt - st.sample, st.sample ? reinterpret_cast<const int*>(st.buf + st.off_thr) : nullptr,
                   reinterpret_cast<uint32_t*>(st.buf + st.off_parts), st.p, s);
}

template <int K>
static void split_merge_k(TopkSplitState& st, uint32_t index_base, uint64_t* out, hipStream_t s) {
    merge_records_launch<K>(reinterpret_cast<const uint32_t*>(st.buf + st.off_parts), st.p, index_base + (uint32_t)st.sample,
                            st.sample ? reinterpret_cast<const uint64_t*>(st.buf + st.off_sample_keys) : nullptr, st.nq, out, s);
    HIP_CHECK(hipGetLastError());
}

void* topk_split_create() {
    TopkSplitState* st = new TopkSplitState();
    st->device = ctx().device;
    return st;
}
void topk_split_destroy(void* h) {
    TopkSplitState* st = static_cast<TopkSplitState*>(h);
    if (!st) return;
    int previous = -1;   // the caller's device is put back: its thread context (stream, workspace) belongs there
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    (void)hipSetDevice(st->device);
    (void)hipDeviceSynchronize();
    if (st->buf) (void)hipFree(st->buf);
    delete st;
    if (previous >= 0) (void)hipSetDevice(previous);
}
void topk_split_prepass(void* h, const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, hipStream_t s) {
    APDS_REQUIRE(h, APDS_ERR_BAD_ARG, "null scan state");
    APDS_REQUIRE(k == 1 || k == 2, APDS_ERR_ASSERT, "the split scan serves k = 1 and k = 2 (what the crate surface consumes, lib.rs:107-111)");
    APDS_REQUIRE(nq > 0 && nt > 0 && nt < (1ll << 31), APDS_ERR_ASSERT, "the split scan needs queries and train rows");
    TopkSplitState& st = *static_cast<TopkSplitState*>(h);
    if (config().match_mfma) {
        st.index_base_mfma = index_base;
        split_prepass_mfma(st, q, nq, t, nt, k, s);
        return;
    }
    st.mfma = false;
    if (k == 1) split_prepass_k<1>(st, q, nq, t, nt, index_base, s);
    else split_prepass_k<2>(st, q, nq, t, nt, index_base, s);
}
void topk_split_use_train(void* h, const void* hm_train) {
    APDS_REQUIRE(h, APDS_ERR_BAD_ARG, "null scan state");
    static_cast<TopkSplitState*>(h)->shared_train = static_cast<const HmTrain*>(hm_train);
}
void topk_split_scan(void* h, const void* q, const void* t, hipStream_t s) {
    APDS_REQUIRE(h, APDS_ERR_BAD_ARG, "null scan state");
    TopkSplitState& st = *static_cast<TopkSplitState*>(h);
    APDS_REQUIRE(st.nq > 0, APDS_ERR_ASSERT, "scan before pre-pass");
    if (st.mfma) {
        split_scan_mfma(st, s);
        return;
    }
    if (st.k == 1) split_scan_k<1>(st, q, t, s);
    else split_scan_k<2>(st, q, t, s);
}
void topk_split_merge(void* h, uint32_t index_base, uint64_t* out, hipStream_t s) {
    APDS_REQUIRE(h && out, APDS_ERR_BAD_ARG, "null scan state / output");
    TopkSplitState& st = *static_cast<TopkSplitState*>(h);
    APDS_REQUIRE(st.nq > 0, APDS_ERR_ASSERT, "merge before pre-pass");
    if (st.mfma) {
        APDS_REQUIRE(index_base == st.index_base_mfma, APDS_ERR_ASSERT, "the merge's index base differs from the pre-pass's");
        split_merge_mfma(st, out, s);
        return;
    }
    if (st.k == 1) split_merge_k<1>(st, index_base, out, s);
    else split_merge_k<2>(st, index_base, out, s);
}

// Full top-k of nq queries over nt train rows (device, 64-byte rows). out: nq*k keys. k in {1,2} is the tuned path (the reference only
// ever consumes the two nearest, lib.rs:107-111); 3 <= k <= 16 run with one query per lane, larger k in pages of 16 (topk_paged_device).
void hamming_topk_device(const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, uint64_t* out,
                         hipStream_t s, int backend) {
    APDS_REQUIRE(k >= 1, APDS_ERR_ASSERT, "top-k needs k >= 1");
    APDS_REQUIRE(nt < (1ll << 31), APDS_ERR_ASSERT, "train set too large for one call; shard it");
    if (nq <= 0) return;
    if (nt <= 0) {
        HIP_CHECK(hipMemsetAsync(out, 0xFF, (size_t)nq * k * 8, s));
        return;
    }
    if (k > 16) {
        topk_paged_device(q, nq, t, nt, index_base, k, out, s);
        return;
    }
    APDS_REQUIRE(backend >= 0 && backend <= 2 && !(backend == 2 && k > 2), APDS_ERR_ASSERT, "backend: 0 default, 1 vector ALU, 2 matrix cores (k <= 2)");
    if (k <= 2 && (backend == 2 || (backend == 0 && config().match_mfma))) {   // the two nearest (all the crate surface consumes) come from the matrix cores
        hamming_mfma_topk_device(q, nq, t, nt, index_base, k, out, s);
        return;
    }
    const int K = k <= 2 ? k : (k <= 4 ? 4 : (k <= 8 ? 8 : 16));
    uint64_t* dst = K == k ? out : ctx().alloc_n<uint64_t>((size_t)nq * K);
    switch (K) {
        case 1: topk_device_k<1>(q, nq, t, nt, index_base, dst, s); break;
        case 2: topk_device_k<2>(q, nq, t, nt, index_base, dst, s); break;
        case 4: topk_device_k<4>(q, nq, t, nt, index_base, dst, s); break;
        case 8: topk_device_k<8>(q, nq, t, nt, index_base, dst, s); break;
        default: topk_device_k<16>(q, nq, t, nt, index_base, dst, s); break;
    }
    if (K != k) {
        hipLaunchKernelGGL(take_first_columns_kernel, dim3(ceil_div((long long)nq * k, 256)), dim3(256), 0, s, (const uint64_t*)dst, nq, K, k, out);
        HIP_CHECK(hipGetLastError());
    }
}

void take_first_columns_device(const uint64_t* in, int nq, int kin, int kout, uint64_t* out, hipStream_t s) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(take_first_columns_kernel, dim3(ceil_div((long long)nq * kout, 256)), dim3(256), 0, s, in, nq, kin, kout, out);
}

void merge_topk_device(const uint64_t* parts, int nparts, int nq, int k, uint64_t* out, hipStream_t s) {
    if (nq <= 0) return;
    switch (k) {
        case 1: merge_launch<1>(parts, nparts, nq, out, s); break;
        case 2: merge_launch<2>(parts, nparts, nq, out, s); break;
        case 4: merge_launch<4>(parts, nparts, nq, out, s); break;
        case 8: merge_launch<8>(parts, nparts, nq, out, s); break;
        case 16: merge_launch<16>(parts, nparts, nq, out, s); break;
        default:
            APDS_REQUIRE(k >= 1 && k <= 4096 && nparts <= 64, APDS_ERR_ASSERT, "merge supports 1 <= k <= 4096 over at most 64 lists");
            hipLaunchKernelGGL(merge_topk_any_kernel, dim3(ceil_div(nq, 64)), dim3(64), 0, s, parts, nparts, nq, k, out);
    }
    HIP_CHECK(hipGetLastError());
}

void pack_rows_device(const void* src, long long n, int desc_bytes, long long src_stride, void* dst, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(n * 16, 256)), dim3(256), 0, s, static_cast<const uint8_t*>(src), n, desc_bytes, src_stride,
                       static_cast<uint32_t*>(dst));
    HIP_CHECK(hipGetLastError());
}

// flags (n bytes) -> block offsets; returns device pointers for the emit kernel; *total_dev holds the count
int* scan_flags_device(const uint8_t* flags, int n, int** total_dev, hipStream_t s) {
    ThreadCtx& c = ctx();
    const int nblocks = std::max(1, ceil_div(n, SCAN_BLOCK));
    int* block_counts = c.alloc_n<int>(nblocks + 1);
    int* total = block_counts + nblocks;
    hipLaunchKernelGGL(scan_block_counts_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, flags, n, block_counts);
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(1024), 0, s, block_counts, nblocks, total);
    HIP_CHECK(hipGetLastError());
    *total_dev = total;
    return block_counts;
}

int ratio_filter_device(const uint64_t* keys, int nq, int k, float fs, apds_dmatch* out, hipStream_t s) {
    if (nq <= 0) return 0;
    ThreadCtx& c = ctx();
    uint8_t* flags = c.alloc_n<uint8_t>(nq);
    hipLaunchKernelGGL(ratio_flag_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, keys, nq, k, fs, flags);
    int* total_dev = nullptr;
    int* offs = scan_flags_device(flags, nq, &total_dev, s);
    hipLaunchKernelGGL(emit_ratio_matches_kernel, dim3(ceil_div(nq, SCAN_BLOCK)), dim3(SCAN_BLOCK), 0, s, keys, nq, k, flags, offs, out);
    HIP_CHECK(hipGetLastError());
    int total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, total_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return total;
}

int cross_check_device(const uint64_t* train_best, long long n_train, int nq, apds_dmatch* out, hipStream_t s) {
    if (nq <= 0 || n_train <= 0) return 0;
    ThreadCtx& c = ctx();
    uint64_t* best = c.alloc_n<uint64_t>(nq);
    HIP_CHECK(hipMemsetAsync(best, 0xFF, (size_t)nq * 8, s));
    hipLaunchKernelGGL(crosscheck_scatter_kernel, dim3(ceil_div(n_train, 256)), dim3(256), 0, s, train_best, n_train,
                       reinterpret_cast<unsigned long long*>(best));
    uint8_t* flags = c.alloc_n<uint8_t>(nq);
    hipLaunchKernelGGL(nonempty_flag_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, best, nq, flags);
    int* total_dev = nullptr;
    int* offs = scan_flags_device(flags, nq, &total_dev, s);
    hipLaunchKernelGGL(emit_ratio_matches_kernel, dim3(ceil_div(nq, SCAN_BLOCK)), dim3(SCAN_BLOCK), 0, s, best, nq, 1, flags, offs, out);
    HIP_CHECK(hipGetLastError());
    int total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, total_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return total;
}

// One launch configuration of the microbenchmark: `waves_per_simd` workgroups of 256 threads resident per CU (one wave of each
// on every SIMD; bounded through an unused dynamic-LDS request), a grid eight rounds deep.
struct ValuPeak {
    double lane_ops_per_s;      // wall clock (HIP events), lane-ops as defined per mode (a packed instruction counts two)
    double cycles_per_inst;     // s_memtime cycles per wave-instruction per SIMD = mean wave span / (instructions per wave * waves per SIMD)
};

template <int MODE>
static ValuPeak run_valu_peak(int waves_per_simd, hipStream_t st) {
    ThreadCtx& c = ctx();
    const int w = std::min(std::max(waves_per_simd, 1), 8);
    const int iters = 2048, cus = 256, blocks = cus * w * 8;
    uint32_t* sink = c.alloc_n<uint32_t>(64);
    unsigned long long* spans = c.alloc_n<unsigned long long>((size_t)blocks * 4);
    // 160 KB of LDS per CU: a request of 160 KB / w (minus the granule) admits exactly w workgroups
    const size_t lds = w >= 8 ? 0 : (size_t)(160 * 1024 / w) - (w == 1 ? 0 : 1024);
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&valu_peak_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    u32x16* rowp = reinterpret_cast<u32x16*>(c.alloc_n<uint32_t>(16 * 8));
    {
        uint32_t h[16 * 8];
        for (int i = 0; i < 16 * 8; i++) h[i] = 0x9E3779B9u * (uint32_t)(i + 1);
        HIP_CHECK(hipMemcpyAsync(rowp, h, sizeof(h), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL((valu_peak_kernel<MODE>), dim3(blocks), dim3(256), lds, st, sink, spans, 32, rowp);
    ValuPeak best{0, 0};
    std::vector<unsigned long long> host((size_t)blocks * 4);
    const double per_inst = MODE == 0 ? 1 : ((MODE == 6 || MODE == 7) ? 2 : 1);   // lane-ops per instruction per lane
    const double insts_per_wave = MODE >= 12 ? (double)iters * 120 : (double)iters * VALU_UNROLL * VALU_CHAINS * (MODE == 0 ? 2 : 1);
    for (int rep = 0; rep < 3; rep++) {
        HIP_CHECK(hipEventRecord(a, st));
        hipLaunchKernelGGL((valu_peak_kernel<MODE>), dim3(blocks), dim3(256), lds, st, sink, spans, iters, rowp);
        HIP_CHECK(hipEventRecord(b, st));
        HIP_CHECK(hipEventSynchronize(b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        const double ops = (double)blocks * 256 * insts_per_wave * per_inst;
        if (ops / (ms * 1e-3) > best.lane_ops_per_s) {
            HIP_CHECK(hipMemcpy(host.data(), spans, host.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (unsigned long long v : host) sum += (double)v;
            best.lane_ops_per_s = ops / (ms * 1e-3);
            best.cycles_per_inst = sum / (double)host.size() / (insts_per_wave * w);
        }
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return best;
}

int valu_peak_modes() { return VALU_MODES; }

const char* valu_peak_mode_name(int mode) {
    static const char* names[VALU_MODES] = {"v_xor_b32(s,v)+v_bcnt_u32_b32", "v_xor_b32(s,v)", "v_bcnt_u32_b32", "v_add_u32(s,v)", "v_fma_f32", "v_add_f32",
                                            "v_pk_fma_f32", "v_pk_add_f32", "v_xor_b32(v,v)", "v_bfi_b32", "v_and_b32(v,v)", "v_mul_f32",
                                            "row x 4 queries: query-sequential", "row x 4 queries: dword-major", "dword-major, xor 1 ahead",
                                            "dword-major, row in VGPRs", "query-sequential, xor 1 ahead", "dword-major, xor 2 ahead",
                                            "query-sequential + s_nop before each bcnt", "dword-major + s_nop before each bcnt",
                                            "query-sequential + s_nop after every op", "query-sequential, one SGPR",
                                            "bcnt + s_nop only (75 ops counted as 120)", "X X nop B B", "X s_nop(1) B", "B nop X", "X s_add B", "dword-major X nop B"};
    return mode >= 0 && mode < VALU_MODES ? names[mode] : "?";
}

void valu_peak_device(int mode, int waves_per_simd, double* lane_ops_per_s, double* cycles_per_inst) {
    hipStream_t st = ctx().stream;
    ValuPeak r{0, 0};
    switch (mode) {
        case 0: r = run_valu_peak<0>(waves_per_simd, st); break;
        case 1: r = run_valu_peak<1>(waves_per_simd, st); break;
        case 2: r = run_valu_peak<2>(waves_per_simd, st); break;
        case 3: r = run_valu_peak<3>(waves_per_simd, st); break;
        case 4: r = run_valu_peak<4>(waves_per_simd, st); break;
        case 5: r = run_valu_peak<5>(waves_per_simd, st); break;
        case 6: r = run_valu_peak<6>(waves_per_simd, st); break;
        case 7: r = run_valu_peak<7>(waves_per_simd, st); break;
        case 8: r = run_valu_peak<8>(waves_per_simd, st); break;
        case 9: r = run_valu_peak<9>(waves_per_simd, st); break;
        case 10: r = run_valu_peak<10>(waves_per_simd, st); break;
        case 11: r = run_valu_peak<11>(waves_per_simd, st); break;
        case 12: r = run_valu_peak<12>(waves_per_simd, st); break;
        case 13: r = run_valu_peak<13>(waves_per_simd, st); break;
        case 14: r = run_valu_peak<14>(waves_per_simd, st); break;
        case 15: r = run_valu_peak<15>(waves_per_simd, st); break;
        case 16: r = run_valu_peak<16>(waves_per_simd, st); break;
        case 17: r = run_valu_peak<17>(waves_per_simd, st); break;
        case 18: r = run_valu_peak<18>(waves_per_simd, st); break;
        case 19: r = run_valu_peak<19>(waves_per_simd, st); break;
        case 20: r = run_valu_peak<20>(waves_per_simd, st); break;
        case 21: r = run_valu_peak<21>(waves_per_simd, st); break;
        case 22: r = run_valu_peak<22>(waves_per_simd, st); break;
        case 23: r = run_valu_peak<23>(waves_per_simd, st); break;
        case 24: r = run_valu_peak<24>(waves_per_simd, st); break;
        case 25: r = run_valu_peak<25>(waves_per_simd, st); break;
        case 26: r = run_valu_peak<26>(waves_per_simd, st); break;
        case 27: r = run_valu_peak<27>(waves_per_simd, st); break;
        default: fail(APDS_ERR_BAD_ARG, "valu peak: mode out of range");
    }
    if (lane_ops_per_s) *lane_ops_per_s = r.lane_ops_per_s;
    if (cycles_per_inst) *cycles_per_inst = r.cycles_per_inst;
}

// lane-ops/s of the xor+bcnt pair at full occupancy: the denominator bench.py divides the match kernel by
double valu_popcount_peak_device() {
    double v = 0;
    valu_peak_device(0, 8, &v, nullptr);
    return v;
}

}  // namespace apds
