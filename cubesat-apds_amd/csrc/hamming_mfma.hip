// csrc/hamming_mfma.hip — Hamming top-1 / top-2 of 512-bit rows on the FP4 matrix pipe (get_knn_matches, lib.rs:94-114, k <= 2).
//
// Brute-force Hamming matching is all pairs x all bits: Q x N x 512 one-bit products. match_hamming.hip forms them on the vector ALU
// (xor + popcount per dword: 32 lane-operations per pair, bound by the half-rate v_bcnt at 52 T lane-op/s = 1.6e12 pairs/s). CDNA4's
// matrix cores multiply 4-bit operands at ~10 PFLOP/s dense, and a bit IS a 4-bit float: with
//     train bit  -> e2m1  1.0 (0x2)        query bit -> e2m1 -2.0 (0xC)        accumulator preset to popcount(train row)
// one v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales) adds -2 * (t AND q) over 128 bit positions for 16 x 16 pairs, and after four of
// them the accumulator holds popcount(t) - 2 popcount(t AND q) = hamming(t, q) - popcount(q): the ranking value of the pair, EXACT (every
// product is 0 or -2, every partial sum an integer of magnitude <= 1024: nothing rounds in binary32). The distances, the ties (lower train
// row first) and therefore the keys are those of hamming_topk_kernel bit for bit (tests/test_match_gpu.py runs both).
// 2 * 512 flop per pair on the 10 PF pipe is a ceiling of 9.8e12 pairs/s, six times the vector formulation's.
//
// Two steps per call:
//   hm_expand_rows_kernel   64-byte rows -> 256-byte rows of fp4 nibbles (one dword -> 16 bytes; bit k of dword d is element 32 d + k;
//                           any order would do as long as both operands use the same one) + popcount per row as a float
//   hamming_mfma_kernel     the structure of l2_screen_kernel (same tile shapes: there K = 128 bf16 elements are 256 bytes, here K = 512
//                           fp4 elements are): block = 8 waves x 48 queries held as B operands in registers for the whole kernel,
//                           128-row train tiles double-buffered in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no staging registers,
//                           no ds_write) into an XOR-swizzled image (conflict-free ds_read_b128), each A read feeds three MFMAs, running
//                           top-2 per query column in the lanes, insertion code only when some lane has a hit. (Ranking a block's
//                           values under the NEXT block's MFMAs was written twice - values carried over, ranking forced between the
//                           MFMA halves - and both times the compiler put the twelve MFMAs of a block back together and ranked
//                           behind them; the wait states that costs, s_nop 1 + five ds_reads, are filled by the SIMD's other waves.)
#include <atomic>
#include <memory>

#include "config.h"
#include "kernels.h"

namespace apds {

typedef int hm_v8i __attribute__((ext_vector_type(8)));
typedef float hm_f32x4 __attribute__((ext_vector_type(4)));

#ifndef APDS_HM_WAVES
#define APDS_HM_WAVES 8
#endif
static constexpr int HM_WAVES = APDS_HM_WAVES;   // waves per workgroup; a wave stages 16 rows of a tile. (16: one 1024-thread workgroup per CU, 256-row
                                                 // tiles, half the barriers: 5.58 against 5.46 ms alone, 148.7 against 147.5 frames/s in the pipeline -
                                                 // no difference worth a second shape; profiles/r04/match_mfma_probe_w16.txt)
static constexpr int HM_TM = 16 * HM_WAVES;      // train rows per tile
#ifndef APDS_HM_WPE
#define APDS_HM_WPE 4
#endif
// waves per SIMD the kernel is compiled for (4: two 8-wave workgroups per CU. The 12-wave shape - APDS_HM_WAVES=12 APDS_HM_WPE=3, one
// workgroup per CU with a quarter of the registers left to the other stages' kernels - is a build-time experiment)
#ifndef APDS_HM_NC
#define APDS_HM_NC 3
#endif
static constexpr int HM_NC = APDS_HM_NC;      // 16-query column blocks per wave (4: 15 registers spill at four waves per SIMD; 7.0 against 5.9 ms on the
                                              // headline shape, 11.0 against 11.9 on 262143^2: profiles/r04/match_mfma_probe.txt)
static constexpr int HM_Q = HM_WAVES * 16 * HM_NC;   // queries per block
static constexpr int HM_UNIT_SCALE = 0x7F7F7F7F;   // E8M0 127 = 2^0 in every byte
static constexpr uint64_t HM_EMPTY = ~0ull;
// Train popcounts are stored with this bias: the ranking value popcount(t) + 1024 - 2 (t AND q) is then a POSITIVE float (>= 512), and
// positive floats order like their bit patterns - the epilogue compares them as unsigned integers (v_min3_u32 / v_min_u32 / v_cmp_lt_u32:
// no NaN canonicalisation in front of every float minimum, six v_max_f32 less per 16 x 48 block of pairs). +inf (rows past the end) is
// 0x7F800000: above every finite value, below the empty marker 0xFFFFFFFF.
static constexpr int HM_BIAS = 1024;

// 8 bits -> 8 nibbles holding `nib` where the bit is set
__device__ __forceinline__ uint32_t hm_spread8(uint32_t b, uint32_t nib) {
    uint32_t t = b & 0xFFu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t * nib;
}

// rows: n x 16 dwords. out: n x 16 uint4 (dword d of a row -> uint4 d). pc: popcount of each row + bias. One thread per dword.
__global__ __launch_bounds__(256) void hm_expand_rows_kernel(const uint32_t* __restrict__ rows, long long n, uint32_t nib, int bias, uint4* __restrict__ out,
                                                             float* __restrict__ pc) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const uint32_t d = i < n * 16 ? rows[i] : 0u;
    int v = __popc(d);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);   // the 16 lanes of a row are aligned: 256 threads = 16 rows
    if (i >= n * 16) return;
    out[i] = make_uint4(hm_spread8(d, nib), hm_spread8(d >> 8, nib), hm_spread8(d >> 16, nib), hm_spread8(d >> 24, nib));
    if ((threadIdx.x & 15) == 0) pc[i >> 4] = (float)(v + bias);
}

// the rows past the end of the last tile: zero operands, popcount +inf (they never rank). One block of 128 threads x 16.
__global__ void hm_pad_rows_kernel(uint4* __restrict__ rows, float* __restrict__ pc, long long n, long long n_pad) {
    const long long i = n + (threadIdx.x >> 4);
    if (i >= n_pad) return;
    for (long long r = i; r < n_pad; r += 8) {
        rows[r * 16 + (threadIdx.x & 15)] = make_uint4(0, 0, 0, 0);
        if ((threadIdx.x & 15) == 0) pc[r] = INFINITY;
    }
}

// thr[q]: the kernel's ranking value (hamming - popcount(query) + bias, as float bits) of the second key of a finished top-2 list
__global__ void hm_thresholds_kernel(const uint64_t* __restrict__ top2, const float* __restrict__ qpc, int nq, uint32_t* __restrict__ thr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint64_t k2 = top2[(size_t)i * 2 + 1];
    thr[i] = k2 == HM_EMPTY ? 0x7F800000u : __float_as_uint((float)((int)(uint32_t)(k2 >> 32) - (int)qpc[i] + HM_BIAS));
}

struct HmTop2 {
    uint32_t d0, d1;   // bit patterns of the (positive) ranking values
    uint32_t i0, i1;
};
__device__ __forceinline__ void hm_insert(HmTop2& b, uint32_t d, uint32_t idx) {   // rows arrive in ascending order: strict '<' keeps the lower row of a tie
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

// out[split][nq][2]: keys (distance << 32 | row + index_base), EMPTY where the split holds fewer than two rows.
// LDS image of a tile (no padding: the tile is filled by LDS-DMA, whose destination is wave-uniform base + 16 * lane): row r at 256 r, and
// its 16-byte chunk c at position c ^ (r & 15) - the 16 rows a ds_read_b128 group reads chunk c of then sit in 16 different bank groups.
// The DMA's per-lane SOURCE address applies the same involution, so the image is a plain lane-linear copy for the hardware.
// Rows and popcounts are padded to whole tiles (zero operands, +inf: rows past the end never rank).
// THR: the launch starts from thresholds (the main launch behind a threshold launch) - a template parameter so that profilers list the two
// launches of a match under two names
template <int PRIO, bool THR>
__global__ __launch_bounds__(64 * HM_WAVES) __attribute__((amdgpu_waves_per_eu(APDS_HM_WPE, APDS_HM_WPE))) void hamming_mfma_kernel(const uint4* __restrict__ train_fp4, const float* __restrict__ tpc, int n_train,
                                                           const uint4* __restrict__ query_fp4, const float* __restrict__ qpc, int nq, int tiles_per_split,
                                                           int q_tiles, int splits, uint32_t index_base, const uint32_t* __restrict__ thr,
                                                           uint64_t* __restrict__ out) {
    // (no APDS_RAISE_WAVE_PRIORITY here: this is the kernel the short kernels of the other stages raise their priority against)
    extern __shared__ __attribute__((aligned(128))) unsigned char hm_lds[];
    constexpr int TILE_BYTES = HM_TM * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has an L2 of its own. With a multiple of eight splits, split
    // x + 8 m belongs to XCD x: the workgroups resident on an XCD at any time are consecutive query tiles of one split, start together and
    // walk the same train tiles at about the same pace - one fetches a tile, the others find it in their L2.
    int split, qtile;
    if ((splits & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        split = xcd + 8 * (j / q_tiles);
        qtile = j % q_tiles;
    } else {
        split = blockIdx.x / q_tiles;
        qtile = blockIdx.x % q_tiles;
    }
    const int q0 = qtile * HM_Q + wave * 16 * HM_NC;              // this wave's queries
    const int n_tiles = (n_train + HM_TM - 1) / HM_TM;
    const int tile_begin = split * tiles_per_split, tile_end = min(n_tiles, tile_begin + tiles_per_split);
    const int col = lane & 15, kq = lane >> 4;                     // accumulator column / k chunk (operands) / row group (accumulators)

    // B operands: query (q0 + 16 c + col), elements 128 s + 32 kq .. + 31 (dword 4 s + kq of the row), for the four k steps s
    uint4 B[HM_NC][4];
    float qq[HM_NC];
#pragma unroll
    for (int c = 0; c < HM_NC; c++) {
        const int qi = min(q0 + 16 * c + col, nq - 1);
#pragma unroll
        for (int s = 0; s < 4; s++) B[c][s] = query_fp4[(size_t)qi * 16 + 4 * s + kq];
        qq[c] = qpc[qi];
    }
    // The running top-2 starts from thr[query] when the caller has one (the second smallest ranking value over a sample of EARLIER rows - lower
    // indices, so a row of this launch enters the query's final top-2 only with a strictly smaller value): both entries are that value
    // without a row, real rows push them out, and what is left of them at the end is written as empty. Without it: +inf.
    HmTop2 best[HM_NC];
#pragma unroll
    for (int c = 0; c < HM_NC; c++) {
        best[c].d0 = best[c].d1 = THR ? thr[min(q0 + 16 * c + col, nq - 1)] : 0x7F800000u;
        best[c].i0 = best[c].i1 = 0xFFFFFFFFu;
    }

    if (tile_begin < tile_end) {
        // staging by LDS-DMA: 32 pieces of 1 KB (4 rows) per tile, four per wave: wave w fills rows [16 w, 16 w + 16); + the popcounts
        // (two pieces of 64 floats, waves 0 and 1)
        // (the expanded rows are padded to whole tiles - zero rows with a popcount of +inf - so a piece's source is a uniform tile base plus a
        // per-lane constant: scalar address arithmetic only)
        int soff[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 16 * wave + 4 * i + (lane >> 4);
            soff[i] = r * 256 + (((lane & 15) ^ (r & 15)) << 4);
        }
        auto stage = [&](int tile, int buf) {
            const unsigned char* tbase = reinterpret_cast<const unsigned char*>(train_fp4) + (size_t)tile * TILE_BYTES;   // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tbase + soff[i]),
                                                 (__attribute__((address_space(3))) void*)(hm_lds + buf * TILE_BYTES + (16 * wave + 4 * i) * 256), 16, 0, 0);
            if (wave < HM_TM / 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tpc + (size_t)tile * HM_TM + 64 * wave + lane),
                                                 (__attribute__((address_space(3))) void*)(hm_lds + 2 * TILE_BYTES + buf * (HM_TM * 4) + 256 * wave), 4, 0, 0);
        };
        // this lane's four A reads of a 16-row block: row (16 rb + col), chunk 4 s + kq at position (4 s + kq) ^ col
        int aoff[4];
#pragma unroll
        for (int s = 0; s < 4; s++) aoff[s] = col * 256 + (((4 * s + kq) ^ col) << 4);
        const int noff = 2 * TILE_BYTES + 16 * kq;                // popcounts of rows 4 kq .. + 3 of a block

        hm_f32x4 acc[HM_NC];
        // Ranking a block: one minimum and one compare per 16 x 16 accumulator in the common case. Only the query blocks in which some lane
        // has a hit run insertion code (a wave-uniform branch each): the counters put the vector instructions beside the MFMAs at 1.9 per
        // MFMA when any hit sent all three query blocks through the twelve insertions, and an MFMA leaves the SIMD's issue port free for
        // only two of them. (One threshold per lane - the largest of the three - and one minimum over all twelve values in front of the
        // per-block tests: 7 instructions and one branch instead of 9 and three in the common case, and 8 % SLOWER on every shape
        // (profiles/r04/match_mfma_probe_single_threshold.txt): a chain of six dependent minima in front of the branch. The twelve MFMAs as
        // three chains of four with the previous chain's minimum / test placed between the next chain's MFMAs by sched_group_barrier: the
        // compiler follows the directives, keeps its wait states (it counts an MFMA as one), and nothing changes: 5.45 - 5.51 ms.)
        auto rank = [&](const hm_f32x4 (&a)[HM_NC], uint32_t row0) {
            bool hit[HM_NC];
#pragma unroll
            for (int c = 0; c < HM_NC; c++) {
                const uint32_t mn = min(min(__float_as_uint(a[c][0]), __float_as_uint(a[c][1])), min(__float_as_uint(a[c][2]), __float_as_uint(a[c][3])));
                hit[c] = mn < best[c].d1;
            }
#pragma unroll
            for (int c = 0; c < HM_NC; c++)
                if (__any(hit[c])) {
#pragma unroll
                    for (int j = 0; j < 4; j++) hm_insert(best[c], __float_as_uint(a[c][j]), row0 + j);
                }
        };

        stage(tile_begin, 0);
        __syncthreads();   // (drains the DMA: vmcnt(0) in front of the barrier)
        for (int tile = tile_begin; tile < tile_end; tile++) {
            const int buf = (tile - tile_begin) & 1;
            if (tile + 1 < tile_end) stage(tile + 1, buf ^ 1);     // lands while this tile is computed; the closing barrier waits for it
            const unsigned char* T = hm_lds + buf * TILE_BYTES;
            const unsigned char* Nn = hm_lds + noff + buf * (HM_TM * 4);
#pragma unroll
            for (int rb = 0; rb < HM_TM / 16; rb++) {                  // 16-row blocks of the tile
                const hm_f32x4 init = *reinterpret_cast<const hm_f32x4*>(Nn + rb * 64);   // biased popcounts of this lane's four rows
                uint4 a[4];
#pragma unroll
                for (int s = 0; s < 4; s++) a[s] = *reinterpret_cast<const uint4*>(T + rb * 4096 + aoff[s]);
                auto steps = [&](int s_lo, int s_hi) {
#pragma unroll
                    for (int s = s_lo; s < s_hi; s++) {
                        const hm_v8i A = {(int)a[s].x, (int)a[s].y, (int)a[s].z, (int)a[s].w, 0, 0, 0, 0};
#pragma unroll
                        for (int c = 0; c < HM_NC; c++) {
                            const hm_v8i Bv = {(int)B[c][s].x, (int)B[c][s].y, (int)B[c][s].z, (int)B[c][s].w, 0, 0, 0, 0};
                            acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, Bv, s == 0 ? init : acc[c], 4, 4, 0, HM_UNIT_SCALE, 0, HM_UNIT_SCALE);
                        }
                    }
                };
                // a wave about to feed the matrix pipe goes ahead of the SIMD's waves that are ranking: 5.94 -> 5.60 ms alone, 139.4 -> 147.3
                // frames/s in the pipeline (profiles/r04/mfma_prio_ab.txt, ab_bench_env.txt); levels 1, 2, 3 alike
                if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
                steps(0, 4);
                if (PRIO) __builtin_amdgcn_s_setprio(0);
                rank(acc, (uint32_t)(tile * HM_TM + rb * 16 + 4 * kq) + index_base);
            }
            __syncthreads();   // the next tile has landed; everybody is done with this one
        }
    }
    // a query column lives in four lanes (kq = 0..3, different rows): fold them with shuffles, lanes 0..15 write
#pragma unroll
    for (int c = 0; c < HM_NC; c++) {
        HmTop2 b = best[c];
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            const uint32_t od0 = (uint32_t)__shfl_xor((int)b.d0, off), od1 = (uint32_t)__shfl_xor((int)b.d1, off);
            const uint32_t oi0 = (uint32_t)__shfl_xor((int)b.i0, off), oi1 = (uint32_t)__shfl_xor((int)b.i1, off);
            HmTop2 m = b;   // merge two sorted pairs; equal values: lower row first
            auto ins = [&](uint32_t d, uint32_t i) {
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
            uint64_t* o = out + ((size_t)split * nq + qi) * 2;
            o[0] = b.i0 == 0xFFFFFFFFu ? HM_EMPTY : ((uint64_t)(uint32_t)((int)(qq[c] + __uint_as_float(b.d0)) - HM_BIAS) << 32) | b.i0;
            o[1] = b.i1 == 0xFFFFFFFFu ? HM_EMPTY : ((uint64_t)(uint32_t)((int)(qq[c] + __uint_as_float(b.d1)) - HM_BIAS) << 32) | b.i1;
        }
    }
}

HmPlan hm_plan(int nq, long long nt) {
    HmPlan p;
    p.q_tiles = ceil_div(nq, HM_Q);
    const int t_tiles = (int)ceil_div(nt, (long long)HM_TM);
    constexpr int SLOTS = 256 * (16 / HM_WAVES);   // two 8-wave workgroups fit a CU (LDS, registers)
    constexpr int OVERHEAD = 4;      // a workgroup's prologue and epilogue (operand loads, pipeline fill, key output) in tile times
    // Splits of the train rows: they fill the slots when the query tiles alone do not, and they set the granularity of the last round of
    // workgroups (92 query tiles x 5 splits = 460 workgroups leave a tenth of the chip idle for the whole launch; x 11 = 1012 fill two
    // rounds to 99 %). The count that minimises rounds x (tiles per split + overhead); the split lists (16 bytes per query and split) stay
    // below 256 MB. Then (APDS_MATCH_MFMA_XCD, on): a multiple of eight splits, pinned to the XCDs, when the model puts one within 5 % of
    // that count - the workgroups of an XCD then share every train tile through their L2. On the headline shape 16 splits instead of 11:
    // 1.1 GB fetched per match instead of 2.65 GB (the expanded DB is 0.25 GB). Before the threshold launch existed that cost 5 % (every
    // workgroup pays its start from +inf again: 6.5 against 6.15 ms); with it 1.7 % alone on the GPU, and in the pipeline the step is
    // 3.4 % SHORTER (138.9 against 134.3 frames/s: the extraction beside it is memory-bound) - profiles/r04/mfma_xcd_ab.txt, ab_bench_env.txt.
    const int max_splits = (int)std::max<long long>(1, std::min<long long>(std::min(t_tiles, 128), (16ll << 20) / std::max(nq, 1)));
    long long best_cost = -1;
    p.splits = 1;
    for (int sp = 1; sp <= max_splits; sp++) {
        const long long rounds = ceil_div((long long)p.q_tiles * sp, (long long)SLOTS);
        const long long cost = rounds * (ceil_div(t_tiles, sp) + OVERHEAD);
        if (best_cost < 0 || cost < best_cost) best_cost = cost, p.splits = sp;
    }
    // a multiple of eight splits (pinned to the XCDs, see the kernel) when the model puts it within 5 % of the best count
    if (config().match_mfma_xcd && t_tiles >= 64) {
        long long best8 = -1;
        int sp8 = 0;
        for (int sp = 8; sp <= std::min(max_splits, 32); sp += 8) {
            const long long rounds = ceil_div((long long)p.q_tiles * sp, (long long)SLOTS);
            const long long cost = rounds * (ceil_div(t_tiles, sp) + OVERHEAD);
            if (best8 < 0 || cost < best8) best8 = cost, sp8 = sp;
        }
        if (sp8 && best8 * 100 <= best_cost * 105) {
            p.splits = sp8;
            p.tiles_per_split = ceil_div(t_tiles, p.splits);   // (the count stays a multiple of eight: trailing splits may be empty)
            return p;
        }
    }
    if (config().match_mfma_splits > 0) p.splits = std::min(t_tiles, config().match_mfma_splits);   // (experiments)
    p.tiles_per_split = ceil_div(t_tiles, p.splits);
    p.splits = ceil_div(t_tiles, p.tiles_per_split);
    return p;
}

void hm_expand_device(const void* rows64, long long n, bool query, void* out_fp4, float* pc, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(hm_expand_rows_kernel, dim3((unsigned)ceil_div(n * 16, 256)), dim3(256), 0, s, static_cast<const uint32_t*>(rows64), n, query ? 0xCu : 0x2u,
                       query ? 0 : HM_BIAS, static_cast<uint4*>(out_fp4), pc);
    if (!query && hm_padded_rows(n) > n)
        hipLaunchKernelGGL(hm_pad_rows_kernel, dim3(1), dim3(128), 0, s, static_cast<uint4*>(out_fp4), pc, n, hm_padded_rows(n));
}
long long hm_padded_rows(long long n) { return ceil_div(n, (long long)HM_TM) * HM_TM; }
// rows of the threshold launch: whole tiles, a sixteenth of the set, at most APDS_MATCH_MFMA_SAMPLE (16 384); none below 65 536 rows
long long hm_sample_rows(long long nt) {
    const long long cap = config().match_mfma_sample;   // APDS_MATCH_MFMA_SAMPLE (0: no threshold launch)
    return (cap > 0 && nt >= 65536) ? std::min<long long>(cap, nt / 16 / HM_TM * HM_TM) / HM_TM * HM_TM : 0;   // whole tiles (HM_TM need not be a power of two)
}

// parts: [p.splits][nq][2] keys
void hm_scan_device(const void* q_fp4, const float* qpc, int nq, const void* t_fp4, const float* tpc, long long nt, const HmPlan& p, uint32_t index_base,
                    uint64_t* parts, hipStream_t s, const uint32_t* thr, bool timed) {
    // (APDS_MATCH_MFMA_LDS_PAD: unused dynamic LDS on top, an experiment knob - e.g. 30000 leaves one workgroup per CU)
    const size_t lds = (size_t)2 * HM_TM * 256 + 2 * HM_TM * sizeof(float) + (size_t)std::max(0, config().match_mfma_lds_pad);
    static std::atomic<bool> opted_dev[64];   // above the default dynamic-LDS limit: opt in once per device (idempotent, so a race is harmless)
    std::atomic<bool>& opted = opted_dev[ctx().device & 63];
    if (!opted.load()) {
        auto opt = [&](auto kernel) { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); };
        opt(&hamming_mfma_kernel<0, false>), opt(&hamming_mfma_kernel<1, false>), opt(&hamming_mfma_kernel<2, false>), opt(&hamming_mfma_kernel<3, false>);
        opt(&hamming_mfma_kernel<0, true>), opt(&hamming_mfma_kernel<1, true>), opt(&hamming_mfma_kernel<2, true>), opt(&hamming_mfma_kernel<3, true>);
        opted.store(true);
    }
    std::unique_ptr<KernelTimer> timer;   // ("hamming_topk": the name the pipeline's counters and bench.py know the main match launch by)
    if (timed) timer.reset(new KernelTimer("hamming_topk", s));
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)p.q_tiles * p.splits), dim3(64 * HM_WAVES), lds, s, static_cast<const uint4*>(t_fp4), tpc, (int)nt,
                           static_cast<const uint4*>(q_fp4), qpc, nq, p.tiles_per_split, p.q_tiles, p.splits, index_base, thr, parts);
    };
    const int prio = std::max(0, std::min(3, config().match_mfma_prio));
    if (thr) {
        switch (prio) {
            case 0: go(&hamming_mfma_kernel<0, true>); break;
            case 1: go(&hamming_mfma_kernel<1, true>); break;
            case 2: go(&hamming_mfma_kernel<2, true>); break;
            default: go(&hamming_mfma_kernel<3, true>); break;
        }
    } else {
        switch (prio) {
            case 0: go(&hamming_mfma_kernel<0, false>); break;
            case 1: go(&hamming_mfma_kernel<1, false>); break;
            case 2: go(&hamming_mfma_kernel<2, false>); break;
            default: go(&hamming_mfma_kernel<3, false>); break;
        }
    }
}

// A train set expanded once (resident databases: the pipeline's, a shard's): rows + popcounts in memory of their own.
void* hm_train_create(const void* rows64, long long n, hipStream_t s) {
    APDS_REQUIRE(rows64 && n > 0 && n < (1ll << 31), APDS_ERR_ASSERT, "an expanded train set needs rows");
    HmTrain* t = new HmTrain();
    t->device = ctx().device;
    t->src = rows64;
    t->n = n;
    if (hipMalloc(&t->rows, (size_t)hm_padded_rows(n) * 256) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&t->pc), (size_t)hm_padded_rows(n) * 4) != hipSuccess) {
        (void)hipGetLastError();
        if (t->rows) (void)hipFree(t->rows);
        delete t;
        fail(APDS_ERR_NOMEM, "no device memory for the expanded train rows");
    }
    hm_expand_device(rows64, n, false, t->rows, t->pc, s);
    HIP_CHECK(hipStreamSynchronize(s));
    return t;
}
void hm_train_destroy(void* h) {
    HmTrain* t = static_cast<HmTrain*>(h);
    if (!t) return;
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    if (t->rows) (void)hipFree(t->rows);
    if (t->pc) (void)hipFree(t->pc);
    delete t;
    if (previous >= 0) (void)hipSetDevice(previous);
}

// Top-k (k = 1 or 2) of nq 64-byte query rows against expanded train rows (t4 / tp: hm_expand_device's output, or an HmTrain's).
static void hm_topk_expanded(const void* q, int nq, const void* t4, const float* tp, long long nt, uint32_t index_base, int k, uint64_t* out, hipStream_t s) {
    ThreadCtx& c = ctx();
    void* q4 = c.alloc((size_t)nq * 256);
    float* qp = c.alloc_n<float>(nq);
    {
        KernelTimer timer("hamming_topk_sample", s);   // (the counters' name for what runs in front of the main match kernel)
        hm_expand_device(q, nq, true, q4, qp, s);
    }
    // Long train sets: a first launch over the leading rows (a sixteenth, at most 16 384: 141.5 frames/s; 32 768: 139.2; 65 536: 138.0; none:
    // 132.6 - profiles/r04/ab_bench_env.txt) gives every query a threshold, and the launch over
    // the rest starts from it - its workgroups then spend their first tiles like their last ones (a workgroup that starts from +inf runs
    // insertion code for every block of its first ~1500 rows), which is also what makes many short workgroups affordable.
    const long long sample = hm_sample_rows(nt);
    uint64_t* top2 = k == 2 ? out : c.alloc_n<uint64_t>((size_t)nq * 2);
    if (sample) {
        const HmPlan pa = hm_plan(nq, sample), pb = hm_plan(nq, nt - sample);
        uint64_t* parts_a = c.alloc_n<uint64_t>((size_t)pa.splits * nq * 2);
        uint64_t* parts_b = c.alloc_n<uint64_t>((size_t)(pb.splits + 1) * nq * 2);   // + one list: the sample's top-2
        uint64_t* top2_a = parts_b + (size_t)pb.splits * nq * 2;
        uint32_t* thr = c.alloc_n<uint32_t>(nq);
        {
            KernelTimer timer("hamming_topk_sample", s);
            hm_scan_device(q4, qp, nq, t4, tp, sample, pa, index_base, parts_a, s, nullptr, /*timed=*/false);
            if (pa.splits > 1) merge_topk_device(parts_a, pa.splits, nq, 2, top2_a, s);
            else HIP_CHECK(hipMemcpyAsync(top2_a, parts_a, (size_t)nq * 16, hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(hm_thresholds_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, (const uint64_t*)top2_a, (const float*)qp, nq, thr);
        }
        hm_scan_device(q4, qp, nq, static_cast<const char*>(t4) + (size_t)sample * 256, tp + sample, nt - sample, pb, index_base + (uint32_t)sample, parts_b, s, thr);
        merge_topk_device(parts_b, pb.splits + 1, nq, 2, top2, s);
    } else {
        const HmPlan p = hm_plan(nq, nt);
        uint64_t* parts = p.splits == 1 ? top2 : c.alloc_n<uint64_t>((size_t)p.splits * nq * 2);
        hm_scan_device(q4, qp, nq, t4, tp, nt, p, index_base, parts, s);
        if (p.splits > 1) merge_topk_device(parts, p.splits, nq, 2, top2, s);
    }
    if (k == 1) take_first_columns_device(top2, nq, 2, 1, out, s);
    HIP_CHECK(hipGetLastError());
}

// Top-k (k = 1 or 2) of nq queries over nt train rows, both 64-byte rows on the device. out: nq * k keys (distance << 32 | row + index_base).
void hamming_mfma_topk_device(const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, uint64_t* out, hipStream_t s) {
    APDS_REQUIRE(k == 1 || k == 2, APDS_ERR_ASSERT, "the matrix-core matcher serves k = 1 and k = 2");
    APDS_REQUIRE(nq > 0 && nt > 0 && nt < (1ll << 31), APDS_ERR_ASSERT, "the matrix-core matcher needs queries and train rows");
    ThreadCtx& c = ctx();
    void* t4 = c.alloc((size_t)hm_padded_rows(nt) * 256);
    float* tp = c.alloc_n<float>(hm_padded_rows(nt));
    {
        KernelTimer timer("hamming_topk_sample", s);
        hm_expand_device(t, nt, false, t4, tp, s);
    }
    hm_topk_expanded(q, nq, t4, tp, nt, index_base, k, out, s);
}

// The same against a train set expanded once (hm_train_create).
void hamming_mfma_topk_train_device(const void* q, int nq, const void* train, uint32_t index_base, int k, uint64_t* out, hipStream_t s) {
    const HmTrain* t = static_cast<const HmTrain*>(train);
    APDS_REQUIRE(t && (k == 1 || k == 2) && nq > 0, APDS_ERR_ASSERT, "the matrix-core matcher needs an expanded train set, queries and k = 1 or 2");
    hm_topk_expanded(q, nq, t->rows, t->pc, t->n, index_base, k, out, s);
}

}  // namespace apds
