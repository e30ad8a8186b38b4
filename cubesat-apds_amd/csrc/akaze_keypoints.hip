// csrc/akaze_keypoints.hip — AKAZE keypoint side on gfx950: scale-space extrema, cross-level suppression,
// sub-pixel refinement with ordered compaction, main orientation and the 486-bit M-LDB descriptor.
//
// Replaces OpenCV AKAZEFeatures::Find_Scale_Space_Extrema / Do_Subpixel_Refinement / Compute_Main_Orientation /
// MLDB_Full_Descriptor_Invoker behind /root/reference/feature_extraction/src/lib.rs:79.
//
// Output order is the reference's: level-major, then row-major (ordered compaction by prefix sums, no atomics
// in anything that decides an index). The cross-level suppression is sequential in OpenCV (each keypoint may
// delete a keypoint of the neighbouring level, which changes what later keypoints find); it is reproduced
// exactly by dependency rounds: a keypoint is processed in the first round in which no EARLIER (row-major)
// keypoint of its own level with an overlapping search window is still pending. All levels run in the same
// rounds because the passes of one phase only read snapshots of the level they iterate over.
#include <atomic>
#include <cstdlib>

#include <chrono>
#include "akaze.h"
#include "config.h"

namespace apds {

__device__ __forceinline__ int clampi2(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// ---- cross-level suppression -------------------------------------------------------------------------------
static constexpr uint8_t ST_PENDING = 255, ST_DONE_OLD = 254;

struct SuppressArgs {
    size_t bstride;   // batch: slab stride in bytes (all pointers below are image 0's)
    int n_levels;
    int phase;   // 0: compare with the previous level (ascending passes), 1: with the next level
    int w[AKAZE_MAX_LEVELS], h[AKAZE_MAX_LEVELS], sigma_size[AKAZE_MAX_LEVELS], iratio[AKAZE_MAX_LEVELS];
    const float* Ldet[AKAZE_MAX_LEVELS];
    uint8_t* mask[AKAZE_MAX_LEVELS];     // live keypoint masks (searched and cleared)
    uint8_t* status[AKAZE_MAX_LEVELS];   // snapshot of the iterated level: 0 none, 255 pending, else done stamp
    const uint32_t* list[AKAZE_MAX_LEVELS];
    const int* list_count;               // [n_levels]
    // candidates still pending after a round, three rotating buffers per level (read / append / being zeroed)
    uint32_t* pend[AKAZE_MAX_LEVELS];    // 3 * pend_cap[lvl] entries
    int pend_cap[AKAZE_MAX_LEVELS];
    int* pend_count;                     // [3][AKAZE_MAX_LEVELS] counters, one 128-byte line each (PEND_PITCH ints apart)
};
static constexpr int PEND_PITCH = 32;   // same-line atomics serialise in L2 (~12 ns each, measured): one line per counter

// Snapshot of levels [snap_lo, snap_lo + snap_n) (status = pending where the mask is set) and zeroing of the three rotating
// pending counters of levels [zero_lo, zero_lo + zero_n) (the levels whose passes are about to run). grid.y = max(snap_n, zero_n).
__global__ void suppress_init_status_kernel(SuppressArgs A, int snap_lo, int snap_n, int zero_lo, int zero_n) {
    APDS_RAISE_WAVE_PRIORITY();
    if ((int)blockIdx.y < zero_n && blockIdx.x == 0 && threadIdx.x < 3)
        bofs(A.pend_count, A.bstride)[(threadIdx.x * AKAZE_MAX_LEVELS + zero_lo + blockIdx.y) * PEND_PITCH] = 0;
    if ((int)blockIdx.y >= snap_n) return;
    const int lvl = snap_lo + blockIdx.y;
    const int cnt = bofs(A.list_count, A.bstride)[lvl];
    const uint32_t* __restrict__ list = bofs(A.list[lvl], A.bstride);
    uint8_t* __restrict__ status = bofs(A.status[lvl], A.bstride);
    const uint8_t* __restrict__ mask = bofs(A.mask[lvl], A.bstride);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const uint32_t e = list[i];
        const size_t p = (size_t)(e >> 16) * A.w[lvl] + (e & 0xFFFF);
        status[p] = mask[p] ? ST_PENDING : 0;
    }
}

// stamps 1..253 of finished candidates -> ST_DONE_OLD, so that the stamp values can be reused (every 253 rounds)
__device__ __forceinline__ void suppress_canon_level(const SuppressArgs& A, int lvl, int first, int stride) {
    const int cnt = bofs(A.list_count, A.bstride)[lvl];
    const uint32_t* __restrict__ list = bofs(A.list[lvl], A.bstride);
    uint8_t* __restrict__ status = bofs(A.status[lvl], A.bstride);
    for (int i = first; i < cnt; i += stride) {
        const uint32_t e = list[i];
        const size_t p = (size_t)(e >> 16) * A.w[lvl] + (e & 0xFFFF);
        const uint8_t s = status[p];
        if (s >= 1 && s <= 253) status[p] = ST_DONE_OLD;
    }
}

static constexpr int SUPPRESS_STAGE = 192;   // still-blocked candidates staged per block before one global append

// One round of one level's pass, executed by the `nwaves` waves of the calling block set that share (s_stage, s_n, s_base):
// wave `wave0` takes candidates wave0, wave0 + nwaves, ... of `in[0, cnt)`.
// One WAVE per candidate keypoint: the readiness window (up to 37 x 19 status bytes) and the neighbour search window
// (up to 16 x 16 mask bytes) are scanned 64 elements at a time; "first hit in row-major order" is the lowest set bit
// of the ballot of the first 64-element slab that has one. (A single thread walking these windows byte by byte took
// ~48 us per round; a frame needs ~20 rounds.) Candidates that are still blocked are appended to `out`.
__device__ __forceinline__ void suppress_round_body(const SuppressArgs& A, int lvl, int other, uint8_t stamp, const uint32_t* in, int cnt, uint32_t* out,
                                                    int* out_count, int wave0, int nwaves, uint32_t* s_stage, int* s_n, int* s_base) {
    constexpr int STAGE = SUPPRESS_STAGE;
    const size_t bstride = A.bstride;
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    const int w = A.w[lvl];
    const int lane = threadIdx.x & 63;
    uint8_t* status = bofs(A.status[lvl], bstride);
    const float* ldet_own = bofs(A.Ldet[lvl], bstride);
    const float* ldet_other = bofs(A.Ldet[other], bstride);
    uint8_t* omask_w = bofs(A.mask[other], bstride);
    // interaction distance in this level's pixels (conservative superset of "search windows overlap")
    int D, diff, radius;
    if (A.phase == 0) {
        diff = A.iratio[lvl] / A.iratio[other];
        radius = A.sigma_size[lvl] * diff;
        D = 2 * A.sigma_size[lvl];
    } else {
        diff = A.iratio[other] / A.iratio[lvl];
        radius = A.sigma_size[other];
        D = (2 * radius + 1) * diff;
    }
    const int W = 2 * D + 1, total = W * (D + 1);
    const int side = 2 * radius, total2 = side * side;
    const int ow = A.w[other], oh = A.h[other];
    const uint8_t* omask = omask_w;
    for (int i = wave0; i < cnt; i += nwaves) {
        const uint32_t e = in[i];
        const int x = e & 0xFFFF, y = e >> 16;
        const size_t p = (size_t)y * w + x;
        if (status[p] != ST_PENDING) continue;   // wave-uniform
        // The first trip of BOTH windows (four 64-element slabs each) and the candidate's own response are loaded before anything is
        // evaluated: the neighbour search of the other level does not depend on the readiness test, so its memory round trip overlaps
        // the readiness one instead of following it (a round is bound by one candidate's chain of dependent loads). A blocked
        // candidate discards the search; a ready candidate's search window cannot be touched by another candidate of the same round
        // (that is what "ready" means), so reading it early sees the same bytes. (The other order — the search first, the readiness
        // window only for the candidates that have a victim, i.e. half the loads for most candidates — was measured: 65 + 44 us for the
        // two first rounds against 57 + 41: the round is bound by the dependent round trips, not by the number of loads. Round 4 measured
        // the opposite remedy as well — two candidates per trip, their entries, status bytes and windows fetched together: 59 + 40 us,
        // no change (profiles/r04/timeline_suppress_two_per_trip.txt): 85 000 candidates x ~25 scattered 64-byte lines in 56 us is
        // 2.4 TB/s of line fetches for a few useful bytes each; the windows' footprint, not their latency, is the round.)
        const int px = A.phase == 0 ? x * diff : x / diff, py = A.phase == 0 ? y * diff : y / diff;
        uint8_t sv[4], mv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = u * 64 + lane;
            sv[u] = 0;
            if (idx < total) {
                const int ry = idx / W, rx = idx - ry * W;
                const int yy = y - D + ry, xx = x - D + rx;
                if (yy >= 0 && xx >= 0 && xx < w && (yy < y || xx < x)) sv[u] = status[(size_t)yy * w + xx];
            }
            mv[u] = 0;
            if (idx < total2) {
                const int iy = idx / side, ix = idx - iy * side;
                const int ii = py - radius + iy, jj = px - radius + ix;
                if (ii >= 0 && ii < oh && jj >= 0 && jj < ow) mv[u] = omask[(size_t)ii * ow + jj];
            }
        }
        const float own_response = ldet_own[p];
        // ready iff no EARLIER (row-major) keypoint of this level within D is pending or finished only in this round
        bool hit0 = false;
#pragma unroll
        for (int u = 0; u < 4; u++) hit0 |= sv[u] == ST_PENDING || sv[u] == stamp;
        bool blocked = __any(hit0);
        for (int base = 256; base < total && !blocked; base += 256) {
            bool hit = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = base + u * 64 + lane;
                if (idx < total) {
                    const int ry = idx / W, rx = idx - ry * W;
                    const int yy = y - D + ry, xx = x - D + rx;
                    if (yy >= 0 && xx >= 0 && xx < w && (yy < y || xx < x)) {
                        const uint8_t st = status[(size_t)yy * w + xx];
                        hit |= st == ST_PENDING || st == stamp;
                    }
                }
            }
            blocked = __any(hit);
        }
        // the candidate's victim: the first live keypoint of the other level inside its search window, row-major
        int found = -1;
        {
#pragma unroll
            for (int u = 0; u < 4; u++) {   // first hit in row-major order: lowest slab, lowest lane
                const int idx = u * 64 + lane;
                bool ok = false;
                if (mv[u]) {
                    const int iy = idx / side, ix = idx - iy * side;
                    const int dx = ix - radius, dy = iy - radius;
                    ok = dx * dx + dy * dy <= radius * radius;
                }
                const unsigned long long bb = __ballot(ok);
                if (found < 0 && bb) {
                    const int first = u * 64 + __ffsll((long long)bb) - 1;
                    const int iy = first / side, ix = first - iy * side;
                    found = (py - radius + iy) * ow + (px - radius + ix);
                }
            }
            for (int base = 256; base < total2 && found < 0; base += 256) {   // (windows wider than 16 x 16: none with AKAZE's parameters)
                unsigned long long b[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int idx = base + u * 64 + lane;
                    bool ok = false;
                    if (idx < total2) {
                        const int iy = idx / side, ix = idx - iy * side;
                        const int ii = py - radius + iy, jj = px - radius + ix;
                        if (ii >= 0 && ii < oh && jj >= 0 && jj < ow && omask[(size_t)ii * ow + jj]) {
                            const int dx = jj - px, dy = ii - py;
                            ok = dx * dx + dy * dy <= radius * radius;
                        }
                    }
                    b[u] = __ballot(ok);
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (found < 0 && b[u]) {
                        const int first = base + u * 64 + __ffsll((long long)b[u]) - 1;
                        const int iy = first / side, ix = first - iy * side;
                        found = (py - radius + iy) * ow + (px - radius + ix);
                    }
            }
        }
        if (found < 0) {
            // No live keypoint of the other level in the window, and keypoints are only ever deleted: whatever the earlier candidates
            // do, this one deletes nothing. It is finished now and, having no effect, never makes a later candidate wait.
            if (lane == 0) status[p] = ST_DONE_OLD;
            continue;
        }
        if (blocked) {
            // An earlier candidate in reach is still pending. It matters only if it can take this candidate's victim v away: victims
            // ahead of v in the window are dead for good, and a deletion behind v does not change "the first live one". So: blocked
            // iff an earlier pending (or finished-this-round) candidate e' of this level has v inside ITS search window. The scan
            // covers every position of this level whose window can contain v, and tests membership exactly as find_neighbor_point
            // does (half-open square [-r, r) and the disc).
            const int vy = found / ow, vx = found - vy * ow;
            int cx0, cx1, cy0, cy1;
            if (A.phase == 0) {   // e' = (x', y') searches around (x' * diff, y' * diff)
                cx0 = (vx - radius) / diff - 1, cx1 = (vx + radius) / diff + 1;
                cy0 = (vy - radius) / diff - 1, cy1 = (vy + radius) / diff + 1;
            } else {              // around (x' / diff, y' / diff)
                cx0 = (vx - radius) * diff, cx1 = (vx + radius + 1) * diff;
                cy0 = (vy - radius) * diff, cy1 = (vy + radius + 1) * diff;
            }
            cx0 = max(cx0, 0), cy0 = max(cy0, 0), cx1 = min(cx1, w - 1), cy1 = min(cy1, y);   // rows after y are later candidates
            const int cw = cx1 - cx0 + 1, cn = cw * (cy1 - cy0 + 1);
            bool still = false;
            for (int base = 0; base < cn && !still; base += 256) {
                bool hit = false;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int idx = base + u * 64 + lane;
                    if (idx < cn) {
                        const int ry = idx / cw, rx = idx - ry * cw;
                        const int yy = cy0 + ry, xx = cx0 + rx;
                        if (yy < y || xx < x) {
                            const uint8_t st = status[(size_t)yy * w + xx];
                            if (st == ST_PENDING || st == stamp) {
                                const int qx = A.phase == 0 ? xx * diff : xx / diff, qy = A.phase == 0 ? yy * diff : yy / diff;
                                const int dx = vx - qx, dy = vy - qy;
                                hit |= dx >= -radius && dx < radius && dy >= -radius && dy < radius && dx * dx + dy * dy <= radius * radius;
                            }
                        }
                    }
                }
                still = __any(hit);
            }
            blocked = still;
        }
        if (blocked) {
            if (lane == 0) {
                const int slot = atomicAdd(s_n, 1);
                if (slot < STAGE) s_stage[slot] = e;
                else out[atomicAdd(out_count, 1)] = e;   // staging full (dense clusters): append directly
            }
            continue;
        }
        if (lane == 0) {
            if (found >= 0 && own_response > ldet_other[found]) omask_w[found] = 0;
            status[p] = stamp;
        }
    }
    __syncthreads();
    const int staged = min(*s_n, STAGE);
    if (threadIdx.x == 0 && staged) *s_base = atomicAdd(out_count, staged);
    __syncthreads();
    for (int i = threadIdx.x; i < staged; i += blockDim.x) out[*s_base + i] = s_stage[i];
}

// Wide form of a round: the passes of levels [lvl0, lvl0 + gridDim.y) with gridDim.x blocks of four waves each. Round r reads the
// candidates that were still pending after round r-1 (in_sel: -1 = the level's full candidate list) and appends the ones that are
// still blocked to buffer out_sel; the counter of buffer zero_sel is cleared for the round after.
__global__ __launch_bounds__(256) void suppress_round_kernel(SuppressArgs A, int lvl0, uint8_t stamp, int in_sel, int out_sel, int zero_sel) {
    APDS_RAISE_WAVE_PRIORITY();
    const int lvl = lvl0 + blockIdx.y;
    const int other = A.phase == 0 ? lvl - 1 : lvl + 1;
    if (other < 0 || other >= A.n_levels) return;
    const size_t bstride = A.bstride;
    int* __restrict__ pend_count = bofs(A.pend_count, bstride);
    uint32_t* __restrict__ pend = bofs(A.pend[lvl], bstride);
    const uint32_t* __restrict__ in = in_sel < 0 ? bofs(A.list[lvl], bstride) : pend + (size_t)in_sel * A.pend_cap[lvl];
    const int cnt = in_sel < 0 ? bofs(A.list_count, bstride)[lvl] : pend_count[(in_sel * AKAZE_MAX_LEVELS + lvl) * PEND_PITCH];
    if (blockIdx.x == 0 && threadIdx.x == 0) pend_count[(zero_sel * AKAZE_MAX_LEVELS + lvl) * PEND_PITCH] = 0;
    if (cnt == 0) return;   // nothing pending for this level: the block has no work
    __shared__ uint32_t s_stage[SUPPRESS_STAGE];
    __shared__ int s_n, s_base;
    suppress_round_body(A, lvl, other, stamp, in, cnt, pend + (size_t)out_sel * A.pend_cap[lvl], &pend_count[(out_sel * AKAZE_MAX_LEVELS + lvl) * PEND_PITCH],
                        blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, s_stage, &s_n, &s_base);
}

// Tail of the passes of levels [lvl0, lvl0 + gridDim.y): ONE block per level runs the remaining rounds (from round `round0`) back to
// back until the level has no pending candidate left. After the first wide rounds only the ends of the dependency chains are left
// (tens of candidates, up to ~15 more rounds): as separate launches each of those rounds cost a launch latency plus a host check for
// convergence every few rounds; inside one block a round costs one candidate's chain of dependent loads and a barrier, and the loop
// ends exactly when the work does: no host synchronisation at all. The passes of different levels are independent within a phase
// (see the file header), so the blocks never wait for each other. Every thread leaves the loop on the same block-uniform count.
__global__ __launch_bounds__(1024) void suppress_tail_kernel(SuppressArgs A, int lvl0, int round0) {
    APDS_RAISE_WAVE_PRIORITY();
    const int lvl = lvl0 + blockIdx.y;
    const int other = A.phase == 0 ? lvl - 1 : lvl + 1;
    if (other < 0 || other >= A.n_levels) return;
    const size_t bstride = A.bstride;
    int* pend_count = bofs(A.pend_count, bstride);
    uint32_t* pend = bofs(A.pend[lvl], bstride);
    __shared__ uint32_t s_stage[SUPPRESS_STAGE];
    __shared__ int s_n, s_base, s_cnt;
    for (int round = round0;; round++) {
        const int in_sel = (round - 1) % 3, out_sel = round % 3;
        if (threadIdx.x == 0) {
            // the counters are updated by L2 atomics: read and reset them there as well, not through this CU's L1
            s_cnt = round == 0 ? bofs(A.list_count, bstride)[lvl]
                               : __hip_atomic_load(&pend_count[(in_sel * AKAZE_MAX_LEVELS + lvl) * PEND_PITCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicExch(&pend_count[(out_sel * AKAZE_MAX_LEVELS + lvl) * PEND_PITCH], 0);
        }
        __syncthreads();
        const int cnt = s_cnt;
        if (cnt == 0) break;   // block-uniform
        if (round > 0 && round % 253 == 0) {   // the stamp values come round again: retire the old ones first
            suppress_canon_level(A, lvl, threadIdx.x, blockDim.x);
            __syncthreads();
        }
        const uint32_t* in = round == 0 ? bofs(A.list[lvl], bstride) : pend + (size_t)in_sel * A.pend_cap[lvl];
        suppress_round_body(A, lvl, other, (uint8_t)(round % 253 + 1), in, cnt, pend + (size_t)out_sel * A.pend_cap[lvl],
                            &pend_count[(out_sel * AKAZE_MAX_LEVELS + lvl) * PEND_PITCH], threadIdx.x >> 6, blockDim.x >> 6, s_stage, &s_n, &s_base);
        __syncthreads();       // this round's status / mask / list writes are visible to the whole block before the next round reads them
    }
}

// ---- a1.7 sub-pixel refinement ---------------------------------------------------------------------------------
struct Refined {
    float x, y, response;
    bool ok;
};

__device__ __forceinline__ Refined refine(const float* __restrict__ ldet, int cols, int x, int y, float ratio) {
    const size_t c = (size_t)y * cols + x;
    const float Dx = 0.5f * (ldet[c + 1] - ldet[c - 1]);
    const float Dy = 0.5f * (ldet[c + cols] - ldet[c - cols]);
    const float Dxx = ldet[c + 1] + ldet[c - 1] - 2.0f * ldet[c];
    const float Dyy = ldet[c + cols] + ldet[c - cols] - 2.0f * ldet[c];
    const float Dxy = 0.25f * (ldet[c + cols + 1] + ldet[c - cols - 1] - ldet[c - cols + 1] - ldet[c + cols - 1]);
    // cv::solve(Matx22f, Vec2f, dst, DECOMP_LU): lapack.cpp's 2 x 2 CV_32F branch - determinant (`det2`) and both numerators in
    // double (products of two floats are exact there), one rounding to float per unknown
    float dx = 0.0f, dy = 0.0f;
    double det = (double)Dxx * (double)Dyy - (double)Dxy * (double)Dxy;
    if (det != 0.) {
        det = 1. / det;
        const float b0 = -Dx, b1 = -Dy;
        dx = (float)(((double)b0 * (double)Dyy - (double)b1 * (double)Dxy) * det);
        dy = (float)(((double)b1 * (double)Dxx - (double)b0 * (double)Dxy) * det);
    }
    Refined r;
    r.ok = !(fabsf(dx) > 1.0f || fabsf(dy) > 1.0f);
    r.x = x * ratio + (dx * ratio + .5f * (ratio - 1.f));
    r.y = y * ratio + (dy * ratio + .5f * (ratio - 1.f));
    r.response = ldet[c];
    return r;
}

// drop candidates whose refinement is unstable, so bit 0 of the concatenated masks becomes the final keypoint flag (the levels of a
// stage are final by now: no suppression pass reads their masks as victims any more)
__global__ void subpixel_filter_kernel(LevelTable T, const int* __restrict__ list_count, int lvl0) {
    APDS_RAISE_WAVE_PRIORITY();
    const int lvl = lvl0 + blockIdx.y;
    const int cnt = bofs(list_count, T.bstride)[lvl];
    const uint32_t* __restrict__ list = bofs(T.list[lvl], T.bstride);
    uint8_t* __restrict__ mask = bofs(T.mask[lvl], T.bstride);
    const float* __restrict__ ldet = bofs(T.Ldet[lvl], T.bstride);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const uint32_t e = list[i];
        const int x = e & 0xFFFF, y = e >> 16;
        const size_t p = (size_t)y * T.w[lvl] + x;
        if (!mask[p]) continue;
        const Refined r = refine(ldet, T.w[lvl], x, y, T.ratio[lvl]);
        if (!r.ok) mask[p] = 2;   // survived the suppression, dropped by the refinement: bit 0 (= "is a keypoint") clear, byte non-zero
    }
}

// ---- ordered compaction without passes over the masks ------------------------------------------------------------------------------
// Output order is level-major, row-major: a keypoint's position is the number of keypoints before it in the concatenated masks.
// The mask-scan path below (kp_block_counts / kp_scan_offsets / emit_keypoints) reads all the masks twice (85 MB at 4096^2) to place
// ~35 000 keypoints. Here the candidates place themselves: (1) subpixel_count_kernel, one thread per list entry: a survivor of the
// suppression is refined; if it stays it keeps its refined values next to its list entry and counts itself in the 128-byte chunk
// (`fine`) and the 128 KiB block (`coarse`, one counter per 128-byte line: same-line atomics serialise) of the masks it lies in;
// (2) kp_scan_fine_kernel, one block per coarse block: exclusive prefix of the fine counts (+ the coarse blocks before it);
// (3) emit_ranked_kernel, one thread per list entry: position = prefix of its chunk + the set flags before it inside the chunk
// (one cache line of the mask). Same keypoints, same order, same values as the mask-scan path.
static constexpr int FINE_SHIFT = 7, COARSE_SHIFT = 17, COARSE_PITCH = 32;
static constexpr uint32_t REF_DEAD = 0xFFFFFFFFu;

__global__ void subpixel_count_kernel(LevelTable T, const int* __restrict__ list_count, int lvl0, int* __restrict__ fine, int* __restrict__ coarse) {
    APDS_RAISE_WAVE_PRIORITY();
    const int lvl = lvl0 + blockIdx.y;
    const int cnt = bofs(list_count, T.bstride)[lvl];
    const uint32_t* __restrict__ list = bofs(T.list[lvl], T.bstride);
    uint8_t* __restrict__ mask = bofs(T.mask[lvl], T.bstride);
    const float* __restrict__ ldet = bofs(T.Ldet[lvl], T.bstride);
    float* __restrict__ ref = bofs(T.ref[lvl], T.bstride);
    fine = bofs(fine, T.bstride);
    coarse = bofs(coarse, T.bstride);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const uint32_t e = list[i];
        const int x = e & 0xFFFF, y = e >> 16;
        const size_t p = (size_t)y * T.w[lvl] + x;
        bool keep = mask[p] & 1;
        Refined r{};
        if (keep) {
            r = refine(ldet, T.w[lvl], x, y, T.ratio[lvl]);
            if (!r.ok) {
                mask[p] = 2;   // survived the suppression, dropped by the refinement: bit 0 (= "is a keypoint") clear, byte non-zero
                keep = false;
            }
        }
        if (!keep) {
            reinterpret_cast<uint32_t*>(ref)[3 * (size_t)i] = REF_DEAD;
            continue;
        }
        ref[3 * (size_t)i] = r.x;
        ref[3 * (size_t)i + 1] = r.y;
        ref[3 * (size_t)i + 2] = r.response;
        const long long eg = T.pix_offset[lvl] + (long long)p;
        atomicAdd(&fine[eg >> FINE_SHIFT], 1);
        atomicAdd(&coarse[(eg >> COARSE_SHIFT) * COARSE_PITCH], 1);
    }
}

// fine[i] <- kp_base[0] + (number of keypoints in the chunks before chunk i); kp_base[1] <- kp_base[0] + all keypoints
__global__ __launch_bounds__(1024) void kp_scan_fine_kernel(int* __restrict__ fine, const int* __restrict__ coarse, int n_fine, int* __restrict__ kp_base,
                                                            size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    fine = bofs(fine, bstride);
    coarse = bofs(coarse, bstride);
    kp_base = bofs(kp_base, bstride);
    __shared__ int wsum[16];
    __shared__ int s_prefix;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int part = 0;
    for (int cb = tid; cb < b; cb += 1024) part += coarse[(size_t)cb * COARSE_PITCH];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) wsum[wv] = part;
    __syncthreads();
    if (tid == 0) {
        int t = kp_base[0];
        for (int k = 0; k < 16; k++) t += wsum[k];
        s_prefix = t;
    }
    __syncthreads();
    const int prefix = s_prefix;
    const int idx = b * 1024 + tid;
    const int v = idx < n_fine ? fine[idx] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();   // wsum is reused
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int k = 0; k < wv; k++) before += wsum[k];
    if (idx < n_fine) fine[idx] = prefix + before + incl - v;
    if (b == (int)gridDim.x - 1 && tid == 1023) kp_base[1] = prefix + before + incl;
}

__global__ void emit_ranked_kernel(LevelTable T, const int* __restrict__ list_count, int lvl0, const uint8_t* __restrict__ flags,
                                   const int* __restrict__ fine_excl, apds_keypoint* __restrict__ kps, int capacity, size_t kp_bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    const int lvl = lvl0 + blockIdx.y;
    const int cnt = bofs(list_count, T.bstride)[lvl];
    const uint32_t* __restrict__ list = bofs(T.list[lvl], T.bstride);
    const float* __restrict__ ref = bofs(T.ref[lvl], T.bstride);
    flags = bofs(flags, T.bstride);
    fine_excl = bofs(fine_excl, T.bstride);
    kps = bofs(kps, kp_bstride);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const uint32_t rx = reinterpret_cast<const uint32_t*>(ref)[3 * (size_t)i];
        if (rx == REF_DEAD) continue;
        const uint32_t e = list[i];
        const int x = e & 0xFFFF, y = e >> 16;
        const long long eg = T.pix_offset[lvl] + (long long)y * T.w[lvl] + x;
        // set flags (bit 0) in [chunk, eg): the chunk is one 128-byte line of the masks
        const uint4* __restrict__ line = reinterpret_cast<const uint4*>(flags + (eg & ~127ll));
        const int nb = (int)(eg & 127);
        int pos = fine_excl[eg >> FINE_SHIFT];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint4 v = line[j];
            const int r = nb - 16 * j;   // bytes of this word group that lie before eg
            if (r <= 0) continue;
            unsigned long long lo = ((unsigned long long)v.y << 32 | v.x) & 0x0101010101010101ull;
            unsigned long long hi = ((unsigned long long)v.w << 32 | v.z) & 0x0101010101010101ull;
            if (r < 8) {
                lo &= (1ull << (8 * r)) - 1;
                hi = 0;
            } else if (r < 16) {
                hi &= r == 8 ? 0ull : (1ull << (8 * (r - 8))) - 1;
            }
            pos += __popcll(lo) + __popcll(hi);
        }
        if (pos >= capacity) continue;
        apds_keypoint kp;
        kp.x = __uint_as_float(rx);
        kp.y = ref[3 * (size_t)i + 1];
        kp.size = (T.esigma[lvl] * 1.5f) * 2.0f;
        kp.angle = 0.0f;
        kp.response = ref[3 * (size_t)i + 2];
        kp.octave = T.octave[lvl];
        kp.class_id = lvl;
        kps[pos] = kp;
    }
}

static constexpr int SCAN_BLOCK = 1024;

// flags[lo, hi) = the masks of a run of consecutive levels (a stage); block b covers the 16 KiB from (lo & ~15) + b * 16 KiB. Bytes
// outside [lo, hi) belong to other stages and read as zero.
__device__ __forceinline__ uint4 load_flags16_range(const uint8_t* __restrict__ flags, long long base, long long lo, long long hi) {
    if (base >= lo && base + 16 <= hi) return *reinterpret_cast<const uint4*>(flags + base);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; b++)
        if (base + b >= lo && base + b < hi && (flags[base + b] & 1)) w[b >> 2] |= 1u << (8 * (b & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// kp_base[0] = index of the stage's first keypoint in the image's output (the keypoints of the earlier stages come first: output
// order is level-major), written by the stage's scan; kp_base == nullptr: 0.
__global__ __launch_bounds__(SCAN_BLOCK) void emit_keypoints_kernel(LevelTable T, const uint8_t* __restrict__ flags, long long lo, long long hi,
                                                                    const int* __restrict__ block_offsets, const int* __restrict__ kp_base,
                                                                    apds_keypoint* __restrict__ kps, int capacity, size_t kp_bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    flags = bofs(flags, T.bstride);
    block_offsets = bofs(block_offsets, T.bstride);
    kps = bofs(kps, kp_bstride);
    const int first = kp_base ? bofs(kp_base, T.bstride)[0] : 0;
    __shared__ int wsum[SCAN_BLOCK / 64];
    const long long base = (lo & ~15ll) + ((long long)blockIdx.x * SCAN_BLOCK + threadIdx.x) * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (base < hi) v = load_flags16_range(flags, base, lo, hi);
    const int mine = __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
    // exclusive position of this thread's first keypoint inside the block: wave prefix (shuffles) + earlier waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (!mine) return;
    int before = 0;
    for (int k = 0; k < wv; k++) before += wsum[k];
    int pos = first + block_offsets[blockIdx.x] + before + incl - mine;
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
    for (int b = 0; b < 16; b++) {
        if (!((words[b >> 2] >> (8 * (b & 3))) & 1)) continue;
        if (pos >= capacity) return;
        const long long e = base + b;
        int lvl = 0;
        while (lvl + 1 < T.n && e >= T.pix_offset[lvl + 1]) lvl++;
        const long long pix = e - T.pix_offset[lvl];
        const int y = (int)(pix / T.w[lvl]), x = (int)(pix - (long long)y * T.w[lvl]);
        const Refined r = refine(bofs(T.Ldet[lvl], T.bstride), T.w[lvl], x, y, T.ratio[lvl]);
        apds_keypoint kp;
        kp.x = r.x;
        kp.y = r.y;
        kp.size = (T.esigma[lvl] * 1.5f) * 2.0f;
        kp.angle = 0.0f;
        kp.response = r.response;
        kp.octave = T.octave[lvl];
        kp.class_id = lvl;
        kps[pos++] = kp;
    }
}

// ---- max_points: keep the `keep` strongest (response desc, ties by detection order), in that order -------------
__global__ __launch_bounds__(256) void rank_select_kernel(const apds_keypoint* __restrict__ in, int n, int keep, apds_keypoint* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ float s_resp[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float mine = i < n ? in[i].response : 0.f;
    int rank = 0;
    for (int base = 0; base < n; base += 256) {
        const int j = base + threadIdx.x;
        s_resp[threadIdx.x] = j < n ? in[j].response : -1.f;
        __syncthreads();
        const int lim = min(256, n - base);
        for (int k = 0; k < lim; k++) {
            const float r = s_resp[k];
            rank += (r > mine) || (r == mine && base + k < i);
        }
        __syncthreads();
    }
    if (i < n && rank < keep) out[rank] = in[i];
}

// ---- a1.8 main orientation: one wave per keypoint ---------------------------------------------------------------
__constant__ float c_gauss25[7][7] = {
    {0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
    {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
    {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
    {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
    {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
    {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
    {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f}};

struct OrientTable {
    int8_t dx[109], dy[109];
};
constexpr OrientTable make_orient_table() {
    OrientTable t{};
    int k = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j < 36) {
                t.dy[k] = (int8_t)i;
                t.dx[k] = (int8_t)j;
                ++k;
            }
    return t;
}
__constant__ OrientTable c_orient = make_orient_table();

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// The per-keypoint kernels below give every wave its own slices of the block's LDS arrays: what one phase writes, only the same wave reads in
// the next. The LDS unit executes a wave's instructions in issue order, so a later read sees an earlier write without any wait; all that is
// needed between two phases is that the compiler does not move LDS accesses across the boundary. (Round 2 used __syncthreads() here: three
// block-wide rendezvous per keypoint in the descriptor kernel and seven in the orientation kernel tied four independent waves together.
// Same time either way - the descriptor kernel waits for HBM, see below - but nothing is left that needs the block to move in step.)
__device__ __forceinline__ void wave_lds_phase() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Keypoints [range[0], min(range[1], n_cap)) of this image (range: two consecutive ints of the image's slab, written by the stage's
// scan; nullptr: [0, n_cap)): the counts stay on the device, and the blocks stride over the range, so the grid need not match it.
// Which groups of four keypoints a block takes. Plain: block b takes groups b, b + gridDim.x, ... XCD-aware (xcd_ranges): workgroups are
// dealt round-robin over the 8 XCDs, so block b runs on XCD b % 8; that XCD gets ONE contiguous eighth of the groups and its blocks walk
// it in order. Keypoints are in level-major, row-major order: neighbours in the list are neighbours in the image, their sample patches
// overlap, and one XCD's L2 then serves both instead of two L2s fetching the same lines.
#define KP_GROUP_LOOP(kb, begin, n, xcd_ranges)                                                                                          \
    const int kp_groups_ = ((n) - (begin) + 3) >> 2, kp_nbx_ = ((int)gridDim.x + 7) >> 3;                                               \
    const int kp_x_ = (int)blockIdx.x & 7, kp_lo_ = (int)((long long)kp_groups_ * kp_x_ >> 3), kp_hi_ = (int)((long long)kp_groups_ * (kp_x_ + 1) >> 3); \
    const int kp_first_ = (xcd_ranges) ? kp_lo_ + ((int)blockIdx.x >> 3) : (int)blockIdx.x, kp_end_ = (xcd_ranges) ? kp_hi_ : kp_groups_; \
    const int kp_stride_ = (xcd_ranges) ? kp_nbx_ : (int)gridDim.x;                                                                       \
    for (int kg_ = kp_first_, kb = (begin) + 4 * kg_; kg_ < kp_end_; kg_ += kp_stride_, kb = (begin) + 4 * kg_)

__global__ __launch_bounds__(256) void orientation_kernel(LevelTable T, apds_keypoint* __restrict__ kps, const int* __restrict__ range, int n_cap, size_t kp_bstride,
                                                          float ang_step, int nkeys, int xcd_ranges) {
    APDS_RAISE_WAVE_PRIORITY();
    const int begin = range ? bofs(range, T.bstride)[0] : 0;
    const int n = range ? min(bofs(range, T.bstride)[1], n_cap) : n_cap;
    kps = bofs(kps, kp_bstride);
    __shared__ float s_x[4][112], s_y[4][112];
    __shared__ float s_xs[4][112], s_ys[4][112];   // the same, in sorted order
    __shared__ uint8_t s_bin[4][112];
    __shared__ int s_start[4][44];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    KP_GROUP_LOOP(kb, begin, n, xcd_ranges) {   // block-uniform trip count
    const int ki = kb + wv;
    const bool live = ki < n;
    const apds_keypoint kp = kps[live ? ki : kb];
    const int lvl = kp.class_id;
    const int w = T.w[lvl], h = T.h[lvl];
    const float ratio = T.ratio[lvl];
    const int scale = __float2int_rn(0.5f * kp.size / ratio);
    const int x0 = __float2int_rn(kp.x / ratio), y0 = __float2int_rn(kp.y / ratio);
    const float2* __restrict__ Lxy = bofs(T.Lxy[lvl], T.bstride);
    const float rad = (float)(3.14159265358979323846 / 180);
    {
        // both of a lane's samples: table entries, then the two gathers, are in flight together (one round trip each)
        int si[2], sj[2];
        float wgt[2];
        float2 d[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = min(lane + 64 * u, 108);
            si[u] = c_orient.dy[k];
            sj[u] = c_orient.dx[k];
            wgt[u] = c_gauss25[si[u] < 0 ? -si[u] : si[u]][sj[u] < 0 ? -sj[u] : sj[u]];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int y = clampi2(y0 + si[u] * scale, h), x = clampi2(x0 + sj[u] * scale, w);
            d[u] = Lxy[(size_t)y * w + x];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = lane + 64 * u;
            if (k < 109) {
                const float rx = wgt[u] * d[u].x, ry = wgt[u] * d[u].y;
                const float ang = fast_atan2_deg(ry, rx) * rad;
                int b = (int)(ang / ang_step);
                if (b < 0 || b >= nkeys) b = 0;
                s_x[wv][k] = rx;
                s_y[wv][k] = ry;
                s_bin[wv][k] = (uint8_t)b;
            }
        }
    }
    wave_lds_phase();
    // counting sort, identical to idx[--cum[b]] = i for ascending i: within a bin the larger sample index comes first. Every lane
    // holds the bins of its samples lane and lane + 64. Which samples share a lane's bin comes from six bit-sliced ballots per sample
    // set (a bin is six bits: the lanes whose bin equals mine are the AND, over the bits, of the ballot or its complement) instead
    // of one ballot pair per bin (43 iterations): the masked population counts give a sample's place inside its bin and the bin's
    // size; the bins' sizes go through LDS (every sample of a bin writes the same number) to a 42-lane prefix scan -> the bins' starts.
    {
        const int bin0 = s_bin[wv][lane];
        const int bin1 = lane + 64 < 109 ? (int)s_bin[wv][lane + 64] : 63;   // 63: no sample (equal to no bin: they are < 42)
        unsigned long long eq00 = ~0ull, eq01 = ~0ull, eq10 = ~0ull, eq11 = ~0ull;   // eqXY: the lanes of set Y whose bin equals my binX
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const unsigned long long b0 = __ballot((bin0 >> k) & 1), b1 = __ballot((bin1 >> k) & 1);
            const unsigned long long n0 = ((bin0 >> k) & 1) ? 0ull : ~0ull, n1 = ((bin1 >> k) & 1) ? 0ull : ~0ull;   // complement unless my bit is set
            eq00 &= b0 ^ n0;
            eq01 &= b1 ^ n0;
            eq10 &= b0 ^ n1;
            eq11 &= b1 ^ n1;
        }
        const unsigned long long higher = ~((2ull << lane) - 1);   // lanes above this one (none for lane 63)
        const int in0 = __popcll(eq00 & higher) + __popcll(eq01);  // same-bin samples with a larger index: every sample lane' + 64 has one
        const int in1 = __popcll(eq11 & higher);
        if (lane < 44) s_start[wv][lane] = 0;
        wave_lds_phase();
        s_start[wv][bin0] = __popcll(eq00) + __popcll(eq01);       // the bin's size (the same number from every sample of the bin)
        if (bin1 < 42) s_start[wv][bin1] = __popcll(eq10) + __popcll(eq11);
        wave_lds_phase();
        int incl = lane < 42 ? s_start[wv][lane] : 0;
        const int mine = incl;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        wave_lds_phase();
        if (lane < 43) s_start[wv][lane] = incl - mine;            // exclusive prefix: bin 42 (no sample) holds the total
        wave_lds_phase();
        // the samples' values go straight to their sorted places: the window sums below then read consecutive elements
        const int p0 = s_start[wv][bin0] + in0;
        s_xs[wv][p0] = s_x[wv][lane];
        s_ys[wv][p0] = s_y[wv][lane];
        if (bin1 < 42) {
            const int p1 = s_start[wv][bin1] + in1;
            s_xs[wv][p1] = s_x[wv][lane + 64];
            s_ys[wv][p1] = s_y[wv][lane + 64];
        }
    }
    wave_lds_phase();
    float sumX = 0.0f, sumY = 0.0f, norm = -1.0f;
    if (lane < 42) {
        const int sn = lane, win = 7, slices = 42;
        const int* st = s_start[wv];
        if (sn <= slices - win) {
            for (int i = st[sn]; i < st[sn + win]; i++) {
                sumX += s_xs[wv][i];
                sumY += s_ys[wv][i];
            }
        } else {
            const int remain = sn + win - slices;
            for (int i = st[sn]; i < st[slices]; i++) {
                sumX += s_xs[wv][i];
                sumY += s_ys[wv][i];
            }
            for (int i = st[0]; i < st[remain]; i++) {
                sumX += s_xs[wv][i];
                sumY += s_ys[wv][i];
            }
        }
        norm = sumX * sumX + sumY * sumY;
    }
    // arg max over windows, the first window wins ties (the reference only replaces on strictly greater)
    int best = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const float on = __shfl_xor(norm, off);
        const int ob = __shfl_xor(best, off);
        const float ox = __shfl_xor(sumX, off), oy = __shfl_xor(sumY, off);
        if (on > norm || (on == norm && ob < best)) {
            norm = on;
            best = ob;
            sumX = ox;
            sumY = oy;
        }
    }
    if (live && lane == 0) kps[ki].angle = fast_atan2_deg(sumY, sumX);
    wave_lds_phase();   // the next keypoint reuses the wave's LDS slices
    }
}

// ---- a1.9 M-LDB 486-bit descriptor: one wave per keypoint ----------------------------------------------------------
// deterministic double sin/cos on [0, 2pi] (Cody-Waite reduction + Taylor/Horner; same arithmetic as the oracle)
__device__ __forceinline__ void det_sincos(double a, double& s, double& c) {
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const int k = (int)(a * two_over_pi + 0.5);
    const double r = (a - k * pio2_hi) - k * pio2_lo;
    const double r2 = r * r;
    double ps = -7.6471637318198164759e-13;          // -1/15!
    ps = ps * r2 + 1.6059043836821614599e-10;        //  1/13!
    ps = ps * r2 + -2.5052108385441718775e-08;       // -1/11!
    ps = ps * r2 + 2.7557319223985890653e-06;        //  1/9!
    ps = ps * r2 + -1.9841269841269841270e-04;       // -1/7!
    ps = ps * r2 + 8.3333333333333333333e-03;        //  1/5!
    ps = ps * r2 + -1.6666666666666666667e-01;       // -1/3!
    const double sr = r + r * (r2 * ps);
    double pc = 4.7794773323873852974e-14;           //  1/16!
    pc = pc * r2 + -1.1470745597729724714e-11;       // -1/14!
    pc = pc * r2 + 2.0876756987868098979e-09;        //  1/12!
    pc = pc * r2 + -2.7557319223985890653e-07;       // -1/10!
    pc = pc * r2 + 2.4801587301587301587e-05;        //  1/8!
    pc = pc * r2 + -1.3888888888888888889e-03;       // -1/6!
    pc = pc * r2 + 4.1666666666666666667e-02;        //  1/4!
    pc = pc * r2 + -0.5;
    const double cr = 1.0 + r2 * pc;
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
}

struct MldbLut {
    uint8_t a[488], b[488];
};
constexpr MldbLut make_mldb_lut() {
    MldbLut L{};
    int dpos = 0, base = 0;
    for (int g = 0; g < 3; g++) {
        const int cnt = (g + 2) * (g + 2);
        for (int pos = 0; pos < 3; pos++)
            for (int i = 0; i < cnt; i++)
                for (int j = i + 1; j < cnt; j++) {
                    L.a[dpos] = (uint8_t)(base + 3 * i + pos);
                    L.b[dpos] = (uint8_t)(base + 3 * j + pos);
                    dpos++;
                }
        base += 3 * cnt;
    }
    return L;
}
__constant__ MldbLut c_mldb = make_mldb_lut();

// One wave per keypoint. The three grids (2x2, 3x3, 4x4 cells of 10^2, 7^2, 5^2 samples) take their samples from the same
// lattice of offsets (k, l) in [-10, 11)^2 around the keypoint -- the 2x2 and 4x4 grids use its [-10, 10)^2 part, the 3x3 grid
// all of it -- and a sample depends on (k, l) only. So the 441 lattice samples (Lt, rotated Lx/Ly) are gathered ONCE into LDS
// by all 64 lanes (they were gathered 1241 times, once per grid), then 29 lanes, one per cell of any grid, add their cell's
// samples in the reference's order (k-major, l-minor; float sums are order dependent), and the 486 comparisons are done 32
// per lane.
// What bounds it (4096^2 frame, 35 k keypoints, 237 us; profiles/r03/mldb_decomposition.txt): with every gather pointed at one cache line
// the kernel takes 92 us, without the cell sums it still takes 239, with neither 67 - the 145 us are gather misses and everything else
// hides behind them. The patches of 25 k octave-0 keypoints (42 - 63 pixels square, one sample every 2 - 3 pixels) cover most of the four
// octave-0 levels, so the kernel reads nearly all of their Lt and Lx/Ly planes (~0.9 GB) once, in scattered 128-byte lines: ~3.8 TB/s of
// HBM. Walking the lattice in the direction closest to the image's rows for the keypoint's angle (fewer lines per load) changed nothing,
// and neither did dropping the block-wide barriers: the bytes have to come from HBM whatever the order.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void mldb_kernel(LevelTable T, const apds_keypoint* __restrict__ kps, const int* __restrict__ range, int n_cap, size_t kp_bstride,
                                                   uint32_t* __restrict__ desc64, size_t desc_bstride, int xcd_ranges) {
    APDS_RAISE_WAVE_PRIORITY();
    const int begin = range ? bofs(range, T.bstride)[0] : 0;
    const int n = range ? min(bofs(range, T.bstride)[1], n_cap) : n_cap;
    kps = bofs(kps, kp_bstride);
    desc64 = bofs(desc64, desc_bstride);
    constexpr int LW = 21;                     // lattice width: offsets -10 .. 10
    // per wave: the three sampled values of every lattice point as separate planes (an invalid point holds zeros) and the validity bitmap.
    // A plane has 25 rows (+): the cell loops below run over a fixed 10 x 10 window whatever the cell's size and mask what lies outside it.
    constexpr int PL = 25 * LW + 6;            // plane pitch (odd: the three planes of a point sit in different banks); the last window read is 24 * 21 + 24
    __shared__ float s_samp[4][3 * PL];
    __shared__ uint32_t s_bits[4][16];
    __shared__ int s_val[4][88];
    __shared__ uint8_t s_lut[976];              // c_mldb: a[488] then b[488]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the comparison table goes to LDS once per block (lane-indexed reads of __constant__ data are global loads)
    for (int i = threadIdx.x; i < 244; i += 256) reinterpret_cast<uint32_t*>(s_lut)[i] = reinterpret_cast<const uint32_t*>(&c_mldb)[i];
    for (int i = lane; i < 3 * PL; i += 64) s_samp[wv][i] = 0.0f;   // the rows past the lattice stay zero
    if (lane < 16) s_bits[wv][lane] = 0;
    __syncthreads();
    // chain = (cell, component): cells 0..3 the 2x2 grid (10 x 10 samples each), 4..12 the 3x3 grid (7 x 7), 13..28 the 4x4 grid (5 x 5);
    // 87 chains: lanes take chains 0..63 in a first pass (all sizes), chains 64..86 (4x4 cells only) in a second
    auto chain_geometry = [](int chain, int& step, int& base) {
        const int cell_all = chain / 3, comp = chain - 3 * cell_all;
        const int g = cell_all < 4 ? 0 : (cell_all < 13 ? 1 : 2);
        const int cell = cell_all - (g == 0 ? 0 : (g == 1 ? 4 : 13));
        const int side = g + 2;
        step = g == 0 ? 10 : (g == 1 ? 7 : 5);
        base = comp * PL + ((cell / side) * step) * LW + (cell % side) * step;
    };
    int step1, base1, step2, base2;
    chain_geometry(lane, step1, base1);
    chain_geometry(min(64 + lane, 86), step2, base2);
    KP_GROUP_LOOP(kb, begin, n, xcd_ranges) {   // block-uniform trip count
    const int ki = kb + wv;
    const bool live = ki < n;
    const apds_keypoint kp = kps[live ? ki : kb];
    const int lvl = kp.class_id;
    const int w = T.w[lvl], h = T.h[lvl];
    const float* __restrict__ Lt = bofs(T.Lt[lvl], T.bstride);
    const float2* __restrict__ Lxy = bofs(T.Lxy[lvl], T.bstride);
    const float ratio = (float)(1 << kp.octave);
    const float scale = (float)__float2int_rn(0.5f * kp.size / ratio);
    const float xf = kp.x / ratio, yf = kp.y / ratio;
    const float angle = kp.angle * (float)(3.14159265358979323846 / 180.f);
    double sd, cd;
    det_sincos((double)angle, sd, cd);
    const float co = (float)cd, si = (float)sd;
    {
        // all of a lane's lattice gathers are issued before the first LDS store: one memory round trip per keypoint, not seven
        constexpr int NS = (LW * LW + 63) / 64;
        float2 d[NS];
        float li[NS];
        bool ok[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int sidx = min(lane + 64 * j, LW * LW - 1);
            const int k = -10 + sidx / LW, l = -10 + sidx % LW;
            const float sample_y = yf + (l * co * scale + k * si * scale);
            const float sample_x = xf + (-l * si * scale + k * co * scale);
            const int y1 = __float2int_rn(sample_y), x1 = __float2int_rn(sample_x);
            ok[j] = !(y1 < 0 || y1 >= h || x1 < 0 || x1 >= w);
            const size_t o = ok[j] ? (size_t)y1 * w + x1 : 0;
            d[j] = Lxy[o];
            li[j] = Lt[o];
        }
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int sidx = lane + 64 * j;
            const bool in = sidx < LW * LW;
            const unsigned long long m = __ballot(in && ok[j]);
            if (lane == 0) {
                s_bits[wv][2 * j] = (uint32_t)m;
                s_bits[wv][2 * j + 1] = (uint32_t)(m >> 32);
            }
            if (in) {
                float vi = 0.f, vx = 0.f, vy = 0.f;
                if (ok[j]) {
                    const float rx = d[j].x, ry = d[j].y;
                    vi = li[j];
                    vx = -rx * si + ry * co;   // rrx
                    vy = rx * co + ry * si;    // rry
                }
                s_samp[wv][sidx] = vi;
                s_samp[wv][PL + sidx] = vx;
                s_samp[wv][2 * PL + sidx] = vy;
            }
        }
    }
    wave_lds_phase();
    // Cell sums in the reference's order (row-major over the cell's samples, one running sum per value). An invalid sample contributes
    // +0: a running sum that starts at +0 is never -0 (x + y is -0 only if both are), so adding +0 never changes it — the reference
    // skips those samples. The loops run over a fixed 10 x 10 window with static LDS offsets; a chain masks what is outside its cell.
    {
        float acc1 = 0.0f, acc2 = 0.0f;
        const float* p1 = &s_samp[wv][base1];
        const float* p2 = &s_samp[wv][base2];
#pragma unroll 1
        for (int a = 0; a < 10; a++) {      // (a row at a time: fully unrolled, the compiler hoists all hundred loads and spills)
            const float* row = p1 + a * LW;
            const bool row_in = a < step1;
#pragma unroll
            for (int b = 0; b < 10; b++) {
                const float v = row[b];
                acc1 += (row_in && b < step1) ? v : 0.0f;
            }
        }
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = 0; b < 5; b++) acc2 += p2[a * LW + b];
        // the number of valid samples of a cell, from the validity bitmap (lane = cell)
        int nsamples = 0;
        if (lane < 29) {
            int st, bs;
            chain_geometry(3 * lane, st, bs);
            for (int a = 0; a < st; a++) {
                const int pos = bs + a * LW;          // (component 0: bs is the cell's first lattice index)
                const unsigned long long two = (unsigned long long)s_bits[wv][(pos >> 5) + 1] << 32 | s_bits[wv][pos >> 5];
                nsamples += __popcll((two >> (pos & 31)) & ((1ull << st) - 1));
            }
        }
        // chain c's cell count sits in lane c / 3
        const int n1 = __shfl(nsamples, lane / 3), n2 = __shfl(nsamples, min(64 + lane, 86) / 3);
        if (n1 > 0) acc1 *= 1.0f / n1;
        if (n2 > 0) acc2 *= 1.0f / n2;
        const int v1 = __float_as_int(acc1), v2 = __float_as_int(acc2);
        s_val[wv][lane] = v1 ^ (v1 < 0 ? 0x7fffffff : 0);   // CV_TOGGLE_FLT: int order == float order
        if (lane < 23) s_val[wv][64 + lane] = v2 ^ (v2 < 0 ? 0x7fffffff : 0);
    }
    wave_lds_phase();
    // 486 comparisons: lane tests bits lane, lane + 64, ...; a ballot is two words of the descriptor
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = 64 * j + lane;
        const bool bit = pos < 486 && s_val[wv][s_lut[min(pos, 487)]] > s_val[wv][s_lut[488 + min(pos, 487)]];
        const unsigned long long m = __ballot(bit);
        if (live && lane == 0) {
            desc64[(size_t)ki * 16 + 2 * j] = (uint32_t)m;      // 61 payload bytes + 3 zero bytes per 64-byte row
            desc64[(size_t)ki * 16 + 2 * j + 1] = (uint32_t)(m >> 32);
        }
    }
    wave_lds_phase();   // the next keypoint reuses the wave's LDS slices
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
namespace {

bool fed_is_prime(int n) {
    if (n <= 1) return false;
    if (n == 2 || n == 3 || n == 5 || n == 7) return true;
    if (n % 2 == 0 || n % 3 == 0 || n % 5 == 0 || n % 7 == 0) return false;
    const int upper = (int)std::sqrt((double)n + 1.0);
    for (int d = 11; d <= upper; d += 2)
        if (n % d == 0) return false;
    return true;
}

// fed_tau_by_process_time(T, 1, 0.25, reordering = true): FED step sizes for one evolution level
int fed_tau(float T, float tau_max, float* tau) {
    const int n = (int)std::ceil(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f);
    if (n <= 0) return 0;
    APDS_REQUIRE(n <= 64, APDS_ERR_INTERNAL, "FED cycle longer than expected");
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    float tauh[64];
    const float c = 1.0f / (4.0f * (float)n + 2.0f);
    const float d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) {
        const float hc = cosf((float)M_PI * (2.0f * (float)k + 1.0f) * c);
        tauh[k] = d / (hc * hc);
    }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}

GaussTaps gauss_taps(int n, double sigma) {
    double k[9], sum = 0;
    const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        k[i] = std::exp(s2 * x * x);
        sum += k[i];
    }
    GaussTaps t{};
    const int r = n / 2;
    for (int j = 0; j <= r; j++) t.k[j] = (float)(k[r + j] / sum);
    return t;
}

// INTER_AREA tap tables for one axis (<= 4 taps per destination pixel)
void area_tables(int ssize, int dsize, std::vector<int>& ofs, std::vector<float>& wgt, std::vector<int>& cnt) {
    const double scale = (double)ssize / dsize;
    ofs.assign((size_t)dsize * 4, 0);
    wgt.assign((size_t)dsize * 4, 0.f);
    cnt.assign(dsize, 0);
    for (int d = 0; d < dsize; d++) {
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = std::min(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
        s2 = std::min(s2, ssize);
        s1 = std::min(s1, s2);
        auto push = [&](int sidx, double a) {
            APDS_REQUIRE(cnt[d] < 4, APDS_ERR_INTERNAL, "area resize tap overflow");
            ofs[(size_t)d * 4 + cnt[d]] = sidx;
            wgt[(size_t)d * 4 + cnt[d]] = (float)a;
            cnt[d]++;
        };
        if (s1 - f1 > 1e-3) push(s1 - 1, (s1 - f1) / cell);
        for (int sx = s1; sx < s2; sx++) push(sx, 1.0 / cell);
        if (f2 - s2 > 1e-3) push(s2, std::min(std::min(f2 - s2, 1.0), cell) / cell);
    }
}

template <class T>
T* upload(const std::vector<T>& v, hipStream_t s) {
    T* d = ctx().alloc_n<T>(v.size());
    HIP_CHECK(hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return d;
}

}  // namespace

// Ordered compaction over the concatenated level masks: every thread owns 16 consecutive mask bytes (one 16-byte load),
// a block 16 KiB; per-block counts -> exclusive offsets (single block) -> emit.
static constexpr int KP_BYTES_PER_BLOCK = SCAN_BLOCK * 16;

__device__ __forceinline__ uint4 load_flags16(const uint8_t* __restrict__ flags, long long base, long long n) {
    if (base + 16 <= n) return *reinterpret_cast<const uint4*>(flags + base);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; b++)
        if (base + b < n && flags[base + b]) w[b >> 2] |= 1u << (8 * (b & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ int nonzero_bytes(uint32_t w) {
    // bit 0 of a mask byte = keypoint (2 = dropped by the sub-pixel refinement)
    return __popc(w & 0x01010101u);
}

__global__ __launch_bounds__(SCAN_BLOCK) void kp_block_counts_kernel(const uint8_t* __restrict__ flags, long long lo, long long hi, int* __restrict__ block_counts,
                                                                     size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    flags = bofs(flags, bstride);
    block_counts = bofs(block_counts, bstride);
    __shared__ int wsum[SCAN_BLOCK / 64];
    const long long base = (lo & ~15ll) + ((long long)blockIdx.x * SCAN_BLOCK + threadIdx.x) * 16;
    int c = 0;
    if (base < hi) {
        const uint4 v = load_flags16_range(flags, base, lo, hi);
        c = nonzero_bytes(v.x) + nonzero_bytes(v.y) + nonzero_bytes(v.z) + nonzero_bytes(v.w);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; w++) sum += wsum[w];
        block_counts[blockIdx.x] = sum;
    }
}

// block counts -> exclusive offsets inside the stage; kp_base[1] = kp_base[0] + the stage's keypoint count (kp_base[0]: the count of
// the stages before it, 0 for the first: the array starts zeroed)
__global__ __launch_bounds__(1024) void kp_scan_offsets_kernel(int* __restrict__ block_counts, int nblocks, int* __restrict__ kp_base, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    block_counts = bofs(block_counts, bstride);
    kp_base = bofs(kp_base, bstride);
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
            const int add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
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
    if (threadIdx.x == 0) kp_base[1] = kp_base[0] + carry;
}

__global__ void pack_desc61_kernel(const uint8_t* __restrict__ d64, int n, uint8_t* __restrict__ d61) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * 61) return;
    const long long r = i / 61;
    d61[i] = d64[r * 64 + (i - r * 61)];
}

void pack_desc61_device(const uint8_t* d64, int n, uint8_t* d61, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(pack_desc61_kernel, dim3(ceil_div((long long)n * 61, 256)), dim3(256), 0, s, d64, n, d61);
}

AkazeDebugRequest& akaze_debug_request() {
    static thread_local AkazeDebugRequest r;
    return r;
}

// Zero the first `bytes` (a multiple of 16) of every image's slab: one launch for the batch. (hipMemset2DAsync / hipMemcpy2DAsync take a
// slow, serialising path in the runtime: with them N host threads extracting concurrently stopped scaling, 2200 -> 880 tiles/s.)
struct ZeroRanges {
    size_t from[3], bytes[3];   // 16-byte aligned offsets into the slab; blockIdx.y picks the range
};
__global__ void zero_slab_heads_kernel(char* __restrict__ base, ZeroRanges r, size_t bstride) {
    APDS_RAISE_WAVE_PRIORITY();
    uint4* p = reinterpret_cast<uint4*>(base + (size_t)blockIdx.z * bstride + r.from[blockIdx.y]);
    const size_t n = r.bytes[blockIdx.y] / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}

// out[b * n + i] = src_b[i] for the first n ints at `src` of every image's slab
__global__ void gather_slab_ints_kernel(const int* __restrict__ src, int n, size_t bstride, int batch, int* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * batch) return;
    const int b = i / n, k = i - b * n;
    out[i] = reinterpret_cast<const int*>(reinterpret_cast<const char*>(src) + (size_t)b * bstride)[k];
}

// Bump layout of one image's workspace slab: run once with base == nullptr to size it, once more with the real base. Every plane
// starts on a 256-byte boundary.
namespace {
struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <class T>
    T* take(size_t n) {
        T* p = reinterpret_cast<T*>(base + off);
        off += (n * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};
}  // namespace

// feature_extraction/src/lib.rs:61-92 on `n_img` device images of one size (a batch goes through every kernel's grid together:
// gridDim.z = n_img; a single image is a batch of one). Image i starts `img_bstride` bytes after image i-1; its keypoints go to
// kps_out + i * capacity, its descriptors to desc64_out + i * capacity * 64, its count to counts[i] (host). Returns the largest count.
int akaze_extract_batch_device(const void* img, int n_img, size_t img_bstride, int rows, int cols, int channels, size_t stride, int max_points,
                               apds_keypoint* kps_out, uint8_t* desc64_out, int capacity, int* counts, hipStream_t s) {
    APDS_REQUIRE(img != nullptr, APDS_ERR_BAD_ARG, "null image");
    APDS_REQUIRE(n_img >= 1 && n_img <= 4096, APDS_ERR_BAD_ARG, "batch must hold 1 .. 4096 images");
    APDS_REQUIRE(channels == 1 || channels == 3 || channels == 4, APDS_ERR_ASSERT, "image must have 1, 3 or 4 channels");
    APDS_REQUIRE(rows > 2 && cols > 2, APDS_ERR_ASSERT, "image must be larger than 2x2");   // AKAZE CV_Assert(img_height > 2 && img_width > 2)
    APDS_REQUIRE(rows < 65536 && cols < 65536, APDS_ERR_ASSERT, "image side must be < 65536");
    APDS_REQUIRE(stride >= (size_t)cols * channels, APDS_ERR_ASSERT, "row stride smaller than a row");
    APDS_REQUIRE(n_img == 1 || img_bstride >= (size_t)rows * stride, APDS_ERR_ASSERT, "image stride smaller than an image");
    if (max_points <= 0) max_points = APDS_MAX_POINTS;
    ThreadCtx& c = ctx();
    KernelTimer whole("akaze_extract", s);   // whole extraction (all kernels + the count read-backs), for bench.py
    const int host_time_env = config().debug_host_time;
    const auto host_t0 = std::chrono::steady_clock::now();
    const int W = cols, H = rows, B = n_img;
    const float soffset = 1.6f, derivative_factor = 1.5f, dthreshold = 0.001f;

    // ---- evolution (Allocate_Memory_Evolution)
    std::vector<LevelDesc> ev;
    {
        int omax = 4;
        const int nsub = 4;
        const float smax = 10.0f * sqrtf(2.0f);
        int lw = W, lh = H, power = 1;
        for (int i = 0; i < omax; i++) {
            for (int j = 0; j < nsub; j++) {
                LevelDesc d{};
                d.w = lw;
                d.h = lh;
                d.esigma = soffset * powf(2.f, (float)j / (float)nsub + i);
                d.sigma_size = (int)lrintf(d.esigma * derivative_factor / power);
                d.etime = 0.5f * (d.esigma * d.esigma);
                d.octave = i;
                d.sublevel = j;
                d.ratio = (float)power;
                d.border = (int)lrintf(smax * d.sigma_size) + 1;
                ev.push_back(d);
            }
            power <<= 1;
            lh >>= 1;
            lw >>= 1;
            if (lw < 80 || lh < 40) break;
        }
        for (size_t i = 1; i < ev.size(); i++) ev[i].nsteps = fed_tau(ev[i].etime - ev[i - 1].etime, 0.25f, ev[i].tau);
    }
    const int L = (int)ev.size();
    const int n_oct = ev.back().octave + 1;
    // The Hessian kernels on a side stream shorten a LONE caller's extraction (4096^2: 2.19 -> 2.08 ms, 512^2: 0.48 -> 0.45), but
    // when several host threads extract at once (the reference's rayon pool, main.rs:233-243) extra streams make the threads'
    // streams share the few hardware queues and the threads serialise each other — even an idle side stream shifts the mapping:
    // 4 threads reach 2400 tiles/s of 1024^2 when no thread ever forked, 1240 - 1320 when some did. So a thread forks only while it
    // is the ONLY host thread holding a library context (apds_thread_release drops one); APDS_AKAZE_FORK = 0 never, 2 always.
    const int fork_env = config().akaze_fork;
    // (A call costs the host ~3.5 us per launch, event record or stream wait, ~90 of them: a 512^2 tile's 0.32 ms is mostly that. Not
    // forking below 0.5 Mpx saves 30 of those calls and was measured both ways: 0.355 against 0.367 ms in tools/ab_probe.py, 0.36 against
    // 0.32 in tools/extract_probe.py — no threshold.)
    const bool fork_doh = fork_env == 2 || (fork_env == 1 && live_contexts().load() <= 1);

    // ---- one image's workspace slab (all images of the batch: the same layout, `slab` bytes apart)
    const size_t n0 = (size_t)W * H;
    long long total_pix = 0;
    for (auto& e : ev) {
        e.pix_offset = total_pix;
        total_pix += (long long)e.w * e.h;
    }
    const int nblocks = ceil_div(total_pix, KP_BYTES_PER_BLOCK);
    const int n_fine = (int)((total_pix >> FINE_SHIFT) + 1), n_coarse = (int)((total_pix >> COARSE_SHIFT) + 1);
    // zero-initialised head of the slab (one 2-D memset clears it for the whole batch): counters, then keypoint masks and statuses
    int *list_count, *hist, *pend_count, *block_counts, *kp_base, *fine_counts, *coarse_counts;
    unsigned int* hmax_bits;
    float *k_oct, *gray, *tmpS, *tmpF, *tmpP, *tmpH;
    uint8_t *mask_all, *status_all;
    std::vector<uint32_t*> lists(L), pend(L);
    std::vector<float*> lsm(L);
    size_t zero_bytes = 0, mask_off = 0, status_off = 0;
    auto layout = [&](Arena& A) {
        list_count = A.take<int>(AKAZE_MAX_LEVELS);
        hmax_bits = A.take<unsigned int>(1);
        hist = A.take<int>(304);                  // 300 bins + the ticket counter of kcontrast_hist_kernel's last block
        pend_count = A.take<int>(3 * AKAZE_MAX_LEVELS * PEND_PITCH);
        kp_base = A.take<int>(8);                 // kp_base[k] = keypoints of the stages before stage k (kp_base[0] stays 0)
        fine_counts = A.take<int>(n_fine + 1024);      // ranked compaction: keypoints per 128-byte chunk of the masks (then their prefix)
        coarse_counts = A.take<int>((size_t)n_coarse * COARSE_PITCH);
        mask_off = A.off;
        mask_all = A.take<uint8_t>((size_t)total_pix + 128);   // (+ a line: the last chunk is read whole)
        status_off = A.off;
        status_all = A.take<uint8_t>((size_t)total_pix);
        zero_bytes = A.off;
        k_oct = A.take<float>(8);
        block_counts = A.take<int>(nblocks + 4);
        gray = A.take<float>(n0);
        tmpS = A.take<float>(n0);
        tmpF = A.take<float>(n0);
        tmpP = A.take<float>(n0);
        tmpH = A.take<float>(n0 / 4 + 64);        // the next octave's start image when the last level of an octave writes it itself
        for (int i = 0; i < L; i++) {
            LevelDesc& e = ev[i];
            const size_t n = (size_t)e.w * e.h;
            e.Lt = A.take<float>(n);
            e.Lxy = A.take<float2>(n);
            e.Ldet = A.take<float>(n);
            const size_t ncand = (size_t)((e.w + 1) / 2) * ((e.h + 1) / 2);   // strict 3x3 maxima are never adjacent
            lists[i] = A.take<uint32_t>(ncand);
            pend[i] = A.take<uint32_t>(3 * ncand);
            // Lsmooth: a plane per level when the Hessian kernels run on the side stream (see below), else one shared plane
            lsm[i] = (fork_doh && i > 0) ? A.take<float>(n) : tmpS;
        }
    };
    Arena sizing;
    layout(sizing);
    const size_t slab = (sizing.off + 4095) & ~(size_t)4095;
    Arena real;
    real.base = static_cast<char*>(c.alloc(slab * (size_t)B));
    layout(real);
    Batch bt;
    bt.n = B;
    bt.stride = slab;
    bt.img_stride = img_bstride;
    if (c.fork_open) {   // an earlier call failed between fork and join (whether or not THIS call forks): its side-stream kernels may still use the workspace
        if (c.side) HIP_CHECK(hipStreamSynchronize(c.side));
        for (hipStream_t st : c.side_pool)
            if (st) HIP_CHECK(hipStreamSynchronize(st));
        c.fork_open = false;
    }
    // The streaming Hessian kernel (akaze_doh_strips.hip) writes the keypoint-mask byte and the suppression-status byte of EVERY pixel of
    // its level, so those levels - the large ones, a prefix of the level list - need no clearing: what is zeroed is the counters at the head
    // of the slab and the masks / statuses of the remaining (small) levels. (Round 2 cleared all of it: 181 MB per 4096^2 frame.)
    int n_strip_levels = 0;
    while (n_strip_levels < L && doh_strips_eligible(ev[n_strip_levels].w, ev[n_strip_levels].h, ev[n_strip_levels].sigma_size, B)) n_strip_levels++;
    {
        const size_t first = n_strip_levels < L ? (size_t)ev[n_strip_levels].pix_offset : (size_t)total_pix;
        ZeroRanges zr{};
        zr.from[0] = 0;
        zr.bytes[0] = mask_off;                                                   // counters (all planes start on 256-byte boundaries)
        zr.from[1] = mask_off + (first & ~(size_t)15);
        zr.bytes[1] = (status_off - zr.from[1]) & ~(size_t)15;                    // masks of the remaining levels + the padding line behind the last one
        zr.from[2] = status_off + (first & ~(size_t)15);
        zr.bytes[2] = (zero_bytes - zr.from[2]) & ~(size_t)15;
        const size_t most = std::max(zr.bytes[0], std::max(zr.bytes[1], zr.bytes[2]));
        // (the runtime's fill kernel reaches 1.7 TB/s; 16-byte stores from a wide grid are quicker)
        hipLaunchKernelGGL(zero_slab_heads_kernel, dim3((unsigned)std::min<size_t>(B > 1 ? 1024 : 4096, (most / 16 + 255) / 256), 3, B), dim3(256), 0, s,
                           real.base, zr, slab);
    }
    int* counts_dev = B > 1 ? c.alloc_n<int>(B) : nullptr;

    // ---- a1.1 / a1.2 / a1.3
    const GaussTaps g16 = gauss_taps(9, (double)soffset), g10 = gauss_taps(5, 1.0);
    bool early_forked = false;
    if (launch_base_strips(img, H, W, channels, stride, g16, g10, ev[0].Lt, tmpF, hmax_bits, L > 1, s, bt)) {   // large images: one fused pass
        // APDS_EARLY_FORK=1: level 0's Hessian kernel needs Lt[0] only and may start here, beside the contrast-factor pass. Measured three
        // times in round 4 (profiles/r04/ab_env_half_fuse.txt: 1.649 / 1.656 / 1.642 against 1.649 / 1.636 / 1.636 ms): the histogram kernel
        // the level chain waits for shares the machine with a kernel nothing waits for, and what the Hessian stream gains at the front it
        // has no use for at the back (its kernels follow the level chain from the second octave on). Off by default.
        if (fork_doh && L > 1 && config().early_fork) {
            HIP_CHECK(hipEventRecord(c.fork_event(0), s));
            early_forked = true;
        }
        if (L > 1) launch_kcontrast(nullptr, tmpF, W, H, hmax_bits, hist, k_oct, n_oct, s, bt, /*gradient_done=*/true);
    } else {
        launch_gray(img, H, W, channels, stride, gray, s, bt);
        launch_gauss(gray, ev[0].Lt, W, H, g16, 4, s, bt);   // Lt[0] == Lsmooth[0]
        if (L > 1) {
            launch_gauss(gray, tmpS, W, H, g10, 2, s, bt);
            launch_kcontrast(tmpS, tmpF, W, H, hmax_bits, hist, k_oct, n_oct, s, bt);
        }
    }
    auto deriv_weights = [](int sc, float& kside, float& kmid) {
        if (sc == 1) {
            kside = 3.0f / 32.0f;
            kmid = 10.0f / 32.0f;
        } else {
            const float wgt = 10.0f / 3.0f;
            const float norm = 1.0f / (2.0f * sc * (wgt + 2.0f));
            kside = norm;
            kmid = wgt * norm;
        }
    };
    // The Hessian / extrema kernel of a level hangs off the main chain (it only needs the level's Lsmooth and nothing waits for it
    // before the suppression): it goes to a second stream, so the latency-bound launches of the small octaves overlap the
    // smoothing and diffusion launches of the levels that follow. Lsmooth then needs a plane per level instead of a shared one.
    hipStream_t s_doh = fork_doh ? side_stream_beside(s) : s;
    // (Round 3, measured and removed: holding the first octave's four big Hessian kernels back until the level chain has left that octave, on
    // a stream of their own, so that they run under the latency-bound chains of the later octaves instead of beside the first octave's
    // bandwidth-bound ones: 1.828 against 1.807 ms at 4096^2 - they are off the critical path either way, which is the level chain
    // followed by the keypoint tail, and beside the small octaves' kernels they delay those. Lowest priority for the side streams: no
    // difference. profiles/r03/doh_ab.txt)
    // (a stream of their own for the small levels' Hessian kernels, which queue up behind the large levels' on this one: measured, no gain)
    // Keypoint stage: once every level has its Hessian, ONE stage makes all levels final and emits them (suppress_stage / emit_stage below
    // take a level range because round 2 also ran the large octaves' levels early, on a third stream under the small octaves' chain -
    // "staged" mode: 2.14 against 2.18 ms stand-alone then, 1.91 against 1.83 in round 3, and slower inside the streamed pipeline: the
    // per-keypoint kernels fill every CU and the chain's blocks wait for room whatever the priorities or occupancy caps; removed).
    hipStream_t s_kp = s;
    // ---- level tables for the keypoint kernels (pointers of image 0; kernels add blockIdx.z * slab)
    LevelTable T{};
    SuppressArgs A{};
    T.n = A.n_levels = L;
    T.bstride = A.bstride = slab;
    for (int i = 0; i < L; i++) {
        const LevelDesc& e = ev[i];
        T.w[i] = A.w[i] = e.w;
        T.h[i] = A.h[i] = e.h;
        T.octave[i] = e.octave;
        T.sigma_size[i] = A.sigma_size[i] = e.sigma_size;
        T.border[i] = e.border;
        T.esigma[i] = e.esigma;
        T.ratio[i] = e.ratio;
        A.iratio[i] = (int)e.ratio;
        T.pix_offset[i] = e.pix_offset;
        T.Lt[i] = e.Lt;
        T.Lxy[i] = e.Lxy;
        T.Ldet[i] = A.Ldet[i] = e.Ldet;
        T.mask[i] = A.mask[i] = mask_all + e.pix_offset;
        A.status[i] = status_all + e.pix_offset;
        T.list[i] = A.list[i] = lists[i];
        T.ref[i] = reinterpret_cast<float*>(pend[i]);   // the pending lists are idle once the suppression passes are done
        A.pend_cap[i] = ((e.w + 1) / 2) * ((e.h + 1) / 2);
        A.pend[i] = pend[i];
    }
    A.pend_count = pend_count;
    T.pix_offset[L] = total_pix;
    A.list_count = list_count;
    const size_t kp_bstride = (size_t)capacity * sizeof(apds_keypoint), desc_bstride = (size_t)capacity * 64;
    const float ang_step = (float)(2.0 * M_PI / 42);
    const int nkeys = (int)((float)(2.0 * M_PI) / ang_step);
    // per-keypoint kernels stride over the stage's keypoints: the grid only has to be of the right order (this thread's last image)
    const int kp_blocks = (std::min(16384, std::max(256, ceil_div((long long)(c.akaze_kp_estimate > 0 ? c.akaze_kp_estimate : 32768) * 5 / 4, 4))) + 7) & ~7;
    const int xcd_ranges = config().kp_xcd ? 1 : 0;

    // Stage: make levels (prev_m, m] final and emit them. Needs the Hessian kernels of levels <= D = min(m + 1, L - 1).
    //   phase 0 (a level's candidates delete weaker neighbours in the level below): passes i in (prev_D, D]
    //   phase 1 (... in the level above): snapshot of levels (prev_m, m] taken after their phase-0 state is final and BEFORE pass i-1
    //   deletes in them, then passes i in [max(prev_m, 0), min(m, L-1) - 1]; pass prev_m uses the snapshot the previous stage took.
    // Level j is final after phase-0 pass j+1 and phase-1 pass j-1; both are in this stage or an earlier one for j <= m. Each pass:
    // a snapshot / counter-reset launch, WIDE_ROUNDS wide rounds (most candidates are ready at once), then suppress_tail_kernel
    // finishes the chains without any host check.
    // (fewer wide rounds for small tiles — whose calls are bound by the host's enqueue time — were measured: the tail kernel's one block
    // per level then walks every candidate, 512^2 0.32 -> 0.37 ms)
    constexpr int WIDE_ROUNDS = 3;
    int n_stage = 0;
    auto suppress_stage = [&](int prev_m, int m, hipStream_t s_kp) {
        const int D = std::min(m + 1, L - 1), prev_D = prev_m < 0 ? 0 : std::min(prev_m + 1, L - 1);
        const dim3 lblock(256);
        auto run_passes = [&](int phase, int lo, int hi, int snap_lo, int snap_hi) {
            const int np = hi - lo + 1, ns = snap_hi - snap_lo + 1;
            if (np <= 0 && ns <= 0) return;
            A.phase = phase;
            hipLaunchKernelGGL(suppress_init_status_kernel, dim3(B > 1 ? 64 : 256, std::max(np, ns), B), lblock, 0, s_kp, A, snap_lo, std::max(ns, 0), lo,
                               std::max(np, 0));
            if (np <= 0) return;
            for (int round = 0; round < WIDE_ROUNDS; round++)
                // (a wider grid for the first round, which visits every candidate, is slower: 512 blocks 59 us, 1024 blocks 69 us against 51 us
                // with 256: more waves in flight do not help, and neither do fewer loads per candidate — see suppress_round_body)
                hipLaunchKernelGGL(suppress_round_kernel, dim3(B > 1 ? 64 : 256, np, B), lblock, 0, s_kp, A, lo, (uint8_t)(round % 253 + 1),
                                   round == 0 ? -1 : (round - 1) % 3, round % 3, (round + 1) % 3);
            hipLaunchKernelGGL(suppress_tail_kernel, dim3(1, np, B), dim3(1024), 0, s_kp, A, lo, WIDE_ROUNDS);
        };
        if (L > 1) {
            run_passes(0, prev_D + 1, D, prev_D + 1, D);                                   // phase 0: each pass snapshots its own level
            run_passes(1, std::max(prev_m, 0), std::min(m, L - 1) - 1, prev_m + 1, m);      // phase 1
        }
    };
    const bool early_count = config().early_count != 0 && !akaze_debug_request().armed;
    int* K_host = nullptr;   // set by the stage that copies the count early
    // sub-pixel filter, ordered compaction, orientation and descriptors of levels (prev_m, m], whose suppression passes are done
    auto emit_stage = [&](int prev_m, int m, hipStream_t s_kp) {
        const int a = prev_m + 1;
        int* base_k = kp_base + n_stage;
        const int ranked_env = config().kp_ranked;
        if (ranked_env && prev_m < 0 && m == L - 1) {
            // all levels in one stage: the candidates count and place themselves (no pass over the masks)
            const dim3 cgrid(B > 1 ? 16 : 128, L, B);
            hipLaunchKernelGGL(subpixel_count_kernel, cgrid, dim3(256), 0, s_kp, T, (const int*)list_count, 0, fine_counts, coarse_counts);
            hipLaunchKernelGGL(kp_scan_fine_kernel, dim3(ceil_div(n_fine, 1024), 1, B), dim3(1024), 0, s_kp, fine_counts, (const int*)coarse_counts, n_fine,
                               base_k, slab);
            hipLaunchKernelGGL(emit_ranked_kernel, cgrid, dim3(256), 0, s_kp, T, (const int*)list_count, 0, (const uint8_t*)mask_all, (const int*)fine_counts,
                               kps_out, capacity, kp_bstride);
        } else {
            hipLaunchKernelGGL(subpixel_filter_kernel, dim3(B > 1 ? 16 : 64, m - a + 1, B), dim3(256), 0, s_kp, T, (const int*)list_count, a);
            const long long lo = T.pix_offset[a], hi = T.pix_offset[m + 1];
            const int nb = ceil_div(hi - (lo & ~15ll), KP_BYTES_PER_BLOCK);
            hipLaunchKernelGGL(kp_block_counts_kernel, dim3(nb, 1, B), dim3(SCAN_BLOCK), 0, s_kp, (const uint8_t*)mask_all, lo, hi, block_counts, slab);
            hipLaunchKernelGGL(kp_scan_offsets_kernel, dim3(1, 1, B), dim3(1024), 0, s_kp, block_counts, nb, base_k, slab);
            hipLaunchKernelGGL(emit_keypoints_kernel, dim3(nb, 1, B), dim3(SCAN_BLOCK), 0, s_kp, T, (const uint8_t*)mask_all, lo, hi,
                               (const int*)block_counts, (const int*)base_k, kps_out, capacity, kp_bstride);
        }
        // The image's keypoint count (kp_base[n_stage + 1]) is final here, before orientation and descriptors: its copy to the host goes
        // out now, so that the call can return ~0.3 ms before the stream is idle (early_count below).
        if (early_count && m == L - 1) {
            K_host = c.pinned_ints(B);
            if (B == 1) {
                HIP_CHECK(hipMemcpyAsync(K_host, base_k + 1, sizeof(int), hipMemcpyDeviceToHost, s_kp));
            } else {
                hipLaunchKernelGGL(gather_slab_ints_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s_kp, (const int*)(base_k + 1), 1, slab, B, counts_dev);
                HIP_CHECK(hipMemcpyAsync(K_host, counts_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s_kp));
            }
            if (!c.count_event) HIP_CHECK(hipEventCreateWithFlags(&c.count_event, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(c.count_event, s_kp));
        }
        // ---- a1.8 / a1.9 over the stage's keypoints [kp_base[k], kp_base[k + 1]), capped at the output capacity
        hipLaunchKernelGGL(orientation_kernel, dim3(kp_blocks, 1, B), dim3(256), 0, s_kp, T, kps_out, (const int*)base_k, std::min(capacity, max_points),
                           kp_bstride, ang_step, nkeys, xcd_ranges);
        hipLaunchKernelGGL(mldb_kernel, dim3(kp_blocks, 1, B), dim3(256), 0, s_kp, T, (const apds_keypoint*)kps_out, (const int*)base_k,
                           std::min(capacity, max_points), kp_bstride, reinterpret_cast<uint32_t*>(desc64_out), desc_bstride, xcd_ranges);
        n_stage++;
    };
    auto run_stage = [&](int prev_m, int m) {
        suppress_stage(prev_m, m, s_kp);
        emit_stage(prev_m, m, s_kp);
    };
    // (Running only the suppression passes of the large octaves' levels early, on a third stream under the small-octave chain, was built
    // and measured as well: bit-identical, 1.844 against 1.825 ms — the passes' scattered loads slow the chain by more than they hide.)
    // ---- a1.4 / a1.5 per level: Lsmooth -> (Lx, Ly, Ldet) and flow; FED steps ping-pong into Lt[i]
    // Round 4: the launch that finishes the last level of an octave also writes the next octave's start image (the 2 x 2 area means of its
    // Lt, into tmpH) when its kernel family can (level_stream, nld_strip, level_fused): half_sample_kernel - 40 + 14 + 5 us of passes over
    // finished planes on the critical path of a 4096^2 frame - then does not run. APDS_HALF_FUSE=0: always the separate kernel.
    const float* fused_start = nullptr;   // set by level i - 1 when it wrote level i's start image
    // The fork to the Hessian stream. In a chain of short kernels an event recorded on the main stream costs 3.4 us each time (the marker
    // packet sits between two dependent kernels: tools/probes/fork_probe.hip, profiles/r04/fork_probe.txt - 16 forks: 236 us against 174 for
    // the same kernels without any dependency), and under rocprofv3 the timeline shows 5 us gaps behind every fork. APDS_FLAG_FORK=1 (round
    // 4): the NEXT kernel of the chain stores a sequence number as its first act (APDS_FORK_SIGNAL; it starts when its predecessor is done),
    // and the Hessian stream waits for that value (hipStreamWaitValue32 on 8 bytes of signal memory - a one-thread polling kernel of the
    // runtime): nothing sits between the chain's kernels (189 us in the probe). The wait is queued only AFTER the kernel that carries the
    // value - every wait depends on something submitted earlier, as with events, so no cycle of blocked hardware queues can form - which is
    // why a level whose fork comes after its last kernel hands its Hessian launch to the next level (deferred). A launcher that cannot carry
    // a signal leaves it armed: a one-thread kernel stores it then. MEASURED on the real frame: 1.635 against 1.641 ms over nine same-box
    // rounds (profiles/r04/ab_env_flag_fork.txt) - the chain's kernels are long and the marker is processed under the tail of the one before
    // it; the 42 us the profiled timeline promises (profiles/r04/timeline_flag_fork.txt) are the profiler's. Bit-identical, covered by
    // tests/test_strip_kernels_gpu.py; off by default: 6 us do not pay for a polling kernel per fork.
    const bool flag_fork = fork_doh && config().flag_fork && c.fork_flag_ready();
    c.fork_pending = ForkSignal{};
    if (flag_fork && c.fork_seq > 0x7FFF0000u) {   // (every earlier wait was joined: start the sequence again)
        launch_fork_signal(ForkSignal{c.fork_flag, 0}, s);
        c.fork_seq = 0;
    }
    struct DeferredHessian {
        int level;
        const float* smooth;
        float kside, kmid;
        unsigned value;
    };
    DeferredHessian deferred{-1, nullptr, 0, 0, 0};
    unsigned fork_value = 0;
    // the determinant plane of a level is stored whole only when somebody asked to see it (apds_akaze_debug_plane)
    const bool dense_det = akaze_debug_request().armed;
    auto launch_hessian = [&](const DeferredHessian& d, hipStream_t st) {
        const LevelDesc& le = ev[d.level];
        if (!(d.level < n_strip_levels &&
              launch_doh_strips(d.smooth, le.Lxy, le.Ldet, le.w, le.h, le.sigma_size, d.kside, d.kmid, le.border, dthreshold, mask_all + le.pix_offset,
                                status_all + le.pix_offset, lists[d.level], list_count + d.level, st, bt, dense_det)))
            launch_doh_fused(d.smooth, le.Lxy, le.Ldet, le.w, le.h, le.sigma_size, d.kside, d.kmid, le.border, dthreshold, mask_all + le.pix_offset,
                             lists[d.level], list_count + d.level, st, bt);
    };
    auto flush_deferred = [&]() {
        if (deferred.level < 0) return;
        launch_fork_signal(c.take_fork_signal(), s);   // (nothing if a chain kernel has taken it)
        HIP_CHECK(hipStreamWaitValue32(s_doh, c.fork_flag, deferred.value, hipStreamWaitValueGte, 0xFFFFFFFFu));
        launch_hessian(deferred, s_doh);
        deferred.level = -1;
    };
    for (int i = 0; i < L; i++) {
        LevelDesc& e = ev[i];
        const float* smooth;
        // does the NEXT level start an octave from exactly half this level's size (the 2 x 2 mean; odd sizes take the general area resize)?
        const bool want_half = config().half_fuse && i > 0 && i + 1 < L && ev[i + 1].octave > e.octave && e.w == 2 * ev[i + 1].w && e.h == 2 * ev[i + 1].h;
        bool did_half = false;
        const float* my_start = fused_start;
        fused_start = nullptr;
        if (i == 0) {
            smooth = e.Lt;
        } else {
            const LevelDesc& p = ev[i - 1];
            const float* P;   // the level's starting image
            // FED steps are issued in fused groups of up to `fuse` steps (temporal blocking in LDS): `launches` passes ping-pong
            // between e.Lt and tmpP and must end in e.Lt. Deeper fusion for the small octaves, whose launches are latency-bound.
            const int fuse = (size_t)e.w * e.h * B <= (size_t)1 << 20 ? 8 : 4;
            // small levels: Lsmooth, conductivity and the first (usually all) FED steps in ONE launch (level_fused_kernel)
            const int level_fuse = config().level_fuse;
            const bool fused_level = e.nsteps > 0 && level_fuse && ((size_t)e.w * e.h * B <= (size_t)1 << 20 || level_fuse == 2);
            const int head = fused_level ? std::min(e.nsteps, level_fused_max_steps()) : 0;
            const int launches = (e.nsteps - head + fuse - 1) / fuse + (fused_level ? 1 : 0);
            if (e.octave > p.octave && my_start) {
                P = my_start;   // written by the previous level's last launch
            } else if (e.octave > p.octave) {
                float* dstP = (launches % 2 == 0) ? e.Lt : tmpP;   // so that the last pass lands in e.Lt
                if (p.w == 2 * e.w && p.h == 2 * e.h) {
                    launch_half_sample(p.Lt, p.w, dstP, e.w, e.h, s, bt);
                } else {
                    std::vector<int> xo, yo, xc, yc;
                    std::vector<float> xw, yw;
                    area_tables(p.w, e.w, xo, xw, xc);
                    area_tables(p.h, e.h, yo, yw, yc);
                    HIP_CHECK(hipStreamSynchronize(s));   // host tables must outlive the async copies
                    launch_area_resize(p.Lt, p.w, dstP, e.w, e.h, upload(xo, s), upload(xw, s), upload(xc, s), upload(yo, s), upload(yw, s), upload(yc, s), s, bt);
                    HIP_CHECK(hipStreamSynchronize(s));
                }
                P = dstP;
            } else {
                P = p.Lt;
            }
            const float* in = P;
            int pass = 0, k = 0;
            float st[32];
            // the last level's Hessian kernel is on the critical path (nothing follows to hide it): its Lsmooth comes from a separate
            // smoothing pass, so that it runs beside the level's FED steps
            smooth = lsm[i];
            auto fork_here = [&]() {   // Lsmooth of this level exists from here on
                if (!fork_doh) return;
                if (flag_fork) {
                    flush_deferred();   // the kernel just launched carried the previous level's signal
                    fork_value = c.arm_fork_signal().value;
                    return;
                }
                HIP_CHECK(hipEventRecord(c.fork_event(i), s));
                HIP_CHECK(hipStreamWaitEvent(s_doh, c.fork_event(i), 0));
            };
            // large levels: the smoothing pass and the first group of FED steps in one pass over register strips (level_strip_kernel)
            const int level_strip = config().level_strip;
            bool strip_done = false;
            // (every level of at least 1 Mpx. On the 16 Mpx levels of a 4096^2 frame the gain is within the box-to-box noise: the Hessian
            // kernel beside them is VALU-bound and the strips' recomputed halos cost issue slots — 1.89 against 1.93 ms in one run, 1.85
            // against 1.83 in another; 2048^2: 0.79 against 0.82. APDS_LEVEL_STRIP=2: every level whatever its size, 0: never)
            const size_t lpx = (size_t)e.w * e.h * B;
            if (!fused_level && e.nsteps > 0 && level_strip && (level_strip == 2 || lpx >= (size_t)1 << 20)) {
                const int g = (e.nsteps + launches - 1) / launches;
                float* out = ((launches - 1) % 2 == 0) ? e.Lt : tmpP;
                for (int j = 0; j < g; j++) st[j] = e.tau[j] * 0.5f;
                const bool last_here = want_half && g == e.nsteps;   // this launch finishes the level
                bool ran = false;
                if (g <= 4) {
                    if (launch_level_stream(P, lsm[i], launches > 1 ? tmpF : nullptr, out, e.w, e.h, g10, k_oct + e.octave, st, g, s, bt, last_here ? tmpH : nullptr)) {
                        ran = true;
                        did_half = last_here;
                    } else {
                        ran = launch_level_strips(P, lsm[i], launches > 1 ? tmpF : nullptr, out, e.w, e.h, g10, k_oct + e.octave, st, g, s, bt);
                    }
                }
                if (ran) {
                    strip_done = true;
                    k = g;
                    in = out;
                    pass = 1;
                    fork_here();
                }
            }
            const bool smooth_first = !strip_done && (!fused_level || (fork_doh && i == L - 1));
            if (smooth_first) launch_smooth_flow(P, lsm[i], tmpF, e.w, e.h, g10, k_oct + e.octave, s, bt);   // Lsmooth and the conductivity in one pass
            if (smooth_first) fork_here();
            if (fused_level) {
                float* out = ((launches - 1) % 2 == 0) ? e.Lt : tmpP;
                for (int j = 0; j < head; j++) st[j] = e.tau[j] * 0.5f;
                const bool last_here = want_half && head == e.nsteps;
                launch_level_fused(P, lsm[i], !smooth_first && head < e.nsteps ? tmpF : nullptr, smooth_first ? tmpF : nullptr, out, e.w, e.h, g10,
                                   k_oct + e.octave, st, head, s, bt, last_here ? tmpH : nullptr);
                did_half = did_half || last_here;
                k = head;
                in = out;
                pass = 1;
                if (!smooth_first) fork_here();
            }
            for (; k < e.nsteps; pass++) {
                float* out = ((launches - 1 - pass) % 2 == 0) ? e.Lt : tmpP;
                // spread the steps evenly over the launches (e.g. 11 steps, fuse 8 -> 6 + 5)
                const int g = (e.nsteps - k + (launches - pass) - 1) / (launches - pass);
                for (int j = 0; j < g; j++) st[j] = e.tau[k + j] * 0.5f;
                const bool last_here = want_half && k + g == e.nsteps;
                if (launch_nld_multi(in, tmpF, out, e.w, e.h, st, g, s, bt, last_here ? tmpH : nullptr)) did_half = true;
                k += g;
                in = out;
            }
            if (did_half) fused_start = tmpH;
            if (e.nsteps == 0 && P != e.Lt)   // (never with AKAZE's parameters: every level but the first has FED steps)
                for (int bi = 0; bi < B; bi++)
                    HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(e.Lt) + (size_t)bi * slab, reinterpret_cast<const char*>(P) + (size_t)bi * slab,
                                             (size_t)e.w * e.h * 4, hipMemcpyDeviceToDevice, s));
        }
        float kside, kmid;
        deriv_weights(e.sigma_size, kside, kmid);
        // a1.5 + a1.6: first / second derivatives, determinant, and the level's 3x3 extrema (mask + candidate list)
        if (fork_doh && i == 0) {   // level 0: Lsmooth is Lt[0], ready after the base stage
            c.fork_open = true;
            if (flag_fork) {
                fork_value = c.arm_fork_signal().value;
            } else {
                if (!early_forked) HIP_CHECK(hipEventRecord(c.fork_event(0), s));
                HIP_CHECK(hipStreamWaitEvent(s_doh, c.fork_event(0), 0));
            }
        }
        if (flag_fork) {
            const DeferredHessian mine{i, smooth, kside, kmid, fork_value};
            if (!c.fork_pending.flag) {   // a later kernel of this level has taken the signal: the Hessian kernel can be queued now
                HIP_CHECK(hipStreamWaitValue32(s_doh, c.fork_flag, mine.value, hipStreamWaitValueGte, 0xFFFFFFFFu));
                launch_hessian(mine, s_doh);
            } else {
                deferred = mine;          // the next level's first kernel will carry it
            }
        } else {
            launch_hessian(DeferredHessian{i, smooth, kside, kmid, 0}, s_doh);
        }
    }
    if (flag_fork) flush_deferred();
    if (fork_doh) {   // join: everything after this point reads what the Hessian kernels wrote
        if (!c.join_event) HIP_CHECK(hipEventCreateWithFlags(&c.join_event, stream_event_flags()));
        HIP_CHECK(hipEventRecord(c.join_event, s_doh));
        HIP_CHECK(hipStreamWaitEvent(s, c.join_event, 0));
        c.fork_open = false;
    }
    HIP_CHECK(hipGetLastError());
    // ---- the keypoint stage (all levels), after the join
    run_stage(-1, L - 1);
    HIP_CHECK(hipGetLastError());

    AkazeDebugRequest& dbg = akaze_debug_request();
    if (dbg.armed && dbg.level >= 0 && dbg.level < L) {   // image 0 of the batch
        const LevelDesc& e = ev[dbg.level];
        const size_t n = (size_t)e.w * e.h;
        const void* src = nullptr;
        size_t bytes = n * 4;
        if (dbg.which == 2 || dbg.which == 3)   // de-interleave one component of (Lx, Ly)
            HIP_CHECK(hipMemcpy2DAsync(dbg.host_out, 4, reinterpret_cast<const char*>(e.Lxy) + (dbg.which == 3 ? 4 : 0), 8, 4, n, hipMemcpyDeviceToHost, s));
        switch (dbg.which) {
            case 0: src = e.Lt; break;
            case 4: src = e.Ldet; break;
            case 7: src = T.mask[dbg.level]; bytes = n; break;   // NB: after the sub-pixel filter as well (stages run it with the suppression)
            case 8: src = k_oct; bytes = 4; break;
        }
        if (src) HIP_CHECK(hipMemcpyAsync(dbg.host_out, src, bytes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        dbg.armed = false;
    }

    if (host_time_env) {   // diagnostic: the host's share of a call (everything up to here is enqueue work; the GPU may still be running)
        static thread_local double acc = 0;
        static thread_local int calls = 0;
        acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - host_t0).count();
        if (++calls % host_time_env == 0) {
            fprintf(stderr, "[apds] akaze_extract %dx%d x%d: %.1f us of host enqueue time per call\n", W, H, B, acc / host_time_env * 1e6);
            acc = 0;
        }
    }
    // ---- the only read-back of the call: every image's keypoint count
    int* K = K_host;
    if (K) {
        // the count went out in front of the orientation / descriptor kernels: wait for IT, not for the stream. What is still running reads
        // this thread's workspace: the thread's next call waits for it unless it goes to the same stream (ThreadCtx::mark_tail / ws_reset).
        c.mark_tail(s);
        HIP_CHECK(hipEventSynchronize(c.count_event));
    } else {
        K = c.pinned_ints(B);
        if (B == 1) {
            HIP_CHECK(hipMemcpyAsync(K, kp_base + n_stage, sizeof(int), hipMemcpyDeviceToHost, s));
        } else {
            hipLaunchKernelGGL(gather_slab_ints_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, (const int*)(kp_base + n_stage), 1, slab, B, counts_dev);
            HIP_CHECK(hipMemcpyAsync(K, counts_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
    }
    int kmax = 0;
    bool over = false;
    for (int bi = 0; bi < B; bi++) {
        counts[bi] = std::min(K[bi], max_points);
        kmax = std::max(kmax, counts[bi]);
        over |= K[bi] > max_points;
        APDS_REQUIRE(counts[bi] <= capacity, APDS_ERR_ASSERT, "output capacity smaller than the keypoint count");
    }
    c.akaze_kp_estimate = kmax;
    if (over) {
        // some image has more keypoints than max_points (rare): those images again, through the rank selection (response descending,
        // ties by detection order) over ALL their keypoints; the tables of image bi are image 0's shifted by bi slabs
        const int nb_all = ceil_div(total_pix, KP_BYTES_PER_BLOCK);
        for (int bi = 0; bi < B; bi++) {
            if (K[bi] <= max_points) continue;
            LevelTable Tb = T;
            Tb.bstride = 0;
            for (int i = 0; i < L; i++) {
                Tb.Lt[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(T.Lt[i]) + (size_t)bi * slab);
                Tb.Lxy[i] = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(T.Lxy[i]) + (size_t)bi * slab);
                Tb.Ldet[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(T.Ldet[i]) + (size_t)bi * slab);
                Tb.mask[i] = T.mask[i] + (size_t)bi * slab;
            }
            const int keep = counts[bi];
            apds_keypoint* kp_b = kps_out + (size_t)bi * capacity;
            apds_keypoint* kps_all = c.alloc_n<apds_keypoint>(K[bi]);
            int* bc = reinterpret_cast<int*>(reinterpret_cast<char*>(block_counts) + (size_t)bi * slab);
            int* base2 = bc + nb_all;   // two ints behind the block counts: {0, total}
            HIP_CHECK(hipMemsetAsync(base2, 0, 2 * sizeof(int), s));
            hipLaunchKernelGGL(kp_block_counts_kernel, dim3(nb_all), dim3(SCAN_BLOCK), 0, s, (const uint8_t*)(mask_all + (size_t)bi * slab), 0ll, total_pix, bc,
                               (size_t)0);
            hipLaunchKernelGGL(kp_scan_offsets_kernel, dim3(1), dim3(1024), 0, s, bc, nb_all, base2, (size_t)0);
            hipLaunchKernelGGL(emit_keypoints_kernel, dim3(nb_all), dim3(SCAN_BLOCK), 0, s, Tb, (const uint8_t*)(mask_all + (size_t)bi * slab), 0ll, total_pix,
                               (const int*)bc, (const int*)nullptr, kps_all, K[bi], (size_t)0);
            hipLaunchKernelGGL(rank_select_kernel, dim3(ceil_div(K[bi], 256)), dim3(256), 0, s, (const apds_keypoint*)kps_all, K[bi], keep, kp_b);
            hipLaunchKernelGGL(orientation_kernel, dim3(ceil_div(keep, 4)), dim3(256), 0, s, Tb, kp_b, (const int*)nullptr, keep, (size_t)0, ang_step, nkeys, 0);
            hipLaunchKernelGGL(mldb_kernel, dim3(ceil_div(keep, 4)), dim3(256), 0, s, Tb, (const apds_keypoint*)kp_b, (const int*)nullptr, keep, (size_t)0,
                               reinterpret_cast<uint32_t*>(desc64_out + (size_t)bi * desc_bstride), (size_t)0, 0);
        }
        HIP_CHECK(hipGetLastError());
        // these kernels read the calling thread's workspace (kps_all, masks, planes): the same thread's NEXT call, possibly on another
        // stream (apds_dev_* take one), starts by re-using that memory, so the rare over-capacity path ends synchronously like the main one
        HIP_CHECK(hipStreamSynchronize(s));
    }
    return kmax;
}

int akaze_extract_device(const void* img, int rows, int cols, int channels, size_t stride, int max_points, apds_keypoint* kps_out,
                         uint8_t* desc64_out, int capacity, hipStream_t s) {
    int count = 0;
    akaze_extract_batch_device(img, 1, 0, rows, cols, channels, stride, max_points, kps_out, desc64_out, capacity, &count, s);
    return count;
}

}  // namespace apds
