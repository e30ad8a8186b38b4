// csrc/keypoint_table.hip — GPU-resident keypoint table: what fills the descriptor database the matcher scans
// (SURVEY §8f-1). Mirrors the reference's `keypoint` table and its access paths, without Postgres:
//   insert  : preprocessor/src/main.rs:296-324 (to_db_type + level-of-detail rescale of x,y, one INSERT per tile)
//   read    : feature_database/src/keypointdb.rs:38-90 (by image id / by level of detail / by bounding box at a level of
//             detail), each `ORDER BY response DESC LIMIT 262143` (OPENCV_KEYPOINT_LIMIT, keypointdb.rs:12)
// Columns are stored SoA (coalesced predicate scans); descriptors as 64-byte rows, i.e. already in the layout the
// Hamming kernel streams, so a selection is directly usable as a train set and `train_idx` = position in the selection
// (the index the reference would get from the returned Vec).
// Selection = predicate flags -> ordered compaction of u64 keys (~response_bits << 32 | row) -> radix select of the
// LIMIT-th key when more rows qualify -> bitonic sort of the survivors -> row gather. Ties in response are ordered by
// insertion order (row id), which SQL leaves unspecified.
#include <cmath>
#include <cstdlib>

#include "kernels.h"

namespace apds {

struct KeypointTable {
    int device = 0;
    int64_t capacity = 0, n = 0;
    float *x = nullptr, *y = nullptr, *size = nullptr, *angle = nullptr, *response = nullptr;
    int *octave = nullptr, *class_id = nullptr, *image_id = nullptr, *lod = nullptr;
    uint8_t* desc64 = nullptr;
    // last selection (device)
    int64_t view_n = 0, view_cap = 0;
    int* view_rows = nullptr;
    apds_keypoint* view_kps = nullptr;
    int* view_image_id = nullptr;
    uint8_t* view_desc64 = nullptr;
};

__global__ void table_insert_kernel(const apds_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc61, int n, int image_id, int lod,
                                    float scale, float xoff, float yoff, long long base, float* x, float* y, float* size, float* angle,
                                    float* response, int* octave, int* class_id, int* img, int* lodcol, uint32_t* desc64) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * 16) return;
    const int i = (int)(t >> 4), w = (int)(t & 15);
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++)
        if (w * 4 + b < 61) v |= (uint32_t)desc61[(size_t)i * 61 + w * 4 + b] << (8 * b);
    desc64[(base + i) * 16 + w] = v;
    if (w == 0) {
        const apds_keypoint k = kps[i];
        // main.rs:299-300: x * 2^lod + (column * tile_w * 2^lod) as f32
        x[base + i] = k.x * scale + xoff;
        y[base + i] = k.y * scale + yoff;
        size[base + i] = k.size;
        angle[base + i] = k.angle;
        response[base + i] = k.response;
        octave[base + i] = k.octave;
        class_id[base + i] = k.class_id;
        img[base + i] = image_id;
        lodcol[base + i] = lod;
    }
}

struct SelectArgs {
    int mode;   // 0 image id, 1 level of detail, 2 level of detail + bounding box
    int value;
    float x0, y0, x1, y1;   // already floor()/ceil()ed
};

__global__ void table_flags_kernel(const int* __restrict__ img, const int* __restrict__ lod, const float* __restrict__ x, const float* __restrict__ y,
                                   long long n, SelectArgs a, uint8_t* __restrict__ flags) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool f;
    if (a.mode == 0) f = img[i] == a.value;
    else if (a.mode == 1) f = lod[i] == a.value;
    else f = lod[i] == a.value && x[i] >= a.x0 && x[i] <= a.x1 && y[i] >= a.y0 && y[i] <= a.y1;
    flags[i] = f;
}

// ascending u64 key order == response descending, then row ascending (responses are positive floats)
__device__ __forceinline__ uint64_t order_key(float response, uint32_t row) {
    return ((uint64_t)(~__float_as_uint(response)) << 32) | row;
}

static constexpr int TB = 1024;

__global__ __launch_bounds__(TB) void table_emit_keys_kernel(const uint8_t* __restrict__ flags, const float* __restrict__ response, int n,
                                                             const int* __restrict__ block_offsets, uint64_t* __restrict__ keys) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ int wsum[TB / 64];
    const int i = blockIdx.x * TB + threadIdx.x;
    const int f = i < n ? (flags[i] != 0) : 0;
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = __popcll(b);
    __syncthreads();
    if (!f) return;
    int before = 0;
    for (int k = 0; k < wv; k++) before += wsum[k];
    keys[block_offsets[blockIdx.x] + before + __popcll(b & ((1ull << lane) - 1ull))] = order_key(response[i], (uint32_t)i);
}

// one radix-select pass: histogram of byte `shift/8` over keys that match `prefix` on the bits above it
__global__ void key_hist_kernel(const uint64_t* __restrict__ keys, int m, uint64_t prefix, uint64_t mask_hi, int shift, unsigned int* __restrict__ hist) {
    APDS_RAISE_WAVE_PRIORITY();
    __shared__ unsigned int s[256];
    if (threadIdx.x < 256) s[threadIdx.x] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        if ((k & mask_hi) == prefix) atomicAdd(&s[(k >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256 && s[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s[threadIdx.x]);
}

__global__ void key_leq_flags_kernel(const uint64_t* __restrict__ keys, int m, uint64_t kth, uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flags[i] = keys[i] <= kth;
}

__global__ __launch_bounds__(TB) void key_compact_kernel(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ flags, int m,
                                                         const int* __restrict__ block_offsets, uint64_t* __restrict__ out) {
    __shared__ int wsum[TB / 64];
    const int i = blockIdx.x * TB + threadIdx.x;
    const int f = i < m ? (flags[i] != 0) : 0;
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = __popcll(b);
    __syncthreads();
    if (!f) return;
    int before = 0;
    for (int k = 0; k < wv; k++) before += wsum[k];
    out[block_offsets[blockIdx.x] + before + __popcll(b & ((1ull << lane) - 1ull))] = keys[i];
}

__global__ void bitonic_step_kernel(uint64_t* __restrict__ keys, int n_pow2, int k, int j) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    const int l = i ^ j;
    if (l > i) {
        const uint64_t a = keys[i], b = keys[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
            keys[i] = b;
            keys[l] = a;
        }
    }
}

__global__ void table_gather_kernel(const uint64_t* __restrict__ keys, int m, const float* x, const float* y, const float* size, const float* angle,
                                    const float* response, const int* octave, const int* class_id, const int* img, const uint32_t* desc64,
                                    int* __restrict__ rows, apds_keypoint* __restrict__ kps, int* __restrict__ out_img, uint32_t* __restrict__ out_desc) {
    APDS_RAISE_WAVE_PRIORITY();
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)m * 16) return;
    const int i = (int)(t >> 4), w = (int)(t & 15);
    const uint32_t r = (uint32_t)keys[i];
    out_desc[(size_t)i * 16 + w] = desc64[(size_t)r * 16 + w];
    if (w == 0) {
        rows[i] = (int)r;
        apds_keypoint k;
        k.x = x[r]; k.y = y[r]; k.size = size[r]; k.angle = angle[r]; k.response = response[r];
        k.octave = octave[r]; k.class_id = class_id[r];
        kps[i] = k;
        out_img[i] = img[r];
    }
}

}  // namespace apds

using namespace apds;

namespace {
template <class T>
void dev_alloc(T*& p, size_t n) {
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)));
}
void table_free(KeypointTable* t) {
    for (void* p : {(void*)t->x, (void*)t->y, (void*)t->size, (void*)t->angle, (void*)t->response, (void*)t->octave, (void*)t->class_id,
                    (void*)t->image_id, (void*)t->lod, (void*)t->desc64, (void*)t->view_rows, (void*)t->view_kps, (void*)t->view_image_id,
                    (void*)t->view_desc64})
        if (p) (void)hipFree(p);
    delete t;
}
}  // namespace

extern "C" {

int apds_db_create(void** db, int64_t capacity) {
    return guarded([&] {
        APDS_REQUIRE(db && capacity > 0 && capacity < (1ll << 31), APDS_ERR_BAD_ARG, "bad capacity");
        ThreadCtx& c = ctx();
        KeypointTable* t = new KeypointTable();
        t->device = c.device;
        t->capacity = capacity;
        try {
            dev_alloc(t->x, capacity); dev_alloc(t->y, capacity); dev_alloc(t->size, capacity); dev_alloc(t->angle, capacity);
            dev_alloc(t->response, capacity); dev_alloc(t->octave, capacity); dev_alloc(t->class_id, capacity);
            dev_alloc(t->image_id, capacity); dev_alloc(t->lod, capacity); dev_alloc(t->desc64, (size_t)capacity * 64);
            t->view_cap = std::min<int64_t>(capacity, APDS_MAX_POINTS);
            dev_alloc(t->view_rows, t->view_cap); dev_alloc(t->view_kps, t->view_cap); dev_alloc(t->view_image_id, t->view_cap);
            dev_alloc(t->view_desc64, (size_t)t->view_cap * 64);
        } catch (...) {
            table_free(t);
            throw;
        }
        *db = t;
    });
}

int apds_db_destroy(void* db) {
    return guarded([&] {
        if (db) table_free(static_cast<KeypointTable*>(db));
    });
}

int64_t apds_db_rows(const void* db) { return db ? static_cast<const KeypointTable*>(db)->n : -1; }

// preprocessor/src/main.rs:296-324: all keypoints of one tile image, coordinates lifted to level-of-detail-0 mosaic pixels
int apds_db_insert_image(void* db, const apds_keypoint* kps, const uint8_t* desc61, int n, int image_id, int lod, uint64_t column, uint64_t row,
                         uint64_t tile_w, uint64_t tile_h) {
    return guarded([&] {
        KeypointTable* t = static_cast<KeypointTable*>(db);
        APDS_REQUIRE(t && n >= 0 && lod >= 0 && lod < 31, APDS_ERR_BAD_ARG, "bad argument");
        APDS_REQUIRE(t->n + n <= t->capacity, APDS_ERR_NOMEM, "keypoint table is full");
        if (!n) return;
        APDS_REQUIRE(kps && desc61, APDS_ERR_BAD_ARG, "null argument");
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        apds_keypoint* dk = c.alloc_n<apds_keypoint>(n);
        uint8_t* dd = c.alloc_n<uint8_t>((size_t)n * 61);
        HIP_CHECK(hipMemcpyAsync(dk, kps, (size_t)n * sizeof(apds_keypoint), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(dd, desc61, (size_t)n * 61, hipMemcpyHostToDevice, s));
        const float scale = ldexpf(1.0f, lod);                                   // 2_f32.powi(lod)
        const float xoff = (float)(column * tile_w * (1ull << lod));               // (u64 product) as f32
        const float yoff = (float)(row * tile_h * (1ull << lod));
        hipLaunchKernelGGL(table_insert_kernel, dim3(ceil_div((long long)n * 16, 256)), dim3(256), 0, s, (const apds_keypoint*)dk, (const uint8_t*)dd, n,
                           image_id, lod, scale, xoff, yoff, (long long)t->n, t->x, t->y, t->size, t->angle, t->response, t->octave, t->class_id, t->image_id,
                           t->lod, reinterpret_cast<uint32_t*>(t->desc64));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s));
        t->n += n;
    });
}

// keypointdb.rs:38-90. mode 0: image_id == value; 1: level_of_detail == value; 2: level_of_detail == value and
// floor(x_start) <= x <= ceil(x_end), floor(y_start) <= y <= ceil(y_end). Result (ordered by response desc, at most 262143
// rows) stays on the device as the table's current view; *n_out is its row count.
int apds_db_select(void* db, int mode, int value, float x_start, float y_start, float x_end, float y_end, int* n_out) {
    return guarded([&] {
        KeypointTable* t = static_cast<KeypointTable*>(db);
        APDS_REQUIRE(t && n_out && mode >= 0 && mode <= 2, APDS_ERR_BAD_ARG, "bad argument");
        *n_out = 0;
        t->view_n = 0;
        const int n = (int)t->n;
        if (!n) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        SelectArgs a{mode, value, floorf(x_start), floorf(y_start), ceilf(x_end), ceilf(y_end)};
        uint8_t* flags = c.alloc_n<uint8_t>(n);
        hipLaunchKernelGGL(table_flags_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, (const int*)t->image_id, (const int*)t->lod, (const float*)t->x,
                           (const float*)t->y, (long long)n, a, flags);
        int* total_dev = nullptr;
        int* offs = scan_flags_device(flags, n, &total_dev, s);
        int m = 0;
        HIP_CHECK(hipMemcpyAsync(&m, total_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (!m) return;
        uint64_t* keys = c.alloc_n<uint64_t>(m);
        hipLaunchKernelGGL(table_emit_keys_kernel, dim3(ceil_div(n, TB)), dim3(TB), 0, s, (const uint8_t*)flags, (const float*)t->response, n, (const int*)offs,
                           keys);
        const int limit = (int)std::min<int64_t>(APDS_MAX_POINTS, t->view_cap);
        if (m > limit) {
            // radix select of the limit-th smallest key, most significant byte first
            unsigned int* hist = c.alloc_n<unsigned int>(256);
            uint64_t prefix = 0, mask_hi = 0;
            unsigned int k = (unsigned int)(limit - 1), h[256];
            for (int shift = 56; shift >= 0; shift -= 8) {
                HIP_CHECK(hipMemsetAsync(hist, 0, 256 * sizeof(unsigned int), s));
                hipLaunchKernelGGL(key_hist_kernel, dim3(256), dim3(256), 0, s, (const uint64_t*)keys, m, prefix, mask_hi, shift, hist);
                HIP_CHECK(hipMemcpyAsync(h, hist, sizeof(h), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                unsigned int b = 0;
                for (; b < 256; b++) {
                    if (k < h[b]) break;
                    k -= h[b];
                }
                prefix |= (uint64_t)b << shift;
                mask_hi |= 0xFFull << shift;
            }
            uint8_t* f2 = c.alloc_n<uint8_t>(m);
            hipLaunchKernelGGL(key_leq_flags_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, s, (const uint64_t*)keys, m, prefix, f2);
            int* tot2 = nullptr;
            int* offs2 = scan_flags_device(f2, m, &tot2, s);
            uint64_t* kept = c.alloc_n<uint64_t>(limit);
            hipLaunchKernelGGL(key_compact_kernel, dim3(ceil_div(m, TB)), dim3(TB), 0, s, (const uint64_t*)keys, (const uint8_t*)f2, m, (const int*)offs2, kept);
            keys = kept;
            m = limit;
        }
        // bitonic sort of the (<= 2^18) surviving keys, padded with +inf keys
        int p2 = 1;
        while (p2 < m) p2 <<= 1;
        uint64_t* sorted = c.alloc_n<uint64_t>(p2);
        HIP_CHECK(hipMemsetAsync(sorted, 0xFF, (size_t)p2 * 8, s));
        HIP_CHECK(hipMemcpyAsync(sorted, keys, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
        for (int k = 2; k <= p2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) hipLaunchKernelGGL(bitonic_step_kernel, dim3(ceil_div(p2, 256)), dim3(256), 0, s, sorted, p2, k, j);
        hipLaunchKernelGGL(table_gather_kernel, dim3(ceil_div((long long)m * 16, 256)), dim3(256), 0, s, (const uint64_t*)sorted, m, (const float*)t->x,
                           (const float*)t->y, (const float*)t->size, (const float*)t->angle, (const float*)t->response, (const int*)t->octave,
                           (const int*)t->class_id, (const int*)t->image_id, reinterpret_cast<const uint32_t*>(t->desc64), t->view_rows, t->view_kps,
                           t->view_image_id, reinterpret_cast<uint32_t*>(t->view_desc64));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s));
        t->view_n = m;
        *n_out = m;
    });
}

// device pointers of the current view: rows64 (n x 64 B, a ready train set), keypoints (n x 28 B), table row ids, image ids
int apds_db_view(void* db, void** rows64, void** kps, void** row_ids, void** image_ids, int* n) {
    return guarded([&] {
        KeypointTable* t = static_cast<KeypointTable*>(db);
        APDS_REQUIRE(t, APDS_ERR_BAD_ARG, "null table");
        if (rows64) *rows64 = t->view_desc64;
        if (kps) *kps = t->view_kps;
        if (row_ids) *row_ids = t->view_rows;
        if (image_ids) *image_ids = t->view_image_id;
        if (n) *n = (int)t->view_n;
    });
}

// BFMatcher.knnMatch of host query descriptors against the CURRENT VIEW (resident on the device): idx = position in the view
int apds_db_knn_match(void* db, const uint8_t* query_desc, int n_query, int desc_bytes, int k, int32_t* idx, int32_t* dist) {
    return guarded([&] {
        KeypointTable* t = static_cast<KeypointTable*>(db);
        APDS_REQUIRE(t && idx && dist && n_query >= 0, APDS_ERR_BAD_ARG, "bad argument");
        APDS_REQUIRE(desc_bytes >= 1 && desc_bytes <= 64, APDS_ERR_ASSERT, "descriptor length must be 1..64 bytes");
        APDS_REQUIRE(k == 1 || k == 2, APDS_ERR_ASSERT, "k in {1,2}");
        if (!n_query) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        uint8_t* raw = c.alloc_n<uint8_t>((size_t)n_query * desc_bytes);
        uint8_t* q64 = c.alloc_n<uint8_t>((size_t)n_query * 64);
        uint64_t* keys = c.alloc_n<uint64_t>((size_t)n_query * k);
        HIP_CHECK(hipMemcpyAsync(raw, query_desc, (size_t)n_query * desc_bytes, hipMemcpyHostToDevice, s));
        pack_rows_device(raw, n_query, desc_bytes, desc_bytes, q64, s);
        hamming_topk_device(q64, n_query, t->view_desc64, t->view_n, 0, k, keys, s);
        std::vector<uint64_t> h((size_t)n_query * k);
        HIP_CHECK(hipMemcpyAsync(h.data(), keys, h.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < h.size(); i++) {
            if (h[i] == ~0ull) idx[i] = -1, dist[i] = 0x7fffffff;
            else idx[i] = (int32_t)(uint32_t)h[i], dist[i] = (int32_t)(h[i] >> 32);
        }
    });
}

// host copy of the current view: what the reference's Vec<models::Keypoint> holds (id = row + 1 like a SERIAL column)
int apds_db_view_download(void* db, apds_keypoint* kps, uint8_t* desc61, int32_t* ids, int32_t* image_ids) {
    return guarded([&] {
        KeypointTable* t = static_cast<KeypointTable*>(db);
        APDS_REQUIRE(t, APDS_ERR_BAD_ARG, "null table");
        const int m = (int)t->view_n;
        if (!m) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        if (kps) HIP_CHECK(hipMemcpyAsync(kps, t->view_kps, (size_t)m * sizeof(apds_keypoint), hipMemcpyDeviceToHost, s));
        if (image_ids) HIP_CHECK(hipMemcpyAsync(image_ids, t->view_image_id, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        std::vector<int> rows;
        if (ids) {
            rows.resize(m);
            HIP_CHECK(hipMemcpyAsync(rows.data(), t->view_rows, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        }
        if (desc61) {
            uint8_t* d61 = c.alloc_n<uint8_t>((size_t)m * 61);
            HIP_CHECK(hipMemcpy2DAsync(d61, 61, t->view_desc64, 64, 61, m, hipMemcpyDeviceToDevice, s));
            HIP_CHECK(hipMemcpyAsync(desc61, d61, (size_t)m * 61, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        if (ids)
            for (int i = 0; i < m; i++) ids[i] = rows[i] + 1;
    });
}

}  // extern "C"
