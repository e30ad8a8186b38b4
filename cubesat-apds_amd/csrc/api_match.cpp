// csrc/api_match.cpp — C-ABI entry points of the descriptor-match step (feature_extraction/src/lib.rs:94-180).
#include <climits>
#include <cstdlib>
#include <cstring>

#include "config.h"
#include "kernels.h"

using namespace apds;

namespace {

// host rows (packed, desc_bytes each) -> device 64-byte rows in the thread workspace
void* upload_rows64(const uint8_t* host, long long n, int desc_bytes, hipStream_t s) {
    ThreadCtx& c = ctx();
    void* raw = c.alloc((size_t)n * desc_bytes);
    void* rows = c.alloc((size_t)n * 64);
    HIP_CHECK(hipMemcpyAsync(raw, host, (size_t)n * desc_bytes, hipMemcpyHostToDevice, s));
    pack_rows_device(raw, n, desc_bytes, desc_bytes, rows, s);
    return rows;
}

void check_desc_args(const void* a, int na, const void* b, int nb, int desc_bytes) {
    APDS_REQUIRE(na >= 0 && nb >= 0, APDS_ERR_ASSERT, "negative row count");
    APDS_REQUIRE(desc_bytes >= 1 && desc_bytes <= 64, APDS_ERR_ASSERT, "descriptor length must be 1..64 bytes (M-LDB is 61)");
    APDS_REQUIRE((a || na == 0) && (b || nb == 0), APDS_ERR_BAD_ARG, "null descriptor pointer");
}

template <class T>
T* host_alloc(size_t n) {
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(n, 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    return p;
}

}  // namespace

extern "C" {

int apds_knn_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes, int k, int32_t* idx, int32_t* dist) {
    APDS_RANGE("apds_knn_match");
    return guarded([&] {
        check_desc_args(q, nq, t, nt, desc_bytes);
        APDS_REQUIRE(k >= 1, APDS_ERR_ASSERT, "k must be >= 1");
        APDS_REQUIRE(idx && dist, APDS_ERR_BAD_ARG, "null output");
        if (nq == 0) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        if (nt == 0) {
            for (long long i = 0; i < (long long)nq * k; i++) idx[i] = -1, dist[i] = INT_MAX;
            return;
        }
        void* dq = upload_rows64(q, nq, desc_bytes, s);
        void* dt = upload_rows64(t, nt, desc_bytes, s);
        uint64_t* keys = c.alloc_n<uint64_t>((size_t)nq * k);
        hamming_topk_device(dq, nq, dt, nt, 0, k, keys, s);
        std::vector<uint64_t> h((size_t)nq * k);
        HIP_CHECK(hipMemcpyAsync(h.data(), keys, h.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < h.size(); i++) {
            if (h[i] == ~0ull) idx[i] = -1, dist[i] = INT_MAX;
            else idx[i] = (int32_t)(uint32_t)h[i], dist[i] = (int32_t)(h[i] >> 32);
        }
    });
}

int apds_get_knn_matches(const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes, int k, float filter_strength,
                         apds_dmatch** matches, int* n_matches) {
    APDS_RANGE("apds_get_knn_matches");
    return guarded([&] {
        APDS_REQUIRE(matches && n_matches, APDS_ERR_BAD_ARG, "null output");
        *matches = nullptr;
        *n_matches = 0;
        check_desc_args(q, nq, t, nt, desc_bytes);
        APDS_REQUIRE(k >= 1, APDS_ERR_ASSERT, "k must be >= 1");
        if (nq == 0 || nt == 0) {   // knnMatch yields no rows, the filter loop never runs
            *matches = host_alloc<apds_dmatch>(0);
            return;
        }
        // lib.rs:108 `i.get(1)?` fails on the first query when fewer than two neighbours exist
        APDS_REQUIRE(k >= 2 && nt >= 2, APDS_ERR_OUT_OF_RANGE, "fewer than 2 neighbours per query (k < 2 or target rows < 2)");
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        void* dq = upload_rows64(q, nq, desc_bytes, s);
        void* dt = upload_rows64(t, nt, desc_bytes, s);
        uint64_t* keys = c.alloc_n<uint64_t>((size_t)nq * 2);
        hamming_topk_device(dq, nq, dt, nt, 0, 2, keys, s);   // only the two nearest are ever read (lib.rs:107-111)
        apds_dmatch* dm = c.alloc_n<apds_dmatch>(nq);
        const int n = ratio_filter_device(keys, nq, 2, filter_strength, dm, s);
        apds_dmatch* out = host_alloc<apds_dmatch>(n);
        if (n) HIP_CHECK(hipMemcpyAsync(out, dm, (size_t)n * sizeof(apds_dmatch), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        *matches = out;
        *n_matches = n;
    });
}

int apds_get_bruteforce_matches(const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes, apds_dmatch** matches, int* n_matches) {
    APDS_RANGE("apds_get_bruteforce_matches");
    return guarded([&] {
        APDS_REQUIRE(matches && n_matches, APDS_ERR_BAD_ARG, "null output");
        *matches = nullptr;
        *n_matches = 0;
        check_desc_args(q, nq, t, nt, desc_bytes);
        if (nq == 0 || nt == 0) {
            *matches = host_alloc<apds_dmatch>(0);
            return;
        }
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        void* dq = upload_rows64(q, nq, desc_bytes, s);
        void* dt = upload_rows64(t, nt, desc_bytes, s);
        // batchDistance(crosscheck): nearest QUERY of every train row, then per query the closest such train row
        uint64_t* tbest = c.alloc_n<uint64_t>((size_t)nt);
        hamming_topk_device(dt, nt, dq, nq, 0, 1, tbest, s);
        apds_dmatch* dm = c.alloc_n<apds_dmatch>(nq);
        const int n = cross_check_device(tbest, nt, nq, dm, s);
        apds_dmatch* out = host_alloc<apds_dmatch>(n);
        if (n) HIP_CHECK(hipMemcpyAsync(out, dm, (size_t)n * sizeof(apds_dmatch), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        *matches = out;
        *n_matches = n;
    });
}

int apds_get_points_from_matches(const apds_keypoint* kp1, int n1, const apds_keypoint* kp2, int n2, const apds_dmatch* m, int nm,
                                 int bug_compatible, float* pts1, float* pts2) {
    return guarded([&] {
        APDS_REQUIRE(nm >= 0 && n1 >= 0 && n2 >= 0, APDS_ERR_ASSERT, "negative count");
        if (nm == 0) return;
        APDS_REQUIRE(kp1 && kp2 && m && pts1 && pts2, APDS_ERR_BAD_ARG, "null argument");
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        apds_keypoint* d1 = c.alloc_n<apds_keypoint>(std::max(n1, 1));
        apds_keypoint* d2 = c.alloc_n<apds_keypoint>(std::max(n2, 1));
        apds_dmatch* dm = c.alloc_n<apds_dmatch>(nm);
        float* p1 = c.alloc_n<float>((size_t)nm * 2);
        float* p2 = c.alloc_n<float>((size_t)nm * 2);
        int* err = c.alloc_n<int>(1);
        if (n1) HIP_CHECK(hipMemcpyAsync(d1, kp1, (size_t)n1 * sizeof(apds_keypoint), hipMemcpyHostToDevice, s));
        if (n2) HIP_CHECK(hipMemcpyAsync(d2, kp2, (size_t)n2 * sizeof(apds_keypoint), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(dm, m, (size_t)nm * sizeof(apds_dmatch), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
        points_from_matches_device(d1, n1, d2, n2, dm, nm, bug_compatible, p1, p2, err, s);
        int herr = 0;
        HIP_CHECK(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(pts1, p1, (size_t)nm * 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(pts2, p2, (size_t)nm * 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        APDS_REQUIRE(herr == 0, APDS_ERR_OUT_OF_RANGE, "match index outside the keypoint vector");
    });
}

int apds_raster_to_mat(const uint8_t* rgba, size_t n_pixels, int w, int h, uint8_t* bgra) {
    return guarded([&] {
        // mod.rs:185-187: Err(MatError::Unknown) if pixels.len() != w*h
        APDS_REQUIRE(w > 0 && h > 0 && n_pixels == (size_t)w * (size_t)h, APDS_ERR_BAD_ARG, "pixel count != w*h");
        APDS_REQUIRE(rgba && bgra, APDS_ERR_BAD_ARG, "null argument");
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        uint8_t* din = c.alloc_n<uint8_t>(n_pixels * 4);
        uint8_t* dout = c.alloc_n<uint8_t>(n_pixels * 4);
        HIP_CHECK(hipMemcpyAsync(din, rgba, n_pixels * 4, hipMemcpyHostToDevice, s));
        rgba_to_bgra_device(din, n_pixels, dout, s);
        HIP_CHECK(hipMemcpyAsync(bgra, dout, n_pixels * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

// ---- device-resident variants ------------------------------------------------------------------------
int apds_dev_pack_descriptors(const void* src, int64_t n, int desc_bytes, int64_t src_stride, void* dst, void* stream) {
    return guarded([&] {
        APDS_REQUIRE(desc_bytes >= 1 && desc_bytes <= 64 && src_stride >= desc_bytes && n >= 0, APDS_ERR_ASSERT, "bad descriptor geometry");
        pack_rows_device(src, n, desc_bytes, src_stride, dst, pick_stream(stream));
    });
}

int apds_dev_hamming_topk(const void* q, int nq, const void* t, int64_t nt, uint32_t index_base, int k, void* out_keys, void* stream) {
    APDS_RANGE("apds_dev_hamming_topk");
    return guarded([&] {
        APDS_REQUIRE(nq >= 0 && nt >= 0, APDS_ERR_ASSERT, "negative row count");
        ctx().ws_reset();
        hamming_topk_device(q, nq, t, nt, index_base, k, static_cast<uint64_t*>(out_keys), pick_stream(stream));
    });
}

int apds_dev_hamming_topk_backend(const void* q, int nq, const void* t, int64_t nt, uint32_t index_base, int k, void* out_keys, int backend, void* stream) {
    APDS_RANGE("apds_dev_hamming_topk_backend");
    return guarded([&] {
        APDS_REQUIRE(nq >= 0 && nt >= 0, APDS_ERR_ASSERT, "negative row count");
        ctx().ws_reset();
        hamming_topk_device(q, nq, t, nt, index_base, k, static_cast<uint64_t*>(out_keys), pick_stream(stream), backend);
    });
}

int apds_dev_topk_state_create(void** state) {
    return guarded([&] {
        APDS_REQUIRE(state, APDS_ERR_BAD_ARG, "null output");
        *state = topk_split_create();
    });
}

int apds_dev_topk_state_destroy(void* state) {
    return guarded([&] { topk_split_destroy(state); });
}

int apds_dev_topk_prepass(void* state, const void* q, int nq, const void* t, int64_t nt, uint32_t index_base, int k, void* stream) {
    APDS_RANGE("apds_dev_topk_prepass");
    return guarded([&] { topk_split_prepass(state, q, nq, t, nt, index_base, k, pick_stream(stream)); });
}

int apds_dev_topk_scan(void* state, const void* q, const void* t, void* stream) {
    APDS_RANGE("apds_dev_topk_scan");
    return guarded([&] { topk_split_scan(state, q, t, pick_stream(stream)); });
}

int apds_dev_topk_merge(void* state, uint32_t index_base, void* out_keys, void* stream) {
    APDS_RANGE("apds_dev_topk_merge");
    return guarded([&] { topk_split_merge(state, index_base, static_cast<uint64_t*>(out_keys), pick_stream(stream)); });
}

int apds_dev_match_lds_cap(int bytes, int* previous) {
    return guarded([&] {
        APDS_REQUIRE(bytes >= 0 && bytes <= 64 * 1024, APDS_ERR_BAD_ARG, "cap must be 0 .. 65536 bytes");
        const int old = match_lds_cap().exchange(bytes);
        if (previous) *previous = old;
    });
}

int apds_dev_match_last_launch_lds(int* bytes) {
    return guarded([&] {
        APDS_REQUIRE(bytes, APDS_ERR_BAD_ARG, "null output");
        *bytes = last_scan_launch_lds().load();
    });
}

int apds_dev_match_backend(int* matrix_cores) {
    return guarded([&] {
        APDS_REQUIRE(matrix_cores, APDS_ERR_BAD_ARG, "null output");
        *matrix_cores = config().match_mfma ? 1 : 0;
    });
}

int apds_dev_merge_topk(const void* parts, int nparts, int nq, int k, void* out_keys, void* stream) {
    return guarded([&] { merge_topk_device(static_cast<const uint64_t*>(parts), nparts, nq, k, static_cast<uint64_t*>(out_keys), pick_stream(stream)); });
}

int apds_dev_ratio_filter(const void* keys, int nq, int k, float fs, void* out_matches, int* n_matches, void* stream) {
    APDS_RANGE("apds_dev_ratio_filter");
    return guarded([&] {
        APDS_REQUIRE(k >= 2, APDS_ERR_OUT_OF_RANGE, "ratio test needs two neighbours");
        APDS_REQUIRE(n_matches, APDS_ERR_BAD_ARG, "null output");
        ctx().ws_reset();
        *n_matches = ratio_filter_device(static_cast<const uint64_t*>(keys), nq, k, fs, static_cast<apds_dmatch*>(out_matches), pick_stream(stream));
    });
}

int apds_dev_cross_check(const void* train_best, int64_t n_train, int nq, void* out_matches, int* n_matches, void* stream) {
    return guarded([&] {
        APDS_REQUIRE(n_matches, APDS_ERR_BAD_ARG, "null output");
        ctx().ws_reset();
        *n_matches = cross_check_device(static_cast<const uint64_t*>(train_best), n_train, nq, static_cast<apds_dmatch*>(out_matches), pick_stream(stream));
    });
}

int apds_dev_points_from_matches(const void* kp1, int n1, const void* kp2, int n2, const void* m, int nm, int bug_compatible, void* pts1,
                                 void* pts2, void* stream) {
    return guarded([&] {
        if (nm <= 0) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = pick_stream(stream);
        int* err = c.alloc_n<int>(1);
        HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
        points_from_matches_device(static_cast<const apds_keypoint*>(kp1), n1, static_cast<const apds_keypoint*>(kp2), n2,
                                   static_cast<const apds_dmatch*>(m), nm, bug_compatible, static_cast<float*>(pts1), static_cast<float*>(pts2), err, s);
        int herr = 0;
        HIP_CHECK(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        APDS_REQUIRE(herr == 0, APDS_ERR_OUT_OF_RANGE, "match index outside the keypoint vector");
    });
}

int apds_dev_valu_popcount_peak(double* lane_ops_per_s) {
    return guarded([&] {
        APDS_REQUIRE(lane_ops_per_s, APDS_ERR_BAD_ARG, "null output");
        ctx().ws_reset();
        *lane_ops_per_s = valu_popcount_peak_device();
    });
}

int apds_dev_valu_peak(int mode, int waves_per_simd, double* lane_ops_per_s, double* cycles_per_inst, const char** name) {
    return guarded([&] {
        APDS_REQUIRE(mode >= 0 && mode < valu_peak_modes(), APDS_ERR_BAD_ARG, "mode out of range");
        APDS_REQUIRE(waves_per_simd >= 1 && waves_per_simd <= 8, APDS_ERR_BAD_ARG, "1 <= waves_per_simd <= 8");
        ctx().ws_reset();
        if (name) *name = valu_peak_mode_name(mode);
        valu_peak_device(mode, waves_per_simd, lane_ops_per_s, cycles_per_inst);
    });
}

int apds_dev_valu_peak_modes(void) { return valu_peak_modes(); }

}  // extern "C"
