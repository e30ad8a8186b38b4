// csrc/api_akaze.cpp — C-ABI entry points of the extraction step (feature_extraction/src/lib.rs:61-92).
#include <cstdlib>
#include <cstring>

#include "akaze.h"
#include "kernels.h"

using namespace apds;

extern "C" {

// device image -> host keypoints + 61-byte descriptors (malloc'ed, the caller frees them with apds_free)
static void extract_to_host(ThreadCtx& c, hipStream_t s, const uint8_t* dimg, int rows, int cols, int channels, size_t dstride, int max_points,
                            apds_keypoint** kps, uint8_t** desc, int* n) {
    // strict 3x3 maxima are never adjacent: at most ceil(w/2)*ceil(h/2) per level, and the cap is max_points
    const int capacity = max_points;
    apds_keypoint* dk = c.alloc_n<apds_keypoint>(capacity);
    uint8_t* dd = c.alloc_n<uint8_t>((size_t)capacity * 64);
    const int K = akaze_extract_device(dimg, rows, cols, channels, dstride, max_points, dk, dd, capacity, s);
    apds_keypoint* hk = static_cast<apds_keypoint*>(std::malloc(std::max<size_t>(1, (size_t)K * sizeof(apds_keypoint))));
    uint8_t* hd = static_cast<uint8_t*>(std::malloc(std::max<size_t>(1, (size_t)K * APDS_DESC_BYTES)));
    if (!hk || !hd) {
        std::free(hk);
        std::free(hd);
        throw std::bad_alloc();
    }
    try {
        if (K) {
            uint8_t* d61 = c.alloc_n<uint8_t>((size_t)K * APDS_DESC_BYTES);
            pack_desc61_device(dd, K, d61, s);
            HIP_CHECK(hipMemcpyAsync(hk, dk, (size_t)K * sizeof(apds_keypoint), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(hd, d61, (size_t)K * APDS_DESC_BYTES, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
        std::free(hk);
        std::free(hd);
        throw;
    }
    *kps = hk;
    *desc = hd;
    *n = K;
}

int apds_akaze_extract(const uint8_t* img, int rows, int cols, int channels, size_t stride, int max_points, apds_keypoint** kps, uint8_t** desc,
                       int* n, int* desc_bytes) {
    return guarded([&] {
        APDS_REQUIRE(kps && desc && n && desc_bytes, APDS_ERR_BAD_ARG, "null output");
        *kps = nullptr;
        *desc = nullptr;
        *n = 0;
        *desc_bytes = APDS_DESC_BYTES;
        APDS_REQUIRE(img != nullptr && rows > 0 && cols > 0, APDS_ERR_ASSERT, "empty image");   // CV_Assert(!image.empty())
        APDS_REQUIRE(channels == 1 || channels == 3 || channels == 4, APDS_ERR_ASSERT, "image must have 1, 3 or 4 channels");
        APDS_REQUIRE(stride >= (size_t)cols * channels, APDS_ERR_ASSERT, "row stride smaller than a row");
        if (max_points <= 0) max_points = APDS_MAX_POINTS;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        const size_t row_bytes = (size_t)cols * channels;
        const size_t dstride = (row_bytes + 3) & ~size_t(3);
        uint8_t* dimg = c.alloc_n<uint8_t>(dstride * rows);
        HIP_CHECK(hipMemcpy2DAsync(dimg, dstride, img, stride, row_bytes, rows, hipMemcpyHostToDevice, s));
        extract_to_host(c, s, dimg, rows, cols, channels, dstride, max_points, kps, desc, n);
    });
}

// One tile of the preprocessor in one call (preprocessor/src/main.rs:258-277): the three f32 band windows go up once, band_merger
// writes BGRA (geotiff_extractor mod.rs:346-378 fused with homographier raster_to_mat mod.rs:183-197) and AKAZE reads it on the device.
int apds_tile_extract(const float* red, const float* green, const float* blue, int rows, int cols, size_t row_stride, const double* minmax6,
                      int max_points, apds_keypoint** kps, uint8_t** desc, int* n, int* desc_bytes) {
    return guarded([&] {
        APDS_REQUIRE(kps && desc && n && desc_bytes, APDS_ERR_BAD_ARG, "null output");
        *kps = nullptr;
        *desc = nullptr;
        *n = 0;
        *desc_bytes = APDS_DESC_BYTES;
        APDS_REQUIRE(red && green && blue && minmax6, APDS_ERR_BAD_ARG, "null argument");
        APDS_REQUIRE(rows > 0 && cols > 0, APDS_ERR_ASSERT, "empty tile");
        APDS_REQUIRE(row_stride >= (size_t)cols, APDS_ERR_ASSERT, "row stride smaller than a row");
        if (max_points <= 0) max_points = APDS_MAX_POINTS;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        const size_t px = (size_t)rows * cols;
        float* bands = c.alloc_n<float>(3 * px);
        const float* src[3] = {red, green, blue};
        for (int b = 0; b < 3; b++)
            HIP_CHECK(hipMemcpy2DAsync(bands + b * px, (size_t)cols * 4, src[b], row_stride * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice, s));
        uint8_t* dimg = c.alloc_n<uint8_t>(px * 4);
        band_merger_device(bands, bands + px, bands + 2 * px, px, minmax6, /*bgra=*/1, dimg, s);
        extract_to_host(c, s, dimg, rows, cols, 4, (size_t)cols * 4, max_points, kps, desc, n);
    });
}

// Test hook (tests/test_akaze_gpu.py): run the extraction and copy one intermediate plane of `level` to out_plane.
int apds_akaze_debug_plane(const uint8_t* img, int rows, int cols, int channels, size_t stride, int level, int which, void* out_plane) {
    AkazeDebugRequest& r = akaze_debug_request();
    r.armed = true;
    r.level = level;
    r.which = which;
    r.host_out = out_plane;
    apds_keypoint* k = nullptr;
    uint8_t* d = nullptr;
    int n = 0, nb = 0;
    const int rc = apds_akaze_extract(img, rows, cols, channels, stride, 0, &k, &d, &n, &nb);
    r.armed = false;
    apds_free(k);
    apds_free(d);
    return rc;
}

int apds_dev_akaze_extract(const void* img, int rows, int cols, int channels, size_t stride, int max_points, void* kps, void* desc64, int capacity,
                           int* n, void* stream) {
    return guarded([&] {
        APDS_REQUIRE(n && kps && desc64, APDS_ERR_BAD_ARG, "null output");
        ctx().ws_reset();
        *n = akaze_extract_device(img, rows, cols, channels, stride, max_points, static_cast<apds_keypoint*>(kps), static_cast<uint8_t*>(desc64), capacity,
                                  pick_stream(stream));
    });
}

}  // extern "C"
