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
    APDS_RANGE("apds_akaze_extract");
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
    APDS_RANGE("apds_tile_extract");
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

// Output rows per image of a batched extraction: the rule of the single-image call (capacity = max_points) unless the image cannot hold
// that many. Strict 3x3 maxima are never adjacent, so a level of w x h pixels has at most ceil(w/2) * ceil(h/2) of them; summed over every
// level of every octave (four levels per octave; octaves as the detector builds them: halved until width < 80 or height < 40). A bound
// below the true ceiling would fail a whole batch on a dense image that extracts fine alone.
static int batch_capacity(int rows, int cols, int max_points) {
    long long bound = 0;
    for (int o = 0, w = cols, h = rows; o < 4 && (o == 0 || (w >= 80 && h >= 40)); o++, w >>= 1, h >>= 1)
        bound += 4LL * ((w + 1) / 2) * ((h + 1) / 2);
    return (int)std::min<long long>(max_points, std::max<long long>(bound, 64));
}

// device results of a batch (capacity rows per image) -> malloc'ed host arrays holding the images' rows back to back
static void batch_results_to_host(ThreadCtx& c, hipStream_t s, const apds_keypoint* dk, const uint8_t* dd, int capacity, int n_images, const int* counts,
                                  apds_keypoint** kps, uint8_t** desc) {
    size_t total = 0;
    for (int i = 0; i < n_images; i++) total += (size_t)counts[i];
    apds_keypoint* hk = static_cast<apds_keypoint*>(std::malloc(std::max<size_t>(1, total * sizeof(apds_keypoint))));
    uint8_t* hd = static_cast<uint8_t*>(std::malloc(std::max<size_t>(1, total * APDS_DESC_BYTES)));
    if (!hk || !hd) {
        std::free(hk);
        std::free(hd);
        throw std::bad_alloc();
    }
    try {
        uint8_t* d61 = c.alloc_n<uint8_t>(std::max<size_t>(1, total * APDS_DESC_BYTES));
        size_t off = 0;
        for (int i = 0; i < n_images; i++) {
            const int K = counts[i];
            if (!K) continue;
            pack_desc61_device(dd + (size_t)i * capacity * 64, K, d61 + off * APDS_DESC_BYTES, s);
            HIP_CHECK(hipMemcpyAsync(hk + off, dk + (size_t)i * capacity, (size_t)K * sizeof(apds_keypoint), hipMemcpyDeviceToHost, s));
            off += (size_t)K;
        }
        if (total) HIP_CHECK(hipMemcpyAsync(hd, d61, total * APDS_DESC_BYTES, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
        std::free(hk);
        std::free(hd);
        throw;
    }
    *kps = hk;
    *desc = hd;
}

// B tiles of the preprocessor in one call (preprocessor/src/main.rs:227-245 spawns one task per tile; 258-277 is the per-tile chain): the
// band windows of all tiles go up (3 B strided copies), ONE band_merger pass writes the B BGRA images and the batched extraction runs
// every kernel once for all of them. red / green / blue: n_tiles pointers each (windows of one size, row_stride elements between rows).
int apds_tile_extract_batch(const float* const* red, const float* const* green, const float* const* blue, int n_tiles, int rows, int cols, size_t row_stride,
                            const double* minmax6, int max_points, apds_keypoint** kps, uint8_t** desc, int* counts, int* desc_bytes) {
    APDS_RANGE("apds_tile_extract_batch");
    return guarded([&] {
        APDS_REQUIRE(kps && desc && counts && desc_bytes, APDS_ERR_BAD_ARG, "null output");
        *kps = nullptr;
        *desc = nullptr;
        *desc_bytes = APDS_DESC_BYTES;
        APDS_REQUIRE(red && green && blue && minmax6, APDS_ERR_BAD_ARG, "null argument");
        APDS_REQUIRE(n_tiles >= 1 && n_tiles <= 4096, APDS_ERR_BAD_ARG, "batch must hold 1 .. 4096 tiles");
        APDS_REQUIRE(rows > 0 && cols > 0, APDS_ERR_ASSERT, "empty tile");
        APDS_REQUIRE(row_stride >= (size_t)cols, APDS_ERR_ASSERT, "row stride smaller than a row");
        if (max_points <= 0) max_points = APDS_MAX_POINTS;
        for (int i = 0; i < n_tiles; i++) counts[i] = 0;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        const size_t px = (size_t)rows * cols, all = px * n_tiles;
        float* bands = c.alloc_n<float>(3 * all);   // band-major: all tiles' red, then green, then blue
        const float* const* src[3] = {red, green, blue};
        for (int b = 0; b < 3; b++)
            for (int i = 0; i < n_tiles; i++) {
                APDS_REQUIRE(src[b][i] != nullptr, APDS_ERR_BAD_ARG, "null band window");
                HIP_CHECK(hipMemcpy2DAsync(bands + b * all + i * px, (size_t)cols * 4, src[b][i], row_stride * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice, s));
            }
        uint8_t* dimg = c.alloc_n<uint8_t>(all * 4);
        band_merger_device(bands, bands + all, bands + 2 * all, all, minmax6, /*bgra=*/1, dimg, s);
        const int capacity = batch_capacity(rows, cols, max_points);
        apds_keypoint* dk = c.alloc_n<apds_keypoint>((size_t)capacity * n_tiles);
        uint8_t* dd = c.alloc_n<uint8_t>((size_t)capacity * 64 * n_tiles);
        akaze_extract_batch_device(dimg, n_tiles, px * 4, rows, cols, 4, (size_t)cols * 4, max_points, dk, dd, capacity, counts, s);
        batch_results_to_host(c, s, dk, dd, capacity, n_tiles, counts, kps, desc);
    });
}

// n_images equal-sized host images in one call: one upload, one batched extraction (every kernel's grid covers all images), one download.
// Outputs: concatenated keypoints / 61-byte descriptors (image 0's rows first), counts[i] rows per image.
int apds_akaze_extract_batch(const uint8_t* imgs, int n_images, size_t image_stride, int rows, int cols, int channels, size_t stride, int max_points,
                             apds_keypoint** kps, uint8_t** desc, int* counts, int* desc_bytes) {
    APDS_RANGE("apds_akaze_extract_batch");
    return guarded([&] {
        APDS_REQUIRE(kps && desc && counts && desc_bytes, APDS_ERR_BAD_ARG, "null output");
        *kps = nullptr;
        *desc = nullptr;
        *desc_bytes = APDS_DESC_BYTES;
        APDS_REQUIRE(imgs != nullptr && rows > 0 && cols > 0, APDS_ERR_ASSERT, "empty image");
        APDS_REQUIRE(n_images >= 1 && n_images <= 4096, APDS_ERR_BAD_ARG, "batch must hold 1 .. 4096 images");
        APDS_REQUIRE(channels == 1 || channels == 3 || channels == 4, APDS_ERR_ASSERT, "image must have 1, 3 or 4 channels");
        APDS_REQUIRE(stride >= (size_t)cols * channels && image_stride >= stride * rows, APDS_ERR_ASSERT, "strides smaller than a row / an image");
        if (max_points <= 0) max_points = APDS_MAX_POINTS;
        for (int i = 0; i < n_images; i++) counts[i] = 0;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        const size_t row_bytes = (size_t)cols * channels;
        const size_t dstride = (row_bytes + 3) & ~size_t(3), dimg_bytes = dstride * rows;
        uint8_t* dimg = c.alloc_n<uint8_t>(dimg_bytes * n_images);
        if (stride == dstride && image_stride == dimg_bytes) {
            HIP_CHECK(hipMemcpyAsync(dimg, imgs, dimg_bytes * n_images, hipMemcpyHostToDevice, s));
        } else {
            for (int i = 0; i < n_images; i++)
                HIP_CHECK(hipMemcpy2DAsync(dimg + i * dimg_bytes, dstride, imgs + i * image_stride, stride, row_bytes, rows, hipMemcpyHostToDevice, s));
        }
        const int capacity = batch_capacity(rows, cols, max_points);
        apds_keypoint* dk = c.alloc_n<apds_keypoint>((size_t)capacity * n_images);
        uint8_t* dd = c.alloc_n<uint8_t>((size_t)capacity * 64 * n_images);
        akaze_extract_batch_device(dimg, n_images, dimg_bytes, rows, cols, channels, dstride, max_points, dk, dd, capacity, counts, s);
        batch_results_to_host(c, s, dk, dd, capacity, n_images, counts, kps, desc);
    });
}

int apds_dev_akaze_extract_batch(const void* imgs, int n_images, size_t image_stride, int rows, int cols, int channels, size_t stride, int max_points,
                                 void* kps, void* desc64, int capacity, int* counts, void* stream) {
    APDS_RANGE("apds_dev_akaze_extract_batch");
    return guarded([&] {
        APDS_REQUIRE(counts && kps && desc64, APDS_ERR_BAD_ARG, "null output");
        ctx().ws_reset(pick_stream(stream));
        akaze_extract_batch_device(imgs, n_images, image_stride, rows, cols, channels, stride, max_points, static_cast<apds_keypoint*>(kps),
                                   static_cast<uint8_t*>(desc64), capacity, counts, pick_stream(stream));
    });
}

int apds_dev_akaze_extract(const void* img, int rows, int cols, int channels, size_t stride, int max_points, void* kps, void* desc64, int capacity,
                           int* n, void* stream) {
    APDS_RANGE("apds_dev_akaze_extract");
    return guarded([&] {
        APDS_REQUIRE(n && kps && desc64, APDS_ERR_BAD_ARG, "null output");
        ctx().ws_reset(pick_stream(stream));
        *n = akaze_extract_device(img, rows, cols, channels, stride, max_points, static_cast<apds_keypoint*>(kps), static_cast<uint8_t*>(desc64), capacity,
                                  pick_stream(stream));
    });
}

}  // extern "C"
