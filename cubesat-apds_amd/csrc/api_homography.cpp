// csrc/api_homography.cpp — C-ABI entry points of homographier::find_homography_mat (mod.rs:231-259).
#include "kernels.h"

using namespace apds;

namespace {
int find_h(const float* src, const float* dst, int n, int method, double thr, int max_iters, double confidence, double* H, uint8_t* mask) {
    APDS_REQUIRE(src && dst && H, APDS_ERR_BAD_ARG, "null argument");
    APDS_REQUIRE(n >= 4, APDS_ERR_ASSERT, "at least 4 point pairs are required");   // OpenCV: StsVecLengthErr
    ThreadCtx& c = ctx();
    c.ws_reset();
    hipStream_t s = c.stream;
    float* ds = c.alloc_n<float>((size_t)n * 2);
    float* dd = c.alloc_n<float>((size_t)n * 2);
    uint8_t* dm = c.alloc_n<uint8_t>(n);
    HIP_CHECK(hipMemcpyAsync(ds, src, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(dd, dst, (size_t)n * 8, hipMemcpyHostToDevice, s));
    const int found = find_homography_device(ds, dd, n, method, thr, max_iters, confidence, H, dm, s);
    if (mask) {
        HIP_CHECK(hipMemcpyAsync(mask, dm, n, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    return found;
}
}  // namespace

extern "C" {

int apds_find_homography_ex(const float* src, const float* dst, int n, int method, double thr, int max_iters, double confidence, double* H,
                            uint8_t* mask) {
    APDS_RANGE("apds_find_homography_ex");
    return guarded([&] {
        const int found = find_h(src, dst, n, method, thr, max_iters, confidence, H, mask);
        if (!found) fail(APDS_ERR_EMPTY, "no homography found (empty model)");
    });
}

int apds_find_homography(const float* src, const float* dst, int n, int method, double thr, double* H, uint8_t* mask) {
    return apds_find_homography_ex(src, dst, n, method, thr, 2000, 0.995, H, mask);   // OpenCV's 5-argument defaults
}

int apds_dev_find_homography(const void* src, const void* dst, int n, int method, double thr, int max_iters, double confidence, double* H,
                             void* mask_dev, void* stream) {
    APDS_RANGE("apds_dev_find_homography");
    return guarded([&] {
        ctx().ws_reset();
        const int found = find_homography_device(static_cast<const float*>(src), static_cast<const float*>(dst), n, method, thr, max_iters, confidence, H,
                                                 static_cast<uint8_t*>(mask_dev), pick_stream(stream));
        if (!found) fail(APDS_ERR_EMPTY, "no homography found (empty model)");
    });
}

}  // extern "C"
