// csrc/api_pending.cpp — entry points whose kernels land in the next commits (replaced file by file).
#include "kernels.h"
using namespace apds;
extern "C" {
int apds_akaze_extract(const uint8_t*, int, int, int, size_t, int, apds_keypoint**, uint8_t**, int*, int*) {
    return guarded([&] { fail(APDS_ERR_INTERNAL, "apds_akaze_extract: kernels not built into this library yet"); });
}
int apds_dev_akaze_extract(const void*, int, int, int, size_t, int, void*, void*, int, int*, void*) {
    return guarded([&] { fail(APDS_ERR_INTERNAL, "apds_dev_akaze_extract: kernels not built into this library yet"); });
}
int apds_find_homography(const float*, const float*, int, int, double, double*, uint8_t*) {
    return guarded([&] { fail(APDS_ERR_INTERNAL, "apds_find_homography: kernels not built into this library yet"); });
}
int apds_find_homography_ex(const float*, const float*, int, int, double, int, double, double*, uint8_t*) {
    return guarded([&] { fail(APDS_ERR_INTERNAL, "apds_find_homography_ex: kernels not built into this library yet"); });
}
int apds_dev_find_homography(const void*, const void*, int, int, double, int, double, double*, void*, void*) {
    return guarded([&] { fail(APDS_ERR_INTERNAL, "apds_dev_find_homography: kernels not built into this library yet"); });
}
}
