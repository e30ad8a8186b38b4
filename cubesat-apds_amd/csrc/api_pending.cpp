// csrc/api_pending.cpp — entry points whose kernels land in the next commits (replaced file by file).
#include "kernels.h"
using namespace apds;
extern "C" {
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
