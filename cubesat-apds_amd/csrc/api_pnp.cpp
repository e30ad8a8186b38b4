// csrc/api_pnp.cpp — C-ABI entry points of homographier::pnp_solver_ransac (mod.rs:320-369).
#include "kernels.h"

using namespace apds;

extern "C" {

int apds_pnp_solver_ransac(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, int iter_count, float reproj_thres,
                           double confidence, int method, double* rvec, double* tvec, int32_t* inliers, int* n_inliers, int* found) {
    return guarded([&] {
        APDS_REQUIRE(found, APDS_ERR_BAD_ARG, "null argument");
        *found = 0;
        ThreadCtx& c = ctx();
        c.ws_reset();
        *found = pnp_ransac_device(obj_xyz, img_xy, n, camera_intrinsic, iter_count, reproj_thres, confidence, method, rvec, tvec, inliers, n_inliers, c.stream);
    });
}

int apds_pnp_hypotheses(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, const int32_t* idx5, int n_samples,
                        double* models) {
    return guarded([&] {
        ThreadCtx& c = ctx();
        c.ws_reset();
        pnp_hypotheses_device(obj_xyz, img_xy, n, camera_intrinsic, idx5, n_samples, models, c.stream);
    });
}

}  // extern "C"
