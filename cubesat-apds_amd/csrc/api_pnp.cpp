// csrc/api_pnp.cpp — C-ABI entry points of homographier::pnp_solver_ransac (mod.rs:320-369).
#include "kernels.h"

using namespace apds;

extern "C" {

int apds_pnp_solver_ransac(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, int iter_count, float reproj_thres,
                           double confidence, int method, double* rvec, double* tvec, int32_t* inliers, int* n_inliers, int* found) {
    APDS_RANGE("apds_pnp_solver_ransac");
    return guarded([&] {
        APDS_REQUIRE(found, APDS_ERR_BAD_ARG, "null argument");
        *found = 0;
        ThreadCtx& c = ctx();
        c.ws_reset();
        *found = pnp_ransac_device(obj_xyz, img_xy, n, camera_intrinsic, iter_count, reproj_thres, confidence, method, rvec, tvec, inliers, n_inliers, c.stream);
    });
}

int apds_pnp_hypotheses(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, const int32_t* idx5, int n_samples,
                        int model_points, double* models) {
    return guarded([&] {
        ThreadCtx& c = ctx();
        c.ws_reset();
        pnp_hypotheses_device(obj_xyz, img_xy, n, camera_intrinsic, idx5, n_samples, model_points, models, c.stream);
    });
}

int apds_pnp_sqpnp(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, double* rvec, double* tvec, int* found) {
    return guarded([&] {
        APDS_REQUIRE(found, APDS_ERR_BAD_ARG, "null argument");
        *found = 0;
        *found = pnp_sqpnp_host(obj_xyz, img_xy, n, camera_intrinsic, rvec, tvec);
    });
}

int apds_pnp_ippe(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, double* rvec, double* tvec, int* found) {
    return guarded([&] {
        APDS_REQUIRE(found, APDS_ERR_BAD_ARG, "null argument");
        *found = 0;
        *found = pnp_ippe_host(obj_xyz, img_xy, n, camera_intrinsic, rvec, tvec);
    });
}

int apds_get_world_coordinates(const double* xy, int n, const double* dataset_gt, const double* elevation_gt, const double* elevation, int ew, int eh,
                               double* xyz) {
    return guarded([&] {
        APDS_REQUIRE(n >= 0 && (n == 0 || (xy && xyz)) && dataset_gt, APDS_ERR_BAD_ARG, "null argument");
        APDS_REQUIRE(!elevation_gt || (elevation && ew > 0 && eh > 0), APDS_ERR_BAD_ARG, "elevation geotransform without a raster");
        if (n == 0) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        double* dxy = c.alloc_n<double>((size_t)n * 2);
        double* dxyz = c.alloc_n<double>((size_t)n * 3);
        double* del = nullptr;
        HIP_CHECK(hipMemcpyAsync(dxy, xy, (size_t)n * 16, hipMemcpyHostToDevice, s));
        if (elevation_gt) {
            del = c.alloc_n<double>((size_t)ew * eh);
            HIP_CHECK(hipMemcpyAsync(del, elevation, (size_t)ew * eh * 8, hipMemcpyHostToDevice, s));
        }
        const int missing = world_coordinates_device(dxy, n, dataset_gt, elevation_gt, del, ew, eh, dxyz, s);
        HIP_CHECK(hipMemcpyAsync(xyz, dxyz, (size_t)n * 24, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (missing) fail(APDS_ERR_OUT_OF_RANGE, "an elevation lookup fell outside the elevation table (those points are NaN)");
    });
}

}  // extern "C"
