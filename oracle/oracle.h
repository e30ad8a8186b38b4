/*
 * oracle/oracle.h — CPU restatement of the cubesat-APDS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker and the timed "port" CPU baseline. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The product
 * (cubesat-apds_amd/, libapds_hip.so) never includes, links or calls anything here.
 *
 * What it restates (reference file:line -> third-party algorithm behind it):
 *   feature_extraction/src/lib.rs:61-92   AKAZE::create(MLDB,0,3,0.001,4,4,PM_G2,max_points)+detectAndCompute
 *                                          -> OpenCV 4.8/4.9 features2d AKAZE (kaze/AKAZEFeatures.cpp,
 *                                             kaze/nldiffusion_functions.cpp, kaze/fed.cpp)
 *   feature_extraction/src/lib.rs:94-114  BFMatcher(NORM_HAMMING,false).knnMatch + Lowe ratio test
 *   feature_extraction/src/lib.rs:116-126 BFMatcher(NORM_HAMMING,true).match (batchDistance cross-check)
 *   feature_extraction/src/lib.rs:161-180 get_points_from_matches
 *   homographier/src/homographier/mod.rs:231-259 calib3d findHomography (0 / RANSAC / LMEDS)
 *   homographier/src/homographier/mod.rs:183-220 raster_to_mat (RGBA -> BGRA)
 *
 * OpenCV (crate opencv = "0.88.8", system OpenCV 4.8.x/4.9.0, unpinned: the reference has no
 * Cargo.lock) is NOT in /root/reference and not installed here, so the arithmetic is restated
 * from the published algorithm.
 *
 * PARITY STATUS
 *   pinned   : homography (reference test homography_success, mod.rs:437-472), raster_to_mat
 *              (mod.rs:556-603), Cmat layout/at_2d (mod.rs:515-553, 606-625).
 *   UNPINNED : AKAZE keypoints/descriptors and Hamming match values. The reference holds only
 *              count assertions (9079/9357 keypoints, 27 and 3228 matches; lib.rs:273,295,314)
 *              on two GeoTIFFs that are git-ignored and absent. "parity unpinned".
 */
#ifndef APDS_ORACLE_H
#define APDS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint layout (28 bytes) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} oracle_keypoint;

/* cv::DMatch layout (16 bytes) */
typedef struct {
    int32_t query_idx, train_idx, img_idx;
    float distance;
} oracle_dmatch;

typedef struct oracle_akaze oracle_akaze;

enum { ORACLE_PLANE_LT = 0, ORACLE_PLANE_LSMOOTH = 1, ORACLE_PLANE_LX = 2, ORACLE_PLANE_LY = 3,
       ORACLE_PLANE_LDET = 4, ORACLE_PLANE_LFLOW = 5, ORACLE_PLANE_MASK0 = 6 /* u8 extrema before cross-level */,
       ORACLE_PLANE_MASK1 = 7 /* u8 after cross-level */ };

void oracle_set_threads(int n);
int  oracle_get_threads(void);

/* feature_extraction/src/lib.rs:61-92. channels in {1,3,4}; stride in bytes. max_points<=0 -> no cap.
 * keep_planes!=0 keeps every intermediate plane for per-stage tests. Returns NULL on bad arguments. */
oracle_akaze* oracle_akaze_run(const uint8_t* img, int rows, int cols, int channels, size_t stride,
                               int max_points, int keep_planes);
int   oracle_akaze_num_keypoints(const oracle_akaze*);
const oracle_keypoint* oracle_akaze_keypoints(const oracle_akaze*);
const uint8_t* oracle_akaze_descriptors(const oracle_akaze*);   /* K x 61 */
int   oracle_akaze_desc_bytes(const oracle_akaze*);
int   oracle_akaze_num_levels(const oracle_akaze*);
float oracle_akaze_kcontrast(const oracle_akaze*);
/* info[8] = {w, h, octave, sublevel, sigma_size, border, nsteps, 0}; finfo[4] = {esigma, etime, ratio, kcontrast_at_level} */
void  oracle_akaze_level_info(const oracle_akaze*, int level, int* info, float* finfo);
int   oracle_akaze_level_tau(const oracle_akaze*, int level, float* tau_out, int cap);
const void* oracle_akaze_plane(const oracle_akaze*, int level, int which);
const float* oracle_akaze_gray(const oracle_akaze*);
void  oracle_akaze_free(oracle_akaze*);

/* Hamming brute force. Rows are desc_bytes long with the given byte strides. idx/dist are nq*k,
 * filled with -1 / INT_MAX where the train set has fewer than k rows. Ties: lower train index first. */
void oracle_knn_hamming(const uint8_t* q, int nq, size_t q_stride, const uint8_t* t, int nt, size_t t_stride,
                        int desc_bytes, int k, int32_t* idx, int32_t* dist);
/* lib.rs:94-114. Returns match count >= 0, or a negative OpenCV-style code (-211 when a query has < 2 neighbours). */
int oracle_get_knn_matches(const uint8_t* q, int nq, size_t q_stride, const uint8_t* t, int nt, size_t t_stride,
                           int desc_bytes, int k, float filter_strength, oracle_dmatch* out /* nq */);
/* lib.rs:116-126 (crossCheck=true). out holds up to nq entries. */
int oracle_get_bruteforce_matches(const uint8_t* q, int nq, size_t q_stride, const uint8_t* t, int nt, size_t t_stride,
                                  int desc_bytes, oracle_dmatch* out /* nq */);
/* lib.rs:161-180. bug_compatible!=0 reproduces the reference's img_idx / img1-twice behaviour.
 * Returns 0, or -211 on an out-of-range index. pts are 2*nm floats each. */
int oracle_get_points_from_matches(const oracle_keypoint* kp1, int n1, const oracle_keypoint* kp2, int n2,
                                   const oracle_dmatch* m, int nm, int bug_compatible, float* pts1, float* pts2);

/* mod.rs:231-259. method: 0 least squares, 4 LMEDS, 8 RANSAC. thr<=0 -> 3. max_iters/confidence: OpenCV
 * defaults are 2000 / 0.995. Returns 1 if a model was found (H filled, H[8]==1), 0 if none, <0 on error
 * (-215 bad args, -2 RHO unsupported). mask (n bytes) may be NULL. */
/* HomographyMethod::RHO (mod.rs:30): OpenCV rho.cpp restated (PROSAC + SPRT + LM refinement, binary32). 1 = model found. */
int oracle_rho_homography(const float* src_xy, const float* dst_xy, int n, double thr, int max_iters, double confidence, double* H,
                          uint8_t* mask);
int oracle_find_homography(const float* src_xy, const float* dst_xy, int n, int method, double thr,
                           int max_iters, double confidence, double* H, uint8_t* mask);
/* helpers exposed for GPU per-stage parity */
int  oracle_homography_4pt(const float* src_xy, const float* dst_xy, int count, double* H);   /* runKernel */
int  oracle_ransac_samples(const float* src_xy, const float* dst_xy, int n, int iters, int32_t* idx4 /* iters*4 */);

/* mod.rs:320-369 pnp_solver_ransac -> cv::solvePnPRansac(obj Point3d[n], img Point2d[n], K 3x3 f64, distCoeffs = zeros(4,1)
 * (mod.rs:344), useExtrinsicGuess false, iterations, reproj_thr, confidence, inliers, method). method: cv::SolvePnPMethod value
 * (1 = SOLVEPNP_EPNP, the reference's default). Returns 1 if a pose was found (rvec, tvec, inliers[0..*n_inliers) filled),
 * 0 if none, -215 for n < 4 / null arguments (mod.rs:627-638), -213 for methods other than ITERATIVE (0), EPNP (1), P3P (2) and
 * AP3P (5). n == 4 runs through the P3P kernel, as in OpenCV.
 * PARITY UNPINNED, see pnp_oracle.cpp. */
int oracle_solve_pnp_ransac(const double* obj_xyz, const double* img_xy, int n, const double* K, int iterations, float reproj_thr,
                            double confidence, int method, double* rvec, double* tvec, int32_t* inliers, int* n_inliers);
/* helpers exposed for GPU per-stage parity */
int  oracle_solve_pnp_epnp(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec);
int  oracle_solve_pnp_ippe(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec);    /* solvePnP(SOLVEPNP_IPPE), 1 = a pose */
int  oracle_solve_pnp_sqpnp(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec);   /* solvePnP(SOLVEPNP_SQPNP), 1 = a pose */
int  oracle_pnp_ransac_samples(int n, int iters, int32_t* idx5 /* iters*5 */);
int  oracle_pnp_hypothesis(const double* obj_xyz, const double* img_xy, const int32_t* idx5, const double* K, double* rvec, double* tvec);
int  oracle_pnp_ransac_samples4(int n, int iters, int32_t* idx4 /* iters*4 */);
int  oracle_pnp_p3p_hypothesis(const double* obj_xyz, const double* img_xy, const int32_t* idx4, const double* K, double* rvec, double* tvec);
int oracle_pnp_ap3p_hypothesis(const double* obj_xyz, const double* img_xy, const int32_t* idx4, const double* K, double* rvec, double* tvec);   /* SOLVEPNP_AP3P kernel */
void oracle_rodrigues(const double* in, int in_is_matrix, double* out);
void oracle_det_acos(const double* c, int n, double* out);

/* mod.rs:183-220: RGBA8 slice -> BGRA rows. Returns 0 or -1 (MatError::Unknown) if len != w*h. */
int oracle_raster_to_mat(const uint8_t* rgba, size_t n_pixels, int w, int h, uint8_t* bgra);

/* feature_database/src/elevationdb.rs:64-104 get_world_coordinates, batched: pixel (x, y) of the reference mosaic -> ECEF metres.
 * dataset_gt / elevation_gt: GDAL geotransforms (6 doubles); elevation_gt NULL -> height 0 (elevationdb.rs:74-77); elev: eh x ew f64.
 * Returns 0, -211 if an elevation lookup misses (Diesel NotFound in the reference; those points are NaN), -5 if elevation_gt is singular.
 * PARITY UNPINNED (GDAL / PROJ absent; see ingest_oracle.cpp). */
int oracle_world_coordinates(const double* xy, int n, const double* dataset_gt, const double* elevation_gt, const double* elev, int ew, int eh, double* xyz);
int oracle_invert_geotransform(const double* gt, double* out);

/* BASELINE config 3 (no reference call site; semantics of cv::BFMatcher(NORM_L2).knnMatch): dist = sqrtf(sum (q-t)^2) accumulated in
 * f32 in index order, k smallest, ties to the lower train index. PARITY UNPINNED. */
void oracle_knn_l2(const float* q, int nq, const float* t, int nt, int dim, int k, int32_t* idx, float* dist);

/* geotiff_extractor/src/image_extractor/mod.rs:346-378 band_merger (+ f32_to_u8 :410-422, gamma_correction :402-408):
 * three f32 bands + per-band min/max -> RGBA8. NaN or out-of-range -> 0; alpha 0 only when all three bands are NaN. */
void oracle_band_merger(const float* red, const float* green, const float* blue, size_t n, const double* minmax6, uint8_t* rgba);
float oracle_gamma_correction(float v, int* ok);
int oracle_f32_to_u8(float v, float mn, float mx, int* ok);

/* homographier mod.rs:271-300 warp_image_perspective: cv::warpPerspective(INTER_LINEAR, BORDER_CONSTANT (1,1,1,1)) on a
 * 4-channel u8 image; M maps src -> dst (it is inverted inside, as OpenCV does without WARP_INVERSE_MAP). Returns 0 / -1. */
int oracle_warp_perspective_8uc4(const uint8_t* src, int rows, int cols, const double* M, int dst_rows, int dst_cols, uint8_t* dst);
/* the same for `channels` (1..4) interleaved elements of elem_bytes 1 (u8, fixed-point bilinear) or 4 (f32, float bilinear) */
int oracle_warp_perspective_any(const void* src, int rows, int cols, int channels, int elem_bytes, const double* M, int dst_rows, int dst_cols, void* dst);

#ifdef __cplusplus
}
#endif
#endif
