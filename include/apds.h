/*
 * include/apds.h — C ABI of libapds_hip.so, the MI355X (gfx950) implementation of the cubesat-APDS
 * hot path: AKAZE feature extraction -> Hamming brute-force match -> RANSAC homography.
 *
 * Every entry point replaces one public item of the reference's Rust crates `feature_extraction`
 * and `homographier` (cited per function as /root/reference/<file>:<line>). The Rust shim that keeps
 * the crates' signatures and forwards here is in INTEGRATION.md / rust_shim/.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross this boundary;
 *   - return 0 on success, a negative OpenCV-style status otherwise (the reference surfaces
 *     opencv::Error{code,..}): APDS_ERR_* below; text via apds_last_error() (thread local);
 *   - inputs are borrowed for the duration of the call; variable-length outputs are allocated by
 *     the library and released with apds_free(); fixed-size outputs are caller allocated;
 *   - every function is re-entrant and may be called concurrently from many host threads (the
 *     reference calls extraction from a rayon pool with no lock: preprocessor/src/main.rs:227-245,277);
 *     each host thread gets its own HIP stream and device workspace;
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     APDS_ERR_NO_DEVICE.
 *
 * The "_dev" functions are the same operations on buffers already resident in HBM (device
 * pointers + an optional hipStream_t passed as void*); they are what bench.py times and what the
 * multi-GPU sharded matcher is built from.
 */
#ifndef APDS_H
#define APDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APDS_OK 0
#define APDS_ERR_INTERNAL (-2)      /* cv::Error::StsError  (HIP runtime failure, unsupported method) */
#define APDS_ERR_NOMEM (-4)         /* cv::Error::StsNoMem */
#define APDS_ERR_BAD_ARG (-5)       /* cv::Error::StsBadArg */
#define APDS_ERR_NO_DEVICE (-216)   /* cv::Error::GpuNotSupported: no HIP device / kernels not loadable */
#define APDS_ERR_OUT_OF_RANGE (-211)/* cv::Error::StsOutOfRange */
#define APDS_ERR_ASSERT (-215)      /* cv::Error::StsAssert (bad shapes, k < 1, too few points) */
#define APDS_ERR_NOT_IMPLEMENTED (-213) /* cv::Error::StsNotImplemented (a method of the reference surface that is not built) */
#define APDS_ERR_EMPTY (-1000)      /* no model found: the shim maps it to MatError::Empty (mod.rs:114-119,258) */

/* feature_extraction/src/lib.rs:12-13  MAX_POINTS_SHIFT / MAX_POINTS (twin: feature_database/src/keypointdb.rs:12) */
#define APDS_MAX_POINTS_SHIFT 18
#define APDS_MAX_POINTS ((1 << APDS_MAX_POINTS_SHIFT) - 1)
#define APDS_DESC_BYTES 61          /* 486-bit M-LDB */
#define APDS_DESC_STRIDE 64         /* device row pitch (one 64-byte line per descriptor) */

/* cv::KeyPoint, 28 bytes (fields read back by to_db_type, lib.rs:34-58) */
typedef struct apds_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} apds_keypoint;

/* cv::DMatch, 16 bytes */
typedef struct apds_dmatch {
    int32_t query_idx, train_idx, img_idx;
    float distance;
} apds_dmatch;

/* homographier/src/homographier/mod.rs:25-31  enum HomographyMethod */
enum { APDS_HOMOGRAPHY_DEFAULT = 0, APDS_HOMOGRAPHY_LMEDS = 4, APDS_HOMOGRAPHY_RANSAC = 8, APDS_HOMOGRAPHY_RHO = 16 };

/* ---- library ------------------------------------------------------------------------------ */
const char* apds_last_error(void);
void apds_free(void* p);
int apds_device_count(void);
int apds_set_device(int ordinal);             /* device used by the calling thread (default 0) */
const char* apds_build_info(void);            /* "gfx950 ..." */

/* ---- feature_extraction ------------------------------------------------------------------- */

/* lib.rs:61-92  akaze_keypoint_descriptor_extraction_def(img:&Mat, max_points:Option<i32>) -> ExtractedKeyPoint
 * img: rows x cols x channels u8 (channels 1 gray, 3 BGR, 4 BGRA), row pitch stride_bytes.
 * max_points <= 0 means None (-> APDS_MAX_POINTS). Outputs: *kps (n x 28 B), *desc (n x 61 B, rows packed),
 * both owned by the caller afterwards (apds_free). */
int apds_akaze_extract(const uint8_t* img, int rows, int cols, int channels, size_t stride_bytes, int max_points,
                       apds_keypoint** kps, uint8_t** desc, int* n, int* desc_bytes);
/* The same for n_images images of ONE size in one call (image i starts image_stride_bytes after image i-1): the batch goes through every
 * kernel's grid together, which is what makes small tiles cheap (a single small tile is launch-latency-bound). This is how a caller
 * that extracts one tile per task (preprocessor/src/main.rs:227-245,277) should hand its tiles over. Per-image results are exactly those
 * of apds_akaze_extract. Outputs: *kps / *desc hold the images' rows back to back (image 0 first), counts[i] = rows of image i. */
int apds_akaze_extract_batch(const uint8_t* imgs, int n_images, size_t image_stride_bytes, int rows, int cols, int channels, size_t stride_bytes,
                             int max_points, apds_keypoint** kps, uint8_t** desc, int* counts, int* desc_bytes);

/* lib.rs:94-114  get_knn_matches(origin_desc, target_desc, k, filter_strength) -> Vector<DMatch>
 * Hamming k-NN of each origin (query) row over the target (train) rows, then keep m[0] iff
 * m[0].distance < m[1].distance * filter_strength. Returns APDS_ERR_OUT_OF_RANGE when a query has fewer
 * than two neighbours (k < 2 or n_target < 2), like the reference's `i.get(1)?`. Rows are desc_bytes long, packed. */
int apds_get_knn_matches(const uint8_t* origin_desc, int n_origin, const uint8_t* target_desc, int n_target,
                         int desc_bytes, int k, float filter_strength, apds_dmatch** matches, int* n_matches);

/* lib.rs:116-126  get_bruteforce_matches(origin_desc, target_desc) -> Vector<DMatch>  (crossCheck = true) */
int apds_get_bruteforce_matches(const uint8_t* origin_desc, int n_origin, const uint8_t* target_desc, int n_target,
                                int desc_bytes, apds_dmatch** matches, int* n_matches);

/* BFMatcher::knnMatch itself (lib.rs:103), any k >= 1 (k <= 2 is the tuned path; above 16 the scan runs once per 16 neighbours): idx/dist are
 * n_query*k, -1 / INT32_MAX where fewer than k exist. */
int apds_knn_match(const uint8_t* query_desc, int n_query, const uint8_t* train_desc, int n_train, int desc_bytes,
                   int k, int32_t* idx, int32_t* dist);

/* lib.rs:161-180  get_points_from_matches. bug_compatible = 1 reproduces the reference exactly (img1 index taken
 * from img_idx, both outputs from img1); 0 is the intended gather (query_idx -> img1, train_idx -> img2).
 * pts1/pts2: n_matches x 2 floats, caller allocated. */
int apds_get_points_from_matches(const apds_keypoint* img1_kp, int n1, const apds_keypoint* img2_kp, int n2,
                                 const apds_dmatch* matches, int n_matches, int bug_compatible, float* pts1, float* pts2);

/* ---- homographier -------------------------------------------------------------------------- */

/* mod.rs:231-259  find_homography_mat(input, reference, method, reproj_threshold) -> (Cmat<f64>, Option<Cmat<u8>>)
 * method: APDS_HOMOGRAPHY_*; reproj_threshold <= 0 -> 3.0. H: 9 doubles row major (H[8] == 1). mask: n bytes or
 * NULL (the shim passes it for RANSAC / LMEDS only, mod.rs:253-257). APDS_ERR_EMPTY when no model is found. */
int apds_find_homography(const float* input_xy, const float* reference_xy, int n, int method, double reproj_threshold,
                         double* H, uint8_t* mask);
/* same with OpenCV's two defaulted arguments exposed (maxIters 2000, confidence 0.995) */
int apds_find_homography_ex(const float* input_xy, const float* reference_xy, int n, int method, double reproj_threshold,
                            int max_iters, double confidence, double* H, uint8_t* mask);

/* mod.rs:183-220  raster_to_mat(pixels:&[RGBA8], w, h) -> Cmat<Vec4b>: RGBA -> BGRA rows.
 * APDS_ERR_BAD_ARG (MatError::Unknown) when n_pixels != w*h. bgra: w*h*4 bytes, caller allocated. */
int apds_raster_to_mat(const uint8_t* rgba, size_t n_pixels, int w, int h, uint8_t* bgra);

/* ---- "next" rows either side of the path ------------------------------------------ */

/* geotiff_extractor/src/image_extractor/mod.rs:346-378 band_merger (f32_to_u8 :410-422, gamma_correction :402-408): three f32 bands
 * + min/max {red_min, red_max, green_min, green_max, blue_min, blue_max} -> n pixels of RGBA8 (bgra = 0, what band_merger returns) or
 * BGRA8 (bgra = 1: raster_to_mat, mod.rs:183-220, fused). NaN / out-of-range -> 0; alpha 0 only if all three bands are NaN. */
int apds_band_merger(const float* red, const float* green, const float* blue, size_t n, const double* minmax6, int bgra, uint8_t* out);
int apds_dev_band_merger(const void* red, const void* green, const void* blue, size_t n, const double* minmax6, int bgra, void* out, void* stream);
/* One preprocessor tile in one call: `to_rgb` of an equal-size window (geotiff_extractor/src/image_extractor/mod.rs:241-269, band_merger
 * :346-378) -> `raster_to_mat` (homographier/src/homographier/mod.rs:183-197) -> `akaze_keypoint_descriptor_extraction_def`
 * (feature_extraction/src/lib.rs:61-92), as preprocessor/src/main.rs:258-277 chains them. red/green/blue: rows x cols f32 windows with
 * `row_stride` elements between rows (they may point into the mosaic); the RGBA / BGRA image exists on the device only. Outputs as apds_akaze_extract. */
int apds_tile_extract(const float* red, const float* green, const float* blue, int rows, int cols, size_t row_stride, const double* minmax6,
                      int max_points, apds_keypoint** kps, uint8_t** desc, int* n, int* desc_bytes);
/* The same for n_tiles equal-sized tiles in one call (the reference spawns one task per tile, main.rs:227-245): red / green / blue are arrays
 * of n_tiles window pointers (one row_stride for all); ONE band_merger pass and ONE batched extraction (apds_akaze_extract_batch) serve all
 * tiles. Per-tile results are exactly those of apds_tile_extract. Outputs as apds_akaze_extract_batch. */
int apds_tile_extract_batch(const float* const* red, const float* const* green, const float* const* blue, int n_tiles, int rows, int cols, size_t row_stride,
                            const double* minmax6, int max_points, apds_keypoint** kps, uint8_t** desc, int* counts, int* desc_bytes);

/* homographier/src/homographier/mod.rs:271-300 warp_image_perspective: warpPerspective(src, M, size, INTER_LINEAR, BORDER_CONSTANT,
 * Scalar(1,1,1,1)). M (9 doubles) maps source to destination coordinates. The reference function is generic over the element type
 * (warp_image_perspective<T: DataType>): u8 elements with 1, 3 or 4 interleaved channels here (u8, Vec3b, Vec4b - the type the reference's
 * own caller warps), f32 elements (f32, Vec3f, Vec4f) through the _f32 entry. */
int apds_warp_perspective(const uint8_t* src, int rows, int cols, int channels, const double* M, int dst_rows, int dst_cols, uint8_t* dst);
int apds_warp_perspective_f32(const float* src, int rows, int cols, int channels, const double* M, int dst_rows, int dst_cols, float* dst);

/* homographier/src/homographier/mod.rs:320-369 pnp_solver_ransac(point_correspondences, camera_intrinsic, iter_count, reproj_thres,
 * confidence, dist_coeffs, method) -> Result<Option<PNPRANSACSolution>, MatError>: cv::solvePnPRansac with useExtrinsicGuess = false and
 * distCoeffs = zeros(4,1) (mod.rs:344 shadows the dist_coeffs argument with zeros, so it never reaches OpenCV and is not part of this ABI).
 * obj_xyz: n Point3d (ImgObjCorrespondence::obj_point, mod.rs:53-65), img_xy: n Point2d, camera_intrinsic: 3x3 f64 row major.
 * method: cv::SolvePnPMethod; the shim passes method.unwrap_or(SOLVEPNP_EPNP) (mod.rs:360). Built: APDS_SOLVEPNP_EPNP (RANSAC kernel
 * EPnP on 5 points), APDS_SOLVEPNP_P3P (Gao's P3P on 4 points; also the kernel OpenCV switches to when n == 4), APDS_SOLVEPNP_AP3P (Ke and
 * Roumeliotis' algebraic P3P on 4 points) - the final pose over the inliers is EPnP in these cases, as in OpenCV - and APDS_SOLVEPNP_ITERATIVE (EPnP kernel; final pose = solvePnP(ITERATIVE) over
 * the inliers WITHOUT an extrinsic guess, as the reference's use_extrinsic_guess = false makes it: a homography (planar object points) or
 * DLT (>= 6 points) start, then <= 20 Levenberg-Marquardt iterations on the reprojection error; with five non-planar inliers the RANSAC
 * model stays, as in solvePnPRansac), APDS_SOLVEPNP_SQPNP (EPnP kernel; final pose = Terzakis and Lourakis' SQPnP over the inliers, calib3d/sqpnp.cpp;
 * no pose in front of the camera -> *found = 0 with the RANSAC model in rvec / tvec, as solvePnPRansac leaves it), APDS_SOLVEPNP_DLS and
 * APDS_SOLVEPNP_UPNP (OpenCV 4 runs EPnP for both: identical to APDS_SOLVEPNP_EPNP), APDS_SOLVEPNP_IPPE_SQUARE (what solvePnPRansac makes of it:
 * four correspondences are solved by P3P directly like under every other flag; with more, the final solvePnP over the >= 5 inliers of the
 * EPnP RANSAC asserts npoints == 4 -> APDS_ERR_ASSERT, the reference's Err(MatError::Opencv); no consensus -> *found = 0), APDS_SOLVEPNP_IPPE
 * (EPnP kernel; final pose = Collins and Bartoli's plane-based solver, calib3d/ippe.cpp, over the inliers: object points moved to the plane
 * z = 0 about their centroid, Harker-O'Leary homography, the better of the two poses; inliers that are not coplanar within 1e-3 of their own
 * unit have no IPPE pose -> *found = 0 with the RANSAC model in rvec / tvec, as solvePnPRansac leaves it). Values past cv::SolvePnPMethod's
 * last member return APDS_ERR_NOT_IMPLEMENTED.
 * n < 4 -> APDS_ERR_ASSERT (mod.rs:627-638).
 * *found = 1: rvec[3], tvec[3], inliers[0..*n_inliers) filled (inliers: caller allocated, n ints); *found = 0: Ok(None). */
#define APDS_SOLVEPNP_ITERATIVE 0
#define APDS_SOLVEPNP_EPNP 1
#define APDS_SOLVEPNP_P3P 2
#define APDS_SOLVEPNP_DLS 3
#define APDS_SOLVEPNP_UPNP 4
#define APDS_SOLVEPNP_AP3P 5
#define APDS_SOLVEPNP_IPPE 6
#define APDS_SOLVEPNP_IPPE_SQUARE 7
#define APDS_SOLVEPNP_SQPNP 8
int apds_pnp_solver_ransac(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, int iter_count, float reproj_thres,
                           double confidence, int method, double* rvec, double* tvec, int32_t* inliers, int* n_inliers, int* found);

/* feature_database/src/elevationdb.rs:64-104 get_world_coordinates, batched (the object points pnp_solver_ransac consumes): pixel
 * (x, y) of the reference mosaic -> dataset geotransform -> elevation through the inverse elevation geotransform (row id of
 * elevationdb.rs:240) -> EPSG:4326 -> EPSG:4978 (ECEF metres). xy: n x 2, xyz: n x 3 doubles; geotransforms: GDAL's 6 doubles;
 * elevation_gt NULL = no elevation data -> height 0 (elevationdb.rs:74-77); elevation: eh x ew doubles, row major.
 * APDS_ERR_OUT_OF_RANGE if a lookup misses the elevation table (the reference returns Err; those points are NaN here),
 * APDS_ERR_BAD_ARG if elevation_gt is singular (the reference panics). */
int apds_get_world_coordinates(const double* xy, int n, const double* dataset_gt, const double* elevation_gt, const double* elevation, int ew, int eh,
                               double* xyz);

/* BASELINE config 3 (no reference call site: the reference matches Hamming only, lib.rs:101,121): brute-force L2 k-NN of float
 * descriptors (dim <= 128) as an MFMA distance GEMM with a fused top-k; semantics of BFMatcher(NORM_L2).knnMatch
 * (dist = sqrt(sum (q-t)^2), ties to the lower train index). k in {1,2}. idx/dist: n_query*k. */
int apds_l2_knn_match(const float* query, int n_query, const float* train, int n_train, int dim, int k, int32_t* idx, float* dist);
/* device form: out_keys = n_query*k uint64 (f32 bits of the SQUARED distance << 32 | train_index + index_base), ascending */
int apds_dev_l2_topk(const void* query, int n_query, const void* train, int64_t n_train, int dim, uint32_t index_base, int k, void* out_keys,
                     void* stream);
/* The same with a choice of arithmetic. APDS_L2_EXACT: every distance in f32 MFMA (what apds_dev_l2_topk does). APDS_L2_SCREEN: a bf16
 * MFMA screen (16x the f32 matrix rate) selects, with a proved error bound, every row that can be among the two nearest; those are
 * re-ranked with exactly the f32 arithmetic of the exact mode, so the returned keys are the exact mode's, bit for bit. The screen is built
 * for dim == 128, k == 2, 16-byte aligned rows; otherwise (or if the candidate buffer overflows) the exact kernel runs: *mode_used says
 * which one did; *candidates_per_query = re-ranked rows per query (both may be NULL). */
enum { APDS_L2_EXACT = 0, APDS_L2_SCREEN = 1 };
int apds_dev_l2_topk_ex(const void* query, int n_query, const void* train, int64_t n_train, int dim, uint32_t index_base, int k, int mode, void* out_keys,
                        void* stream, int* mode_used, double* candidates_per_query);

/* GPU-resident keypoint table: the reference's `keypoint` table and its access paths without Postgres.
 * insert: preprocessor/src/main.rs:296-324 (x,y lifted to level-of-detail-0 pixels: v * 2^lod + index * tile * 2^lod).
 * select: feature_database/src/keypointdb.rs:38-90, mode 0 by image id, 1 by level of detail, 2 by level of detail and bounding box
 * (floor(start) <= v <= ceil(end)); always ORDER BY response DESC LIMIT 262143 (keypointdb.rs:12). The selection stays on the
 * device as the table's "view": 64-byte descriptor rows ready to be a train set for apds_dev_hamming_topk. */
int apds_db_create(void** db, int64_t capacity);
int apds_db_destroy(void* db);
int64_t apds_db_rows(const void* db);
int apds_db_insert_image(void* db, const apds_keypoint* kps, const uint8_t* desc61, int n, int image_id, int level_of_detail, uint64_t column,
                         uint64_t row, uint64_t tile_w, uint64_t tile_h);
int apds_db_select(void* db, int mode, int value, float x_start, float y_start, float x_end, float y_end, int* n_out);
int apds_db_view(void* db, void** rows64_dev, void** kps_dev, void** row_ids_dev, void** image_ids_dev, int* n);
int apds_db_view_download(void* db, apds_keypoint* kps, uint8_t* desc61, int32_t* ids, int32_t* image_ids);
/* knnMatch (k in {1,2}) of host query descriptors against the current view; idx = position in the view (= trainIdx of the Vec the reference would hold) */
int apds_db_knn_match(void* db, const uint8_t* query_desc, int n_query, int desc_bytes, int k, int32_t* idx, int32_t* dist);

/* ---- multi-GPU: the descriptor DB row-sharded over the GPUs of one node (BASELINE config 5) -------------------------------
 * No counterpart upstream: the reference matches on one CPU (feature_extraction/src/lib.rs:94-126) against whatever train set
 * feature_database/src/keypointdb.rs:50-90 returned, and its only parallelism is the preprocessor's rayon pool
 * (preprocessor/src/main.rs:86-89,227-245). Here one rank (a process, or a thread) drives each GPU; rank r keeps rows
 * [index_base, index_base + n_rows) of the train set resident, every rank brings its own frame's queries, and apds_shard_knn returns to
 * every rank the top-k of ITS queries over the WHOLE train set: all-gather of the query rows -> local scan -> all-to-all of the
 * per-shard keys -> u64-min merge. The result equals apds_dev_hamming_topk over the unsharded rows bit for bit (ties to the lower
 * global row). All ranks must issue the collective calls (create, counts, knn, gather, exchange_merge, destroy) in the same order.
 *
 * Transports: APDS_TRANSPORT_RCCL (one rank per GPU; the id comes from apds_comm_id_create on rank 0 and reaches the other ranks by any
 * host-side means - a file, MPI, the parent process), APDS_TRANSPORT_LOOPBACK (the ranks are threads of one process, any number per
 * GPU: the identical choreography on a one-GPU box), APDS_TRANSPORT_HOST (the host program's communicator as two callbacks on HOST
 * buffers, e.g. gloo or MPI; device data is staged around them, so it synchronises with the host). */
#define APDS_COMM_ID_BYTES 128
typedef struct apds_comm_id {
    char bytes[APDS_COMM_ID_BYTES];
} apds_comm_id;
enum { APDS_TRANSPORT_RCCL = 0, APDS_TRANSPORT_LOOPBACK = 1, APDS_TRANSPORT_HOST = 2, APDS_TRANSPORT_DEVICE = 3 };
typedef struct apds_host_transport {
    void* user;
    /* recv = every rank's bytes_per_rank bytes, rank-major; return 0 on success */
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank);
    /* to rank p: send[send_off[p] .. + send_bytes[p]); from rank p: recv_bytes[p] bytes to recv + recv_off[p] */
    int (*all_to_all)(void* user, const void* send, const size_t* send_off, const size_t* send_bytes, void* recv, const size_t* recv_off,
                      const size_t* recv_bytes);
} apds_host_transport;
/* APDS_TRANSPORT_DEVICE: the host program's communicator as two callbacks on DEVICE buffers, ordered on the hipStream_t they are given (e.g.
 * torch.distributed's device collectives over its own RCCL communicator: the second way onto xGMI if the library's own communicator cannot be
 * set up beside the host's). Passed to apds_shard_create through the `host` argument (same layout, one more argument per callback). */
typedef struct apds_device_transport {
    void* user;
    int (*all_gather)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
    int (*all_to_all)(void* user, const void* send_dev, const size_t* send_off, const size_t* send_bytes, void* recv_dev, const size_t* recv_off,
                      const size_t* recv_bytes, void* stream);
} apds_device_transport;
/* A fresh communicator id for `transport` (RCCL: ncclGetUniqueId; loopback: a process-unique name; host: zeros). Called by ONE rank. */
int apds_comm_id_create(int transport, apds_comm_id* id);
/* Collective. The shard lives on the calling thread's device (apds_set_device first). rows64_dev: n_rows x 64-byte rows, borrowed for
 * the life of the handle. index_base = global index of the shard's first row. id: RCCL / loopback; host: the host transport's callbacks
 * (APDS_TRANSPORT_DEVICE: a pointer to an apds_device_transport, cast). */
int apds_shard_create(void** shard, int rank, int world, int transport, const apds_comm_id* id, const apds_host_transport* host,
                      const void* rows64_dev, int64_t n_rows, uint32_t index_base);
int apds_shard_destroy(void* shard);
/* any output may be NULL; *rccl_version = ncclGetVersion() of the RCCL this library runs on */
int apds_shard_info(const void* shard, int* rank, int* world, int64_t* n_rows, uint32_t* index_base, const char** transport_name, int* rccl_version);
/* Collective: every rank's query count of one frame, as host ints (counts[world]). Synchronises `stream` on the RCCL transport. */
int apds_shard_counts(void* shard, int n_query, int* counts, void* stream);
/* Collective: out_keys_dev = n_query x k uint64 ((distance << 32) | global row; 0xFFFF... when absent) for THIS rank's queries over the
 * whole train set. counts: every rank's n_query (apds_shard_counts), or NULL to exchange them inside the call. Any k >= 1 the
 * single-device scan serves (k <= 2 tuned, above 16 in pages of 16); the exchange buffers bound it at 4096. */
int apds_shard_knn(void* shard, const void* q_rows64_dev, int n_query, const int* counts, int k, void* out_keys_dev, void* stream);
/* Strong-scaling (latency) form, SURVEY 8e's literal shape. Collective: ONE frame whose n_query queries are the same on every rank -
 * root < 0: every rank passes them (each extracted the frame itself: ~1.5 ms, no communication); root >= 0: rank `root`'s rows are broadcast
 * first, the other ranks' q_rows64_dev is ignored (n_query must still be the same number everywhere). Every rank scans its shard (1 / world of
 * the rows), the [n_query, k] key lists are ALL-GATHERED and merged by u64 min: out_keys_dev = n_query x k keys of the whole frame over the
 * whole train set on EVERY rank, equal to apds_dev_hamming_topk over the unsharded rows bit for bit. */
int apds_shard_knn_replicated(void* shard, const void* q_rows64_dev, int n_query, int root, int k, void* out_keys_dev, void* stream);
/* The same in three steps on per-frame exchange slots, for pipelines that keep two frames in flight: frame i+1's gather (collective) may be
 * issued - on another stream - before frame i's exchange_merge (collective), so it travels under frame i's scan (no collective). */
int apds_shard_slot_create(void* shard, int max_queries_per_rank, int kmax, void** slot);
int apds_shard_slot_destroy(void* shard, void* slot);
int apds_shard_gather(void* shard, void* slot, const void* q_rows64_dev, int n_query, const int* counts, void* stream);
int apds_shard_scan(void* shard, void* slot, int k, void* stream);
int apds_shard_exchange_merge(void* shard, void* slot, int k, void* out_keys_dev, void* stream);
/* SURVEY §8(b) "apds_db_create / append / shard / destroy": this rank's block of the resident keypoint table's current view (every rank
 * holds the same table and selection; rank r keeps rows [r n / world, (r + 1) n / world) of the view) as a shard; train indices stay the
 * positions in the view, i.e. what apds_db_knn_match returns. */
int apds_db_shard(void* db, int rank, int world, int transport, const apds_comm_id* id, const apds_host_transport* host, void** shard);

/* ---- the streamed frame pipeline ------------------------------------------------------------------------------------------------------
 * frame -> apds_dev_akaze_extract -> Hamming top-2 against the resident train rows -> ratio test (lib.rs:107-111) -> matched points
 * (lib.rs:161-180, the intended gather) -> find_homography_mat (mod.rs:231-259), software-pipelined over a stream of frames by host
 * threads INSIDE the library: two extraction workers on alternate frames | the match (matrix-core matcher, the default: one stream, against
 * a copy of the train rows expanded to FP4 operands once at create - 256 bytes per row on top of the caller's 64; vector-ALU matcher:
 * threshold pre-pass, main scan and record merge of consecutive frames on three streams; with a shard handle: the query gather of frame
 * i+1 issued before the key exchange of frame i) |
 * ratio filter + points + homography, each with its own HIP stream and device workspace. This is the composed path the north-star metric
 * (frames/s) is measured on; a host needs four calls. The reference chains the steps only inside unit tests (lib.rs:197-249); its
 * production caller runs extraction from a rayon pool without a lock (preprocessor/src/main.rs:227-245), which is what the workers mirror.
 * Results leave in frame order and equal, frame by frame, what the one-call entry points give (tests/cpp/pipeline_test.cpp). */
typedef struct apds_pipeline_params {
    int rows, cols, channels;   /* frame geometry: every frame of one pipeline has it (channels 1, 3 or 4) */
    int max_points;             /* <= 0: APDS_MAX_POINTS (lib.rs:12-13) */
    int n_slots;                /* frames in flight; 0 = 6 (never fewer than two per extraction worker) */
    int extract_workers;        /* host threads extracting alternate frames; 0 = 2 */
    float filter_strength;      /* Lowe ratio of get_knn_matches (lib.rs:107-111; the reference's test uses 0.3, lib.rs:222) */
    int homography_method;      /* APDS_HOMOGRAPHY_* (mod.rs:25-31) */
    double reproj_threshold;    /* <= 0: 3.0 (mod.rs:248) */
    int max_iters;              /* <= 0: 2000 (OpenCV's default) */
    double confidence;          /* outside (0, 1): 0.995 (OpenCV's default) */
    int timing;                 /* 1: HIP-event timing of the stage kernels on their launch streams (apds_pipeline_stats) */
    int match_lds_cap;          /* > 0: occupancy cap of this pipeline's scans from the first frame on (see apds_dev_match_lds_cap); 0: the starvation watch decides */
    void* match_stream;         /* optional stream for the main scan (e.g. apds_stream_create with a CU mask); NULL = the pipeline's own */
    double debug_extract_delay_ms; /* test hook: every extraction is handed on this much late (exercises the starvation watch) */
} apds_pipeline_params;
typedef struct apds_frame_result {
    int64_t frame;              /* the number apds_pipeline_submit gave the frame */
    int status;                 /* APDS_OK, or the status of the stage that failed for THIS frame (the pipeline goes on; text: apds_last_error after the poll) */
    int n_keypoints, n_matches, n_inliers;
    int homography_found;       /* 0: fewer than four matches, or no model (MatError::Empty, mod.rs:258) */
    double H[9];                /* row major, H[8] == 1 */
} apds_frame_result;
typedef struct apds_pipeline_counters {
    int64_t frames_submitted, frames_done;
    /* summed HIP-event times and launch counts since the last reset (params.timing = 1): the main scan, the threshold pre-pass, whole
     * extractions (wall span on their stream), RANSAC scoring */
    double hamming_topk_ms, hamming_topk_sample_ms, akaze_extract_ms, ransac_score_ms;
    int hamming_topk_launches, hamming_topk_sample_launches, akaze_extract_calls, ransac_score_launches;
    /* idle time of the match stream in front of each frame's main scan (the starvation watch's measurement) */
    double match_gap_mean_ms;
    int match_gaps;
    float match_gaps_first_ms[16];
    int match_lds_cap_bytes, match_lds_cap_set_at_frame;   /* 0 / -1: the watch never capped */
    float match_lds_cap_gaps_ms[6];                        /* the six gaps that made it cap */
    int extract_workers, slots, split_scan, world;
} apds_pipeline_counters;
#define APDS_PIPELINE_NOT_READY 1   /* apds_pipeline_poll: the next frame (in submission order) is not finished, or nothing is in flight */
/* The train set: db_rows64_dev = n_rows x 64-byte rows whose global indices start at index_base (one GPU), or `shard` = an apds_shard_*
 * handle (the rows arguments are then ignored; every rank must submit the same number of frames, and all collective calls of that handle
 * come from the pipeline from then on). db_kps_dev: the keypoints of ALL train rows (n_db_total x 28 bytes, indexed by global row: the
 * matched points' coordinates). Everything is borrowed until apds_pipeline_destroy and must not change meanwhile (the train rows are resident:
 * the pipeline, like a shard handle, keeps derived copies of them). The pipeline lives on the calling thread's device. */
int apds_pipeline_create(void** pipe, const void* db_rows64_dev, int64_t n_rows, uint32_t index_base, void* shard, const void* db_kps_dev, int64_t n_db_total,
                         const apds_pipeline_params* params);
/* Hands one frame over (on_device 0: host memory, uploaded by an extraction worker on its own stream - pinned memory makes the copy
 * overlap; 1: device memory) and returns at once unless every slot is in flight (then it blocks until one is free). The frame must stay
 * valid until its result has been polled. *frame_id (may be NULL) = its number, counted from 0. One submitting thread at a time. */
int apds_pipeline_submit(void* pipe, const void* frame, size_t stride_bytes, int on_device, int64_t* frame_id);
/* The next result in submission order. wait 0: APDS_PIPELINE_NOT_READY if that frame is not finished; wait 1: blocks until it is (returns
 * APDS_PIPELINE_NOT_READY only when no frame is in flight). A negative status = the pipeline itself has failed (every later call repeats it). */
int apds_pipeline_poll(void* pipe, apds_frame_result* result, int wait);
/* Counters and (params.timing) kernel times. Waits for the frames in flight to finish first. reset 1: the sums start again from zero. */
int apds_pipeline_stats(void* pipe, apds_pipeline_counters* out, int reset);
/* Drains the frames in flight, joins the workers, releases streams and buffers. */
int apds_pipeline_destroy(void* pipe);

/* ---- device-resident API ------------------------------------------------------------------- */
/* All pointers below are HIP device pointers. stream: hipStream_t or NULL (the thread's own stream).
 * Calls are asynchronous on that stream unless they return a count to the host; apds_dev_akaze_extract(_batch) returns as soon as the
 * count is known, with the orientation / descriptor kernels still queued on the stream: consumers order themselves on that stream (or an
 * event recorded on it), as with any asynchronous call. */

/* Pack n rows of desc_bytes (<= 64) bytes into 64-byte rows (zero padded). */
int apds_dev_pack_descriptors(const void* src_rows, int64_t n, int desc_bytes, int64_t src_stride, void* dst_rows64, void* stream);

/* Hamming top-k (k >= 1) of n_query rows against n_train rows, both 64-byte pitch. k in {1, 2} - all that lib.rs:94-126 consumes - runs on
 * the FP4 matrix pipe (bits as e2m1 operands, exact integer distances; the train rows are expanded to 256-byte operand rows in the calling
 * thread's workspace per call - apds_dev_match_backend); every other k on the vector ALU, above 16 one pass per 16 neighbours.
 * out_keys: n_query*k uint64 = (distance << 32) | (train_index + index_base), ascending; 0xFFFF... when absent.
 * Ordering equals BFMatcher's: by distance, ties to the lower train index. */
int apds_dev_hamming_topk(const void* query_rows64, int n_query, const void* train_rows64, int64_t n_train,
                          uint32_t index_base, int k, void* out_keys, void* stream);
/* apds_dev_hamming_topk in three separately launched steps (k = 1 or 2), for pipelines that keep several frames in flight: the threshold
 * pre-pass over the first train rows, the main scan, and the merge of its per-chunk records. The intermediate buffers live in a state
 * object (one per frame in flight; grow-only device memory) instead of the calling thread's workspace, so frame i + 1's pre-pass may run
 * on another stream while frame i's scan is on the GPU. Order per frame: prepass -> scan -> merge, each after the previous one on the GPU
 * (same stream, or events); q / t / n as given to the pre-pass must stay valid until the merge has run. Result == apds_dev_hamming_topk. */
int apds_dev_topk_state_create(void** state);
int apds_dev_topk_state_destroy(void* state);
int apds_dev_topk_prepass(void* state, const void* query_rows64, int n_query, const void* train_rows64, int64_t n_train, uint32_t index_base, int k,
                          void* stream);
int apds_dev_topk_scan(void* state, const void* query_rows64, const void* train_rows64, void* stream);
int apds_dev_topk_merge(void* state, uint32_t index_base, void* out_keys, void* stream);
/* Merge `parts` candidate lists (each n_query*k keys, e.g. gathered from DB shards) into the global top-k. */
int apds_dev_merge_topk(const void* keys_parts, int parts, int n_query, int k, void* out_keys, void* stream);
/* Occupancy cap of the main Hamming scan, process-wide: the kernel requests `bytes` of (unused) dynamic LDS per workgroup, which bounds
 * the workgroups resident per CU (160 KB / bytes; 55000 -> two). A pipeline that overlaps the scan with short kernels of other stages
 * sets it when those kernels cannot get onto the GPU (the scan alone is ~1.5 % slower with the cap). 0 = none (default, or
 * APDS_MATCH_LDS_CAP). The value is process-wide: a caller that sets it for its own launches restores *previous afterwards. *previous (may be NULL) receives the old value. */
int apds_dev_match_lds_cap(int bytes, int* previous);
/* Test hook: the dynamic-LDS request of the most recent scan launch of the process (-1 before the first); equals the cap in force for
 * every kernel variant (all tile widths, all k, persistent grid). */
int apds_dev_match_last_launch_lds(int* bytes);
/* apds_dev_hamming_topk on a named backend, whatever APDS_MATCH_MFMA says: 0 = the configured one, 1 = vector ALU (xor + popcount,
 * any k), 2 = matrix cores (k <= 2). The keys are the same bit for bit; bench.py times both in one run and tests compare them in one process. */
int apds_dev_hamming_topk_backend(const void* query_rows64, int n_query, const void* train_rows64, int64_t n_train, uint32_t index_base, int k,
                                  void* out_keys, int backend, void* stream);
/* Which kernel serves k <= 2 (everything lib.rs:94-126 consumes): *matrix_cores = 1: hamming_mfma_kernel - bits as FP4 (e2m1) operands of
 * v_mfma_scale_f32_16x16x128_f8f6f4, exact integer distances (default); 0: hamming_topk_kernel, xor + popcount on the vector ALU
 * (APDS_MATCH_MFMA=0; also what serves every k > 2). The keys are the same bit for bit. */
int apds_dev_match_backend(int* matrix_cores);
/* Lowe ratio filter on merged keys (k >= 2): writes compacted matches in query order, count to *n_matches (host). */
int apds_dev_ratio_filter(const void* keys, int n_query, int k, float filter_strength, void* out_matches, int* n_matches, void* stream);
/* Cross-check: given for every train row its best query key (from apds_dev_hamming_topk with roles swapped, k=1),
 * produce matches in query order. */
int apds_dev_cross_check(const void* train_best_keys, int64_t n_train, int n_query, void* out_matches, int* n_matches, void* stream);

/* AKAZE on a device image. Results stay on the device: kps (capacity x 28 B), desc64 (capacity x 64 B).
 * Returns the keypoint count in *n (host). capacity >= min(max_points, APDS_MAX_POINTS). */
int apds_dev_akaze_extract(const void* img, int rows, int cols, int channels, size_t stride_bytes, int max_points,
                           void* kps, void* desc64, int capacity, int* n, void* stream);
/* Batched form on device images: image i's keypoints land at kps + i * capacity rows, its descriptors at desc64 + i * capacity * 64 bytes,
 * its count in counts[i] (host). capacity (rows per image) >= the largest count. */
int apds_dev_akaze_extract_batch(const void* imgs, int n_images, size_t image_stride_bytes, int rows, int cols, int channels, size_t stride_bytes,
                                 int max_points, void* kps, void* desc64, int capacity, int* counts, void* stream);

/* gather matched coordinates on the device: pts1/pts2 n_matches x 2 float */
int apds_dev_points_from_matches(const void* kp1, int n1, const void* kp2, int n2, const void* matches, int n_matches,
                                 int bug_compatible, void* pts1, void* pts2, void* stream);

/* RANSAC / LMEDS / least-squares homography on device point lists. H (9 doubles) and found flag go to the host. */
int apds_dev_find_homography(const void* input_xy, const void* reference_xy, int n, int method, double reproj_threshold,
                             int max_iters, double confidence, double* H, void* mask_dev, void* stream);

/* Streams for callers that overlap stages. cu_mask (optional): bit i set = compute unit i may run the stream's kernels
 * (hipExtStreamCreateWithCUMask); priority 0 normal, -1 high (ignored when a mask is given). */
int apds_stream_create(int priority, const uint32_t* cu_mask, int cu_mask_words, void** stream);
int apds_stream_destroy(void* stream);

/* Device memory and copies on the calling thread's device, so that a host which keeps buffers resident (apds_dev_*, apds_shard_*) needs
 * no HIP binding of its own. upload is asynchronous on `stream` (the host buffer must stay valid until the stream has passed it; pinned
 * memory makes it a true async copy); download returns when the bytes are in dst_host. stream NULL = the thread's own stream. */
int apds_dev_alloc(size_t bytes, void** ptr);
int apds_dev_release(void* ptr);
int apds_dev_upload(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int apds_dev_download(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int apds_stream_synchronize(void* stream);

/* Releases the calling thread's HIP stream and device workspace (they are created lazily by the first call on a thread and
 * otherwise live as long as the thread; nothing is freed from thread-exit destructors, which may run after the HIP runtime has
 * shut down). Call it before a worker thread that used the library exits; the thread may use the library again afterwards. */
int apds_thread_release(void);
/* Host threads that currently hold a library context (stream + workspace): a lone caller lets the extraction use side streams
 * (apds_dev_akaze_extract forks its Hessian kernels only then). Diagnostic. */
int apds_live_contexts(void);
/* apds_thread_release keeps the thread's device workspace in a process-wide cache (a later thread reuses it instead of allocating);
 * this frees everything in that cache. */
int apds_release_cached_memory(void);

/* Test hook: run apds_akaze_extract and copy one intermediate plane of evolution level `level` to out_plane
 * (which: 0 Lt, 2 Lx, 3 Ly, 4 Ldet as f32 w*h; 7 keypoint mask after cross-level suppression as u8 w*h; 8 contrast factor, 1 float). */
int apds_akaze_debug_plane(const uint8_t* img, int rows, int cols, int channels, size_t stride_bytes, int level, int which, void* out_plane);

/* Test hook: the pose (rvec, tvec: 6 doubles per sample) of n_samples explicit samples as the RANSAC kernels compute them:
 * model_points 5 = EPnP on 5 correspondences, 4 = P3P on 4 (three solve, the fourth ranks; NaNs when there is no pose), 40 = AP3P on 4. */
int apds_pnp_hypotheses(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, const int32_t* samples, int n_samples,
                        int model_points, double* models);

/* Test hook: cv::solvePnP(..., SOLVEPNP_SQPNP) alone on n >= 3 correspondences - the final pose apds_pnp_solver_ransac computes over
 * its inliers when the caller names APDS_SOLVEPNP_SQPNP (mod.rs:327,359). Host arithmetic only: one 9 x 9 problem per call whatever n.
 * *found = 0: no pose (degenerate points, or none in front of the camera). */
int apds_pnp_sqpnp(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, double* rvec, double* tvec, int* found);
/* Test hook: cv::solvePnP(..., SOLVEPNP_IPPE) alone on n >= 4 correspondences (host arithmetic); *found = 0: the object points are not coplanar. */
int apds_pnp_ippe(const double* obj_xyz, const double* img_xy, int n, const double* camera_intrinsic, double* rvec, double* tvec, int* found);

/* Measurement helpers used by bench.py (not part of the reference surface). */
/* Register-only xor+popcount loop: returns measured lane-ops/s (32-bit xor + bcnt counted as 2 ops). */
int apds_dev_valu_popcount_peak(double* lane_ops_per_s);
/* Calibration of that denominator: the same register-only loop with ONE instruction kind (`mode`, 0 <= mode < apds_dev_valu_peak_modes():
 * integer xor / bcnt / add / and / xor3 and FP32 fma / add / mul / pk_fma / pk_add; *name = the mnemonic) at `waves_per_simd` (1..8) resident
 * waves per SIMD. Returns wall-clock lane-ops/s (a packed instruction counts two) and s_memtime cycles per wave-instruction per SIMD. */
int apds_dev_valu_peak(int mode, int waves_per_simd, double* lane_ops_per_s, double* cycles_per_inst, const char** name);
int apds_dev_valu_peak_modes(void);
/* Time of the last hamming_topk main kernel launched by this thread, measured with hipEvents on its stream (ms). */
int apds_dev_last_kernel_ms(const char* which, float* ms, int* launches);
int apds_dev_timing_enable(int on);

#ifdef __cplusplus
}
#endif
#endif /* APDS_H */
