// csrc/kernels.h — host-callable launchers implemented in the .hip files.
#pragma once
#include <atomic>
#include "common.h"

namespace apds {

// match_hamming.hip
// backend: 0 = the configured one (APDS_MATCH_MFMA), 1 = vector ALU (hamming_topk_kernel), 2 = matrix cores (hamming_mfma_kernel, k <= 2)
void hamming_topk_device(const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, uint64_t* out, hipStream_t s, int backend = 0);
void merge_topk_device(const uint64_t* parts, int nparts, int nq, int k, uint64_t* out, hipStream_t s);
void take_first_columns_device(const uint64_t* in, int nq, int kin, int kout, uint64_t* out, hipStream_t s);
// hamming_mfma.hip: the same keys for k = 1, 2 from the FP4 matrix pipe (bit -> e2m1 operand, exact)
void hamming_mfma_topk_device(const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, uint64_t* out, hipStream_t s);
struct HmPlan {
    int q_tiles, splits, tiles_per_split;
};
struct HmTrain {   // a train set expanded once (256 bytes of fp4 operands per row + popcounts), in device memory of its own
    int device = 0;
    const void* src = nullptr;
    long long n = 0;
    void* rows = nullptr;
    float* pc = nullptr;
};
HmPlan hm_plan(int nq, long long nt);
long long hm_padded_rows(long long n);   // expanded train rows and their popcounts are padded to whole tiles: allocate this many rows / floats
void hm_expand_device(const void* rows64, long long n, bool query, void* out_fp4, float* pc, hipStream_t s);
void hm_scan_device(const void* q_fp4, const float* qpc, int nq, const void* t_fp4, const float* tpc, long long nt, const HmPlan& p, uint32_t index_base,
                    uint64_t* parts, hipStream_t s, const uint32_t* thr = nullptr, bool timed = true);
long long hm_sample_rows(long long nt);
void* hm_train_create(const void* rows64, long long n, hipStream_t s);
void hm_train_destroy(void* train);
void hamming_mfma_topk_train_device(const void* q, int nq, const void* train, uint32_t index_base, int k, uint64_t* out, hipStream_t s);
void topk_split_use_train(void* state, const void* hm_train);   // a pre-expanded train set for the state's matrix-core scans (or null)
// the scan of hamming_topk_device in three separately launched steps on a per-frame state object (k = 1, 2)
void* topk_split_create();
void topk_split_destroy(void* state);
void topk_split_prepass(void* state, const void* q, int nq, const void* t, long long nt, uint32_t index_base, int k, hipStream_t s);
void topk_split_scan(void* state, const void* q, const void* t, hipStream_t s);
void topk_split_merge(void* state, uint32_t index_base, uint64_t* out, hipStream_t s);
void pack_rows_device(const void* src, long long n, int desc_bytes, long long src_stride, void* dst, hipStream_t s);
int* scan_flags_device(const uint8_t* flags, int n, int** total_dev, hipStream_t s);
int ratio_filter_device(const uint64_t* keys, int nq, int k, float fs, apds_dmatch* out, hipStream_t s);
int cross_check_device(const uint64_t* train_best, long long n_train, int nq, apds_dmatch* out, hipStream_t s);
double valu_popcount_peak_device();
int valu_peak_modes();
const char* valu_peak_mode_name(int mode);
void valu_peak_device(int mode, int waves_per_simd, double* lane_ops_per_s, double* cycles_per_inst);

// akaze.hip
int akaze_extract_device(const void* img, int rows, int cols, int channels, size_t stride, int max_points, apds_keypoint* kps,
                         uint8_t* desc64, int capacity, hipStream_t s);
int akaze_extract_batch_device(const void* img, int n_img, size_t img_bstride, int rows, int cols, int channels, size_t stride, int max_points,
                               apds_keypoint* kps, uint8_t* desc64, int capacity, int* counts, hipStream_t s);
void points_from_matches_device(const apds_keypoint* kp1, int n1, const apds_keypoint* kp2, int n2, const apds_dmatch* m, int nm,
                                int bug_compatible, float* pts1, float* pts2, int* err_flag, hipStream_t s);
void rgba_to_bgra_device(const uint8_t* rgba, size_t n_pixels, uint8_t* bgra, hipStream_t s);

// ingest.hip
std::atomic<int>& match_lds_cap();   // match_hamming.hip: occupancy cap of the main Hamming scan
std::atomic<int>& last_scan_launch_lds();
void set_thread_scan_cap(int bytes);   // occupancy cap of the scans THIS thread launches (-1: the process-wide one)
int count_nonzero_device(const uint8_t* bytes, int n, int* count_dev, hipStream_t s);   // misc.hip: synchronises s
void band_merger_device(const float* r, const float* g, const float* b, size_t n, const double* mm, int bgra, uint8_t* out, hipStream_t s);
void warp_perspective_device(const uint8_t* src, int rows, int cols, const double* M, int dst_rows, int dst_cols, uint8_t* dst, hipStream_t s);
void warp_perspective_any_device(const void* src, int rows, int cols, int channels, int elem_bytes, const double* M, int dst_rows, int dst_cols, void* dst,
                                 hipStream_t s);

int world_coordinates_device(const double* xy, int n, const double* dgt_host, const double* egt_host, const double* elev, int ew, int eh, double* xyz,
                             hipStream_t s);

// homography.hip
int find_homography_device(const float* src, const float* dst, int n, int method, double thr, int max_iters, double confidence,
                           double* H_host, uint8_t* mask_dev, hipStream_t s);

// l2_match.hip / l2_screen.hip
void l2_row_norms_device(const float* x, long long n, int dim, float* out, hipStream_t s);
void l2_topk_device(const float* q, int nq, const float* t, long long nt, int dim, uint32_t index_base, int k, uint64_t* out, hipStream_t s);
bool l2_topk_screen_device(const float* q, int nq, const float* t, long long nt, int dim, uint32_t index_base, int k, uint64_t* out, hipStream_t s,
                           double* candidates_per_query);

// homography_rho.hip
int find_homography_rho_device(const float* src, const float* dst, int n, double thr, int max_iters, double confidence, double* H_host, uint8_t* mask_dev,
                               hipStream_t s);

// pnp.hip
int pnp_ransac_device(const double* obj_xyz, const double* img_xy, int n, const double* K, int iterations, float reproj_thr, double confidence, int method,
                      double* rvec, double* tvec, int32_t* inliers, int* n_inliers, hipStream_t s);
void pnp_hypotheses_device(const double* obj_xyz, const double* img_xy, int n, const double* K, const int32_t* idx5, int B, int model_points,
                           double* models_host, hipStream_t s);
int pnp_ippe_host(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec);
int pnp_sqpnp_host(const double* obj_xyz, const double* img_xy, int n, const double* K, double* rvec, double* tvec);

}  // namespace apds
