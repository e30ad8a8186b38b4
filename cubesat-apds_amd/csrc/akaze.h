// csrc/akaze.h — shared declarations of the AKAZE pipeline (filters, keypoint kernels, host driver).
#pragma once
#include "common.h"

namespace apds {

struct GaussTaps {
    float k[5];   // k[0] centre, k[j] the two taps at distance j
};

static constexpr int AKAZE_MAX_LEVELS = 16;

// One evolution level (mirrors OpenCV's MEvolution / the oracle's Level)
struct LevelDesc {
    int w, h, octave, sublevel, sigma_size, border;
    float esigma, etime, ratio;
    int nsteps;
    float tau[64];
    // device planes
    float *Lt, *Ldet;
    float2* Lxy;   // (Lx, Ly) interleaved
    uint8_t* mask;      // extrema / suppression state
    uint8_t* mask_aux;  // scratch copy for the suppression rounds
    long long pix_offset;   // offset of this level in the level-major concatenated pixel index space
};

// level table handed to keypoint kernels by value
struct LevelTable {
    int n;
    int w[AKAZE_MAX_LEVELS], h[AKAZE_MAX_LEVELS], octave[AKAZE_MAX_LEVELS], sigma_size[AKAZE_MAX_LEVELS], border[AKAZE_MAX_LEVELS];
    float esigma[AKAZE_MAX_LEVELS], ratio[AKAZE_MAX_LEVELS];
    long long pix_offset[AKAZE_MAX_LEVELS + 1];
    const float* Lt[AKAZE_MAX_LEVELS];
    const float2* Lxy[AKAZE_MAX_LEVELS];   // (Lx, Ly) interleaved
    const float* Ldet[AKAZE_MAX_LEVELS];
    uint8_t* mask[AKAZE_MAX_LEVELS];
};

// akaze_filters.hip
void launch_gray(const void* img, int rows, int cols, int channels, size_t stride, float* out, hipStream_t s);
void launch_gauss(const float* src, float* dst, int w, int h, const GaussTaps& taps, int radius, hipStream_t s);
void launch_deriv_pair(const float* src, float* outA, float* outB, int w, int h, int sc, float kside, float kmid, hipStream_t s);
void launch_flow(const float* src, float* flow, int w, int h, const float* kptr, hipStream_t s);
void launch_smooth_flow(const float* src, float* smooth, float* flow, int w, int h, const GaussTaps& taps, const float* kptr, hipStream_t s);
void launch_kcontrast(const float* smooth, float* modg_tmp, int w, int h, unsigned int* hmax_bits, int* hist, float* k_oct, int n_oct, hipStream_t s,
                      bool gradient_done = false);
bool launch_base_strips(const void* img, int rows, int cols, int channels, size_t stride, const GaussTaps& g16, const GaussTaps& g10, float* Lt0, float* modg,
                        unsigned int* hmax_bits, bool want_modg, hipStream_t s);
void launch_nld_multi(const float* Lt, const float* Lf, float* Lnew, int w, int h, const float* step_sizes, int nsteps, hipStream_t s);
void launch_half_sample(const float* src, int sw, float* dst, int dw, int dh, hipStream_t s);
void launch_area_resize(const float* src, int sw, float* dst, int dw, int dh, const int* xofs, const float* xw, const int* xcnt, const int* yofs,
                        const float* yw, const int* ycnt, hipStream_t s);
void launch_doh_fused(const float* Lsmooth, float2* Lxy, float* Ldet, int w, int h, int sc, float kside, float kmid, int border, float thr, uint8_t* mask,
                      uint32_t* list, int* list_count, hipStream_t s);

// Test hook: when armed (per thread) the next akaze_extract_device copies one intermediate plane to the host.
// which: 0 Lt, 2 Lx, 3 Ly, 4 Ldet (f32), 7 keypoint mask after cross-level suppression (u8), 8 kcontrast (1 float)
struct AkazeDebugRequest {
    bool armed = false;
    int level = 0, which = 0;
    void* host_out = nullptr;
};
AkazeDebugRequest& akaze_debug_request();

// akaze_keypoints.hip (host driver: kernels.h declares akaze_extract_device)
void pack_desc61_device(const uint8_t* d64, int n, uint8_t* d61, hipStream_t s);

}  // namespace apds
