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

// Batched launches (gridDim.z = images): every image of a batch owns an identical workspace slab `bstride` bytes after the previous
// one, so a plane of image blockIdx.z is the plane of image 0 shifted by blockIdx.z * bstride (akaze_keypoints.hip lays the slab
// out). A single image is a batch of one (blockIdx.z = 0).
#ifdef __HIPCC__
template <class T>
__device__ __forceinline__ T* bofs(T* p, size_t bstride) {
    return p ? reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (size_t)blockIdx.z * bstride) : p;
}
#endif

// level table handed to keypoint kernels by value (pointers: image 0 of the batch)
struct LevelTable {
    int n;
    size_t bstride;
    int w[AKAZE_MAX_LEVELS], h[AKAZE_MAX_LEVELS], octave[AKAZE_MAX_LEVELS], sigma_size[AKAZE_MAX_LEVELS], border[AKAZE_MAX_LEVELS];
    float esigma[AKAZE_MAX_LEVELS], ratio[AKAZE_MAX_LEVELS];
    long long pix_offset[AKAZE_MAX_LEVELS + 1];
    const float* Lt[AKAZE_MAX_LEVELS];
    const float2* Lxy[AKAZE_MAX_LEVELS];   // (Lx, Ly) interleaved
    const float* Ldet[AKAZE_MAX_LEVELS];
    uint8_t* mask[AKAZE_MAX_LEVELS];
    const uint32_t* list[AKAZE_MAX_LEVELS];   // the level's candidate list (unordered), counts in list_count[level]
    float* ref[AKAZE_MAX_LEVELS];             // three floats per list entry: the refined (x, y, response) of a surviving candidate
};

// A batched launch: `n` images of one size through one grid (gridDim.z = n). Every image owns an identical workspace slab, `stride`
// bytes apart (so a plane of image i is the plane of image 0 + i * stride); the input images are `img_stride` bytes apart.
struct Batch {
    int n = 1;
    size_t stride = 0, img_stride = 0;
};

// akaze_filters.hip
void launch_gray(const void* img, int rows, int cols, int channels, size_t stride, float* out, hipStream_t s, const Batch& b);
void launch_gauss(const float* src, float* dst, int w, int h, const GaussTaps& taps, int radius, hipStream_t s, const Batch& b);
void launch_smooth_flow(const float* src, float* smooth, float* flow, int w, int h, const GaussTaps& taps, const float* kptr, hipStream_t s, const Batch& b);
void launch_kcontrast(const float* smooth, float* modg_tmp, int w, int h, unsigned int* hmax_bits, int* hist, float* k_oct, int n_oct, hipStream_t s,
                      const Batch& b, bool gradient_done = false);
bool launch_base_strips(const void* img, int rows, int cols, int channels, size_t stride, const GaussTaps& g16, const GaussTaps& g10, float* Lt0, float* modg,
                        unsigned int* hmax_bits, bool want_modg, hipStream_t s, const Batch& b);
bool launch_nld_multi(const float* Lt, const float* Lf, float* Lnew, int w, int h, const float* step_sizes, int nsteps, hipStream_t s, const Batch& b,
                      float* half_out = nullptr);
bool launch_level_strips(const float* src, float* smooth, float* flow_out, float* Lnew, int w, int h, const GaussTaps& taps, const float* kptr,
                         const float* step_sizes, int nsteps, hipStream_t s, const Batch& b);
bool launch_level_stream(const float* src, float* smooth, float* flow_out, float* Lnew, int w, int h, const GaussTaps& taps, const float* kptr,
                         const float* step_sizes, int nsteps, hipStream_t s, const Batch& b, float* half_out = nullptr);
int level_fused_max_steps();
void launch_level_fused(const float* src, float* smooth, float* flow_out, const float* flow_in, float* Lnew, int w, int h, const GaussTaps& taps,
                        const float* kptr, const float* step_sizes, int nsteps, hipStream_t s, const Batch& b, float* half_out = nullptr);
void launch_half_sample(const float* src, int sw, float* dst, int dw, int dh, hipStream_t s, const Batch& b);
void launch_area_resize(const float* src, int sw, float* dst, int dw, int dh, const int* xofs, const float* xw, const int* xcnt, const int* yofs,
                        const float* yw, const int* ycnt, hipStream_t s, const Batch& b);
void launch_doh_fused(const float* Lsmooth, float2* Lxy, float* Ldet, int w, int h, int sc, float kside, float kmid, int border, float thr, uint8_t* mask,
                      uint32_t* list, int* list_count, hipStream_t s, const Batch& b);

// akaze_doh_strips.hip: the streaming form of the same stage for the large levels; it also writes the mask and the suppression status of
// every pixel of the level (so neither needs clearing). false = not a level for it (the caller launches doh_fused instead).
bool doh_strips_eligible(int w, int h, int sc, int batch);
bool launch_doh_strips(const float* Lsmooth, float2* Lxy, float* Ldet, int w, int h, int sc, float kside, float kmid, int border, float thr, uint8_t* mask,
                       uint8_t* status, uint32_t* list, int* list_count, hipStream_t s, const Batch& b, bool dense_det);

// Band height of a streaming kernel (a wave walks a band of rows of one 64-column strip; 256-thread blocks = four waves): the waves of a
// launch should fill the resident wave slots a WHOLE number of times - 5248 waves on 5120 slots run as two rounds, the second one 2.5 %
// full (the first measurements of both streaming kernels had exactly that). Asks the runtime how many blocks of `kernel` fit a CU, takes
// bands of about `want_rows` rows and then stretches them so that the last round is (just) full.
long long stream_wave_slots(const void* kernel, int dynamic_lds = 0);   // resident waves of a 256-thread-block kernel on the current device (cached)
template <class K>
inline int stream_band_rows(K kernel, int strips, int h, int batch, int want_rows, int min_rows, int dynamic_lds = 0) {
    const long long slots = stream_wave_slots(reinterpret_cast<const void*>(kernel), dynamic_lds);
    const long long lanes = (long long)strips * batch;                              // waves per band row
    long long bands = (h + want_rows - 1) / want_rows;
    const long long rounds = std::max<long long>(1, (lanes * bands + slots / 2) / slots);
    bands = std::max<long long>(1, rounds * slots / lanes);                          // as many bands as fill `rounds` rounds, not one more
    const int rb = (int)((h + bands - 1) / bands);
    return std::max(rb, min_rows);
}

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
