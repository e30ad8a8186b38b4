// csrc/misc.hip — small byte/gather kernels on the edges of the path.
#include <chrono>
#include "config.h"
#include "kernels.h"

namespace apds {

// feature_extraction/src/lib.rs:161-180. bug_compatible reproduces :169 (img_idx) and :176-177 (img1 twice).
__global__ void points_from_matches_kernel(const apds_keypoint* __restrict__ kp1, int n1, const apds_keypoint* __restrict__ kp2, int n2,
                                           const apds_dmatch* __restrict__ m, int nm, int bug, float2* __restrict__ p1,
                                           float2* __restrict__ p2, int* __restrict__ err) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nm) return;
    const apds_dmatch mm = m[i];
    const int i1 = bug ? mm.img_idx : mm.query_idx;
    const int i2 = mm.train_idx;
    if (i1 < 0 || i1 >= n1 || i2 < 0 || i2 >= n2) {
        atomicExch(err, 1);
        return;
    }
    const float2 a = make_float2(kp1[i1].x, kp1[i1].y);
    p1[i] = a;
    p2[i] = bug ? a : make_float2(kp2[i2].x, kp2[i2].y);
}

// homographier/src/homographier/mod.rs:183-220: RGBA8 -> BGRA (Vec4b), one dword per pixel
__global__ void rgba_to_bgra_kernel(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t v = in[i];   // bytes R,G,B,A = bits 0-7, 8-15, 16-23, 24-31
        out[i] = (v & 0xFF00FF00u) | ((v & 0xFFu) << 16) | ((v >> 16) & 0xFFu);
    }
}

// set bytes of a mask (the inlier count of a homography's mask: the pipeline's per-frame figure)
__global__ void count_nonzero_kernel(const uint8_t* __restrict__ b, int n, int* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += b[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

int count_nonzero_device(const uint8_t* bytes, int n, int* count_dev, hipStream_t s) {
    if (n <= 0) return 0;
    HIP_CHECK(hipMemsetAsync(count_dev, 0, sizeof(int), s));
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(std::min(64, ceil_div(n, 256))), dim3(256), 0, s, bytes, n, count_dev);
    HIP_CHECK(hipGetLastError());
    int* host = ctx().pinned_ints(1);
    HIP_CHECK(hipMemcpyAsync(host, count_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return host[0];
}

void points_from_matches_device(const apds_keypoint* kp1, int n1, const apds_keypoint* kp2, int n2, const apds_dmatch* m, int nm,
                                int bug_compatible, float* pts1, float* pts2, int* err_flag, hipStream_t s) {
    if (nm <= 0) return;
    hipLaunchKernelGGL(points_from_matches_kernel, dim3(ceil_div(nm, 256)), dim3(256), 0, s, kp1, n1, kp2, n2, m, nm, bug_compatible,
                       reinterpret_cast<float2*>(pts1), reinterpret_cast<float2*>(pts2), err_flag);
    HIP_CHECK(hipGetLastError());
}

void rgba_to_bgra_device(const uint8_t* rgba, size_t n_pixels, uint8_t* bgra, hipStream_t s) {
    if (!n_pixels) return;
    const int blocks = (int)std::min<size_t>((n_pixels + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(rgba_to_bgra_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(rgba), n_pixels,
                       reinterpret_cast<uint32_t*>(bgra));
    HIP_CHECK(hipGetLastError());
}


// ---- side_stream_beside (common.h) ---------------------------------------------------------------------------------------------
__global__ void spin_kernel(long long cycles, int* sink) {
    const long long t0 = __builtin_readcyclecounter();
    long long t = t0;
    for (int it = 0; it < 20000 && t - t0 < cycles; it++) {   // bounded whatever the counter does: the wave always ends
        __builtin_amdgcn_s_sleep(8);
        t = __builtin_readcyclecounter();
    }
    if (sink && cycles < 0) *sink = (int)t;   // never: keeps the loop
}

hipStream_t side_stream_beside(hipStream_t caller) {
    ThreadCtx& c = ctx();
    if (c.side_probe_choice && c.side_probe_caller == caller) return c.side_probe_choice;
    if (!config().side_probe) return c.side_stream();
    for (hipStream_t& st : c.side_pool)
        if (!st && !take_cached_side_stream(c.device, st)) HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, 0));
    HIP_CHECK(hipStreamSynchronize(caller));
    const long long spin = 60000;   // ~25 us of the 2.4 GHz shader clock (s_memtime counts at 100 MHz on some parts: then longer, still bounded)
    double best = 1e30;
    int best_i = 0;
    for (int rep = 0; rep < 2; rep++) {       // the first round also warms the launch path
        for (int i = 0; i < 4; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, caller, spin, (int*)nullptr);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, c.side_pool[i], spin, (int*)nullptr);
            HIP_CHECK(hipStreamSynchronize(caller));
            HIP_CHECK(hipStreamSynchronize(c.side_pool[i]));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 1 && dt < best) {
                best = dt;
                best_i = i;
            }
        }
    }
    c.side_probe_caller = caller;
    c.side_probe_choice = c.side_pool[best_i];
    return c.side_probe_choice;
}

__global__ void fork_signal_kernel(ForkSignal sig) { APDS_FORK_SIGNAL(sig); }
void launch_fork_signal(ForkSignal sig, hipStream_t s) {
    if (sig.flag) hipLaunchKernelGGL(fork_signal_kernel, dim3(1), dim3(64), 0, s, sig);
}

}  // namespace apds
