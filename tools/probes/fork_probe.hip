// How much does a fork (main stream -> side stream dependency) cost the MAIN stream's chain of dependent kernels?
//   mode 0: plain chain, no fork            mode 1: hipEventRecord(main) + hipStreamWaitEvent(side) + side kernel after every chain kernel
//   mode 4: the record alone   mode 5: one fork per two chain kernels   mode 6: side kernels without any dependency
//   mode 7: every chain kernel stores "my predecessor is done" as its first act; the side stream waits for that value
//   mode 2: flag kernel on main + hipStreamWaitValue32(side)     mode 3: hipStreamWriteValue32(main) + hipStreamWaitValue32(side)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/fork_probe.hip -o gpurun_out/fork_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void work(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i % n];
    for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f;
    p[i % n] = v;
}
__global__ void work_flag(float* p, int n, int iters, unsigned* f, unsigned v) {
    if (f && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // "my predecessor is done"
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v0 = p[i % n];
    for (int k = 0; k < iters; k++) v0 = v0 * 1.0001f + 0.5f;
    p[i % n] = v0;
}
__global__ void set_flag(unsigned* f, unsigned v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
int main(int argc, char** argv) {
    const int N = 16, reps = 30;
    float *a, *b;
    CK(hipMalloc(&a, 1 << 24));
    CK(hipMalloc(&b, 1 << 24));
    unsigned* flag = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(fe));
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice));
    hipEvent_t join;
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming | hipEventReleaseToDevice));
    for (int mode = 0; mode < 8; mode++) {
        if ((mode == 2 || mode == 3 || mode == 7) && fe != hipSuccess) continue;
        double best = 1e9;
        unsigned epoch = 0;
        for (int r = 0; r < reps; r++) {
            CK(hipDeviceSynchronize());
            if (flag) { *flag = 0; }
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (mode == 7) {   // the NEXT chain kernel reports its predecessor done; the side stream waits for that value
                    hipLaunchKernelGGL(work_flag, dim3(1024), dim3(256), 0, s, a, 1 << 22, 400, flag, (unsigned)i);
                    if (i > 0) {
                        CK(hipStreamWaitValue32(side, flag, i, hipStreamWaitValueGte, 0xFFFFFFFFu));
                        hipLaunchKernelGGL(work, dim3(512), dim3(256), 0, side, b, 1 << 22, 300);
                    }
                    continue;
                }
                hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, s, a, 1 << 22, 400);
                if (mode == 1) {
                    CK(hipEventRecord(ev[i], s));
                    CK(hipStreamWaitEvent(side, ev[i], 0));
                } else if (mode == 2) {
                    hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s, flag, (unsigned)(i + 1));
                    CK(hipStreamWaitValue32(side, flag, i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
                } else if (mode == 3) {
                    CK(hipStreamWriteValue32(s, flag, i + 1, 0));
                    CK(hipStreamWaitValue32(side, flag, i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
                }
                else if (mode == 4) {   // the record alone: nobody waits, no side kernel
                    CK(hipEventRecord(ev[i], s));
                } else if (mode == 5 && (i & 1)) {   // one record for two chain kernels
                    CK(hipEventRecord(ev[i], s));
                    CK(hipStreamWaitEvent(side, ev[i], 0));
                }
                if (mode == 6) {   // side kernels with no dependency at all (what sharing the machine costs the chain)
                }
                if (mode && mode != 4) hipLaunchKernelGGL(work, dim3(512), dim3(256), 0, side, b, 1 << 22, 300);
            }
            if (mode == 7) {   // the last chain kernel's report comes from the kernel after it
                hipLaunchKernelGGL(work_flag, dim3(64), dim3(256), 0, s, a, 1 << 22, 10, flag, (unsigned)N);
                CK(hipStreamWaitValue32(side, flag, N, hipStreamWaitValueGte, 0xFFFFFFFFu));
                hipLaunchKernelGGL(work, dim3(512), dim3(256), 0, side, b, 1 << 22, 300);
            }
            if (mode) {
                CK(hipEventRecord(join, side));
                CK(hipStreamWaitEvent(s, join, 0));
            }
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, a, 1 << 22, 10);
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
            (void)epoch;
        }
        printf("mode %d: %.1f us for %d chain kernels (best of %d)\n", mode, best, N, reps);
    }
    return 0;
}
