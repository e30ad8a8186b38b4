// Hamming distances of 16 x 16 pairs of 512-bit rows on v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 (e2m1) operands: train bit -> 1.0 (0x2),
// query bit -> -2.0 (0xC), accumulator preset to popcount(train): acc = pc(t) - 2 t.q, distance = pc(q) + acc. Checked against popcount(xor).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/fp4_probe.hip -o /tmp/fp4_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t spread8(uint32_t b, uint32_t nib) {   // 8 bits -> 8 nibbles of value nib (a single-bit-pattern multiple)
    uint32_t t = b & 0xFF;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t * nib;
}
template <int SCALE>
__global__ void k(const uint32_t* train, const uint32_t* query, float* out) {   // 16 rows x 16 dwords each
    const int lane = threadIdx.x, col = lane & 15, kq = lane >> 4;
    f32x4 acc;
    for (int j = 0; j < 4; j++) {
        int pc = 0;
        for (int d = 0; d < 16; d++) pc += __popc(train[(4 * kq + j) * 16 + d]);
        acc[j] = (float)pc;
    }
    for (int s = 0; s < 4; s++) {
        const uint32_t ta = train[col * 16 + 4 * s + kq], qb = query[col * 16 + 4 * s + kq];
        v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            A[i] = (int)spread8(ta >> (8 * i), 0x2);
            B[i] = (int)spread8(qb >> (8 * i), 0xC);
        }
        if (SCALE) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 4, 4, 0, 0, 0, 0);
    }
    for (int j = 0; j < 4; j++) out[(4 * kq + j) * 16 + col] = acc[j];   // row 4 kq + j (train), column col (query)
}
int main() {
    uint32_t ht[256], hq[256];
    srand(7);
    for (int i = 0; i < 256; i++) { ht[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); hq[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); }
    for (int d = 0; d < 16; d++) { ht[3 * 16 + d] = 0; hq[5 * 16 + d] = 0xFFFFFFFFu; ht[9 * 16 + d] = 0xFFFFFFFFu; }   // extreme rows
    uint32_t *dt, *dq; float* dout;
    hipMalloc(&dt, 1024); hipMalloc(&dq, 1024); hipMalloc(&dout, 1024);
    hipMemcpy(dt, ht, 1024, hipMemcpyHostToDevice); hipMemcpy(dq, hq, 1024, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; variant++) {
        if (variant) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dt, dq, dout);
        else hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dt, dq, dout);
        float ho[256];
        hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) {
                int want = 0, pcq = 0;
                for (int d = 0; d < 16; d++) { want += __builtin_popcount(ht[r * 16 + d] ^ hq[c * 16 + d]); pcq += __builtin_popcount(hq[c * 16 + d]); }
                const float got = ho[r * 16 + c] + (float)pcq;
                if (got != (float)want) { if (bad < 5) printf("  row %d query %d: got %g want %d\n", r, c, got, want); bad++; }
            }
        printf("variant %s: %d of 256 wrong\n", variant ? "scale 0x7F (2^0)" : "scale operands 0", bad);
    }
    return 0;
}
