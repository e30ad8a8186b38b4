// csrc/ingest.hip — the steps on either side of the hot path (SURVEY §8f "next" rows 2 and 4):
//   * tile ingest: per-band min-max normalisation + gamma 1/2.2 + NaN handling -> RGBA8 / BGRA8
//     (/root/reference/geotiff_extractor/src/image_extractor/mod.rs:346-378, 402-422), one streaming kernel,
//     optionally fused with raster_to_mat's R<->B swap (homographier mod.rs:183-220);
//   * warp_image_perspective (homographier mod.rs:271-300): cv::warpPerspective, INTER_LINEAR fixed point,
//     BORDER_CONSTANT (1,1,1,1), 4-channel u8 — a gather kernel, one thread per destination pixel.
#include <climits>
#include <cmath>

#include "kernels.h"
#include "pnp_core.h"

namespace apds {

struct GeoTransform {
    double c[6];
};

// f32_to_u8 (mod.rs:410-422): None -> 0 in band_merger. powf(x, 1/2.2f) is evaluated in f64 and rounded to f32: that is
// the correctly rounded f32 result (what glibc's powf returns) except for values within ~1e-8 relative of a rounding tie.
__device__ __forceinline__ uint32_t band_to_u8(float v, float mn, float mx) {
    if (isnan(v)) return 0;
    const float f = (v - mn) / (mx - mn);
    if (!(f >= 0.0f && f <= 1.0f)) return 0;
    const float g = (float)pow((double)f, (double)(1.0f / 2.2f));
    const float r = roundf(g * 255.0f);
    return r <= 0.0f ? 0u : (r >= 255.0f ? 255u : (uint32_t)r);
}

__global__ void band_merger_kernel(const float* __restrict__ red, const float* __restrict__ green, const float* __restrict__ blue, size_t n,
                                   float rmin, float rmax, float gmin, float gmax, float bmin, float bmax, int bgra, uint32_t* __restrict__ out) {
    APDS_RAISE_WAVE_PRIORITY();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float r = red[i], g = green[i], b = blue[i];
        const uint32_t alpha = (isnan(r) && isnan(g) && isnan(b)) ? 0u : 255u;
        const uint32_t R = band_to_u8(r, rmin, rmax), G = band_to_u8(g, gmin, gmax), B = band_to_u8(b, bmin, bmax);
        out[i] = bgra ? (B | (G << 8) | (R << 16) | (alpha << 24)) : (R | (G << 8) | (B << 16) | (alpha << 24));
    }
}

__device__ __forceinline__ int sat_int_rn(double v) {
    return v <= (double)INT_MIN ? INT_MIN : (v >= (double)INT_MAX ? INT_MAX : __double2int_rn(v));
}

__global__ void warp_perspective_kernel(const uint32_t* __restrict__ src, int rows, int cols, double m0, double m1, double m2, double m3, double m4,
                                        double m5, double m6, double m7, double m8, const short* __restrict__ tab, int dst_rows, int dst_cols,
                                        uint32_t* __restrict__ dst) {
    APDS_RAISE_WAVE_PRIORITY();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dst_cols || y >= dst_rows) return;
    const double X0 = m0 * x + m1 * y + m2, Y0 = m3 * x + m4 * y + m5;
    double W = m6 * x + m7 * y + m8;
    W = W ? 32. / W : 0;
    const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, X0 * W));
    const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, Y0 * W));
    const int X = sat_int_rn(fX), Y = sat_int_rn(fY);
    const int sx = X >> 5, sy = Y >> 5;
    const short* w = &tab[((Y & 31) * 32 + (X & 31)) * 4];
    uint32_t p[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xx = sx + (k & 1), yy = sy + (k >> 1);
        p[k] = (xx >= 0 && xx < cols && yy >= 0 && yy < rows) ? src[(size_t)yy * cols + xx] : 0x01010101u;   // border value (1,1,1,1)
    }
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int v = (int)((p[0] >> (8 * c)) & 0xFF) * w[0] + (int)((p[1] >> (8 * c)) & 0xFF) * w[1] + (int)((p[2] >> (8 * c)) & 0xFF) * w[2] +
                      (int)((p[3] >> (8 * c)) & 0xFF) * w[3];
        out |= (uint32_t)(((v + (1 << 14)) >> 15) & 0xFF) << (8 * c);
    }
    dst[(size_t)y * dst_cols + x] = out;
}

// The same map for the other element types warp_image_perspective<T: DataType> admits (mod.rs:271-300 is generic): CH interleaved channels
// of u8 (OpenCV's fixed-point path: the 15-bit weight table above, (sum + 2^14) >> 15) or f32 (remapBilinear's float path: the four
// weights (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx as float products of the 1/32-step fractions, ((v0 w0 + v1 w1) + v2 w2) + v3 w3 in
// binary32, one operation each; a destination pixel whose 2x2 footprint lies wholly outside the source IS the border value).
template <class T, int CH>
__global__ void warp_perspective_generic_kernel(const T* __restrict__ src, int rows, int cols, double m0, double m1, double m2, double m3, double m4, double m5,
                                                double m6, double m7, double m8, const short* __restrict__ tab, int dst_rows, int dst_cols, T* __restrict__ dst) {
    APDS_RAISE_WAVE_PRIORITY();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dst_cols || y >= dst_rows) return;
    const double X0 = m0 * x + m1 * y + m2, Y0 = m3 * x + m4 * y + m5;
    double W = m6 * x + m7 * y + m8;
    W = W ? 32. / W : 0;
    const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, X0 * W));
    const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, Y0 * W));
    const int X = sat_int_rn(fX), Y = sat_int_rn(fY);
    const int sx = X >> 5, sy = Y >> 5;
    T* d = dst + ((size_t)y * dst_cols + x) * CH;
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xx = sx + (k & 1), yy = sy + (k >> 1);
        in[k] = xx >= 0 && xx < cols && yy >= 0 && yy < rows;
    }
    if constexpr (sizeof(T) == 1) {
        const short* w = &tab[((Y & 31) * 32 + (X & 31)) * 4];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v += (in[k] ? (int)src[((size_t)(sy + (k >> 1)) * cols + sx + (k & 1)) * CH + c] : 1) * w[k];   // border value 1
            d[c] = (T)((v + (1 << 14)) >> 15);
        }
    } else {
        if (!(in[0] || in[1] || in[2] || in[3])) {
#pragma unroll
            for (int c = 0; c < CH; c++) d[c] = (T)1;
            return;
        }
        const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
        const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = in[k] ? (float)src[((size_t)(sy + (k >> 1)) * cols + sx + (k & 1)) * CH + c] : 1.0f;
            float acc = v[0] * w0;
            acc += v[1] * w1;
            acc += v[2] * w2;
            acc += v[3] * w3;
            d[c] = (T)acc;
        }
    }
}

namespace {

// 32x32 table of 2x2 fixed-point (15 bit) bilinear weights that sum to 2^15 (OpenCV's BilinearTab_i restated)
const short* bilinear_tab_host() {
    static short tab[32 * 32 * 4];
    static bool init = false;
    if (!init) {
        float lin[32][2];
        for (int i = 0; i < 32; i++) {
            const float x = i * (1.f / 32);
            lin[i][0] = 1.f - x;
            lin[i][1] = x;
        }
        for (int fy = 0; fy < 32; fy++)
            for (int fx = 0; fx < 32; fx++) {
                short* w = &tab[(fy * 32 + fx) * 4];
                int isum = 0;
                for (int k1 = 0; k1 < 2; k1++)
                    for (int k2 = 0; k2 < 2; k2++) {
                        const int iv = (int)lrintf(lin[fy][k1] * lin[fx][k2] * 32768.f);
                        w[k1 * 2 + k2] = (short)(iv > 32767 ? 32767 : iv);
                        isum += w[k1 * 2 + k2];
                    }
                if (isum != 32768) {
                    const int diff = isum - 32768;
                    int Mk = 0;
                    for (int k = 1; k < 4; k++)
                        if (w[k] > w[Mk]) Mk = k;
                    if (w[Mk] - diff > 32767) Mk = 3;
                    w[Mk] = (short)(w[Mk] - diff);
                }
            }
        init = true;
    }
    return tab;
}

bool invert3x3(const double* m, double* inv) {
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0) return false;
    const double id = 1. / d;
    inv[0] = (m[4] * m[8] - m[5] * m[7]) * id;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = (m[5] * m[6] - m[3] * m[8]) * id;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = (m[3] * m[7] - m[4] * m[6]) * id;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

}  // namespace

void band_merger_device(const float* r, const float* g, const float* b, size_t n, const double* mm, int bgra, uint8_t* out, hipStream_t s) {
    if (!n) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(band_merger_kernel, dim3(blocks), dim3(256), 0, s, r, g, b, n, (float)mm[0], (float)mm[1], (float)mm[2], (float)mm[3], (float)mm[4],
                       (float)mm[5], bgra, reinterpret_cast<uint32_t*>(out));
    HIP_CHECK(hipGetLastError());
}

void warp_perspective_device(const uint8_t* src, int rows, int cols, const double* M, int dst_rows, int dst_cols, uint8_t* dst, hipStream_t s) {
    double inv[9];
    APDS_REQUIRE(invert3x3(M, inv), APDS_ERR_ASSERT, "perspective matrix is singular");
    ThreadCtx& c = ctx();
    short* tab = c.alloc_n<short>(32 * 32 * 4);
    HIP_CHECK(hipMemcpyAsync(tab, bilinear_tab_host(), sizeof(short) * 32 * 32 * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(warp_perspective_kernel, dim3(ceil_div(dst_cols, 256), dst_rows), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(src), rows, cols,
                       inv[0], inv[1], inv[2], inv[3], inv[4], inv[5], inv[6], inv[7], inv[8], (const short*)tab, dst_rows, dst_cols,
                       reinterpret_cast<uint32_t*>(dst));
    HIP_CHECK(hipGetLastError());
}

// element = u8 (elem_bytes 1) or f32 (4), 1 / 3 / 4 interleaved channels; 8UC4 keeps its dword kernel
void warp_perspective_any_device(const void* src, int rows, int cols, int channels, int elem_bytes, const double* M, int dst_rows, int dst_cols, void* dst,
                                 hipStream_t s) {
    if (elem_bytes == 1 && channels == 4) return warp_perspective_device(static_cast<const uint8_t*>(src), rows, cols, M, dst_rows, dst_cols, static_cast<uint8_t*>(dst), s);
    double inv[9];
    APDS_REQUIRE(invert3x3(M, inv), APDS_ERR_ASSERT, "perspective matrix is singular");
    ThreadCtx& c = ctx();
    short* tab = c.alloc_n<short>(32 * 32 * 4);
    HIP_CHECK(hipMemcpyAsync(tab, bilinear_tab_host(), sizeof(short) * 32 * 32 * 4, hipMemcpyHostToDevice, s));
    const dim3 grid(ceil_div(dst_cols, 256), dst_rows), block(256);
    auto go = [&](auto kernel, auto* typed_src, auto* typed_dst) {
        hipLaunchKernelGGL(kernel, grid, block, 0, s, typed_src, rows, cols, inv[0], inv[1], inv[2], inv[3], inv[4], inv[5], inv[6], inv[7], inv[8], (const short*)tab,
                           dst_rows, dst_cols, typed_dst);
    };
    const uint8_t* s8 = static_cast<const uint8_t*>(src);
    uint8_t* d8 = static_cast<uint8_t*>(dst);
    const float* sf = static_cast<const float*>(src);
    float* df = static_cast<float*>(dst);
    if (elem_bytes == 1 && channels == 1) go(warp_perspective_generic_kernel<uint8_t, 1>, s8, d8);
    else if (elem_bytes == 1 && channels == 3) go(warp_perspective_generic_kernel<uint8_t, 3>, s8, d8);
    else if (elem_bytes == 4 && channels == 1) go(warp_perspective_generic_kernel<float, 1>, sf, df);
    else if (elem_bytes == 4 && channels == 3) go(warp_perspective_generic_kernel<float, 3>, sf, df);
    else if (elem_bytes == 4 && channels == 4) go(warp_perspective_generic_kernel<float, 4>, sf, df);
    else fail(APDS_ERR_ASSERT, "warp_perspective: element type must be u8 or f32 with 1, 3 or 4 channels");
    HIP_CHECK(hipGetLastError());
}

// ---- feature_database/src/elevationdb.rs:64-104 get_world_coordinates, batched -----------------------------------------
// mosaic pixel -> dataset geotransform -> (lon, lat); elevation through the inverse elevation geotransform and the reference's
// row id (elevationdb.rs:240: round(y) * x_size + round(x) + 1); EPSG:4326 -> EPSG:4978 (PROJ's geodetic -> geocentric closed
// form, WGS 84). One thread per point, f64; sin/cos are the fixed polynomials of pnp_core.h so the CPU oracle agrees bit for bit.
__device__ __forceinline__ void signed_sincos(double a, double& s, double& c) {
    const bool neg = a < 0;
    pnp::sincos_fixed(neg ? -a : a, s, c);
    if (neg) s = -s;
}

__global__ void world_coordinates_kernel(const double* __restrict__ xy, int n, GeoTransform dgt, GeoTransform inv_egt, int has_elev,
                                         const double* __restrict__ elev, int ew, int eh, double* __restrict__ xyz, int* __restrict__ missing) {
    APDS_RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xy[2 * i], y = xy[2 * i + 1];
    const double gx = dgt.c[0] + x * dgt.c[1] + y * dgt.c[2];
    const double gy = dgt.c[3] + x * dgt.c[4] + y * dgt.c[5];
    double h = 0.0;
    if (has_elev) {
        const double px = inv_egt.c[0] + gx * inv_egt.c[1] + gy * inv_egt.c[2];
        const double py = inv_egt.c[3] + gx * inv_egt.c[4] + gy * inv_egt.c[5];
        const long long id0 = (long long)(int)round(py) * ew + (long long)(int)round(px);
        if (id0 < 0 || id0 >= (long long)ew * eh) {
            *missing = 1;
            const double nanv = __longlong_as_double(0x7FF8000000000000ll);
            xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = nanv;
            return;
        }
        h = elev[id0];
    }
    const double a = 6378137.0, f = 1.0 / 298.257223563, es = f * (2.0 - f), deg = 0.017453292519943296;
    double sp, cp, sl, cl;
    signed_sincos(gy * deg, sp, cp);
    signed_sincos(gx * deg, sl, cl);
    const double N = a / sqrt(1.0 - es * sp * sp);
    xyz[3 * i] = (N + h) * cp * cl;
    xyz[3 * i + 1] = (N + h) * cp * sl;
    xyz[3 * i + 2] = (N * (1.0 - es) + h) * sp;
}

bool invert_geotransform(const double* gt, double* out) {   // GDALInvGeoTransform
    if (gt[2] == 0.0 && gt[4] == 0.0 && gt[1] != 0.0 && gt[5] != 0.0) {
        out[0] = -gt[0] / gt[1];
        out[1] = 1.0 / gt[1];
        out[2] = 0.0;
        out[3] = -gt[3] / gt[5];
        out[4] = 0.0;
        out[5] = 1.0 / gt[5];
        return true;
    }
    const double det = gt[1] * gt[5] - gt[2] * gt[4];
    const double mag = fmax(fmax(fabs(gt[1]), fabs(gt[2])), fmax(fabs(gt[4]), fabs(gt[5])));
    if (fabs(det) <= 1e-10 * mag * mag) return false;
    const double inv_det = 1.0 / det;
    out[1] = gt[5] * inv_det;
    out[4] = -gt[4] * inv_det;
    out[2] = -gt[2] * inv_det;
    out[5] = gt[1] * inv_det;
    out[0] = (gt[2] * gt[3] - gt[0] * gt[5]) * inv_det;
    out[3] = (-gt[1] * gt[3] + gt[0] * gt[4]) * inv_det;
    return true;
}

// all pointers device; returns 1 if some elevation lookup missed (those points are NaN)
int world_coordinates_device(const double* xy, int n, const double* dgt_host, const double* egt_host, const double* elev, int ew, int eh, double* xyz,
                             hipStream_t s) {
    GeoTransform d{}, inv{};
    for (int i = 0; i < 6; i++) d.c[i] = dgt_host[i];
    if (egt_host) APDS_REQUIRE(invert_geotransform(egt_host, inv.c), APDS_ERR_BAD_ARG, "elevation geotransform is not invertible");
    int* missing = ctx().alloc_n<int>(1);
    HIP_CHECK(hipMemsetAsync(missing, 0, sizeof(int), s));
    if (n > 0)
        hipLaunchKernelGGL(world_coordinates_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, xy, n, d, inv, egt_host ? 1 : 0, elev, ew, eh, xyz, missing);
    HIP_CHECK(hipGetLastError());
    int m = 0;
    HIP_CHECK(hipMemcpyAsync(&m, missing, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return m;
}

}  // namespace apds

using namespace apds;

extern "C" {

int apds_band_merger(const float* red, const float* green, const float* blue, size_t n, const double* minmax6, int bgra, uint8_t* out) {
    return guarded([&] {
        APDS_REQUIRE(n == 0 || (red && green && blue && out && minmax6), APDS_ERR_BAD_ARG, "null argument");
        if (!n) return;
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        float* d = c.alloc_n<float>(3 * n);
        uint8_t* o = c.alloc_n<uint8_t>(4 * n);
        HIP_CHECK(hipMemcpyAsync(d, red, n * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d + n, green, n * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d + 2 * n, blue, n * 4, hipMemcpyHostToDevice, s));
        band_merger_device(d, d + n, d + 2 * n, n, minmax6, bgra, o, s);
        HIP_CHECK(hipMemcpyAsync(out, o, 4 * n, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int apds_dev_band_merger(const void* red, const void* green, const void* blue, size_t n, const double* minmax6, int bgra, void* out, void* stream) {
    return guarded([&] {
        APDS_REQUIRE(minmax6, APDS_ERR_BAD_ARG, "null argument");
        band_merger_device(static_cast<const float*>(red), static_cast<const float*>(green), static_cast<const float*>(blue), n, minmax6, bgra,
                           static_cast<uint8_t*>(out), pick_stream(stream));
    });
}

static int warp_host(const void* src, int rows, int cols, int channels, int elem_bytes, const double* M, int dst_rows, int dst_cols, void* dst) {
    return guarded([&] {
        APDS_REQUIRE(src && M && dst, APDS_ERR_BAD_ARG, "null argument");
        APDS_REQUIRE(rows > 0 && cols > 0 && dst_rows > 0 && dst_cols > 0, APDS_ERR_ASSERT, "empty image");
        APDS_REQUIRE(channels == 1 || channels == 3 || channels == 4, APDS_ERR_ASSERT, "warp_perspective serves 1-, 3- and 4-channel images (u8 or f32 elements)");
        ThreadCtx& c = ctx();
        c.ws_reset();
        hipStream_t s = c.stream;
        const size_t px = (size_t)channels * elem_bytes;
        uint8_t* ds = c.alloc_n<uint8_t>((size_t)rows * cols * px);
        uint8_t* dd = c.alloc_n<uint8_t>((size_t)dst_rows * dst_cols * px);
        HIP_CHECK(hipMemcpyAsync(ds, src, (size_t)rows * cols * px, hipMemcpyHostToDevice, s));
        warp_perspective_any_device(ds, rows, cols, channels, elem_bytes, M, dst_rows, dst_cols, dd, s);
        HIP_CHECK(hipMemcpyAsync(dst, dd, (size_t)dst_rows * dst_cols * px, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int apds_warp_perspective(const uint8_t* src, int rows, int cols, int channels, const double* M, int dst_rows, int dst_cols, uint8_t* dst) {
    return warp_host(src, rows, cols, channels, 1, M, dst_rows, dst_cols, dst);
}

int apds_warp_perspective_f32(const float* src, int rows, int cols, int channels, const double* M, int dst_rows, int dst_cols, float* dst) {
    return warp_host(src, rows, cols, channels, 4, M, dst_rows, dst_cols, dst);
}

}  // extern "C"
