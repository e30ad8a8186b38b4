// oracle/ingest_oracle.cpp — tile ingest pixel math and perspective warp restated on the CPU. TEST INFRASTRUCTURE ONLY.
//
// band_merger / f32_to_u8 / gamma_correction: /root/reference/geotiff_extractor/src/image_extractor/mod.rs:346-378,
// 402-422 (first-party Rust; pinned by its tests gamma_correct_input :517-525, convert_f32_to_u8_success :547-555,
// merging_bands :626-646). warp_image_perspective: homographier/src/homographier/mod.rs:271-300 -> OpenCV imgproc
// warpPerspective / remap (fixed-point bilinear), restated; pinned by warp_image_empty (mod.rs:683-707).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

extern "C" {

float oracle_gamma_correction(float v, int* ok) {
    if (!(v >= 0.0f && v <= 1.0f)) {   // !(0.0..=1.0).contains(): NaN fails too
        *ok = 0;
        return 0.0f;
    }
    *ok = 1;
    return powf(v, 1.0f / 2.2f);
}

int oracle_f32_to_u8(float v, float mn, float mx, int* ok) {
    if (std::isnan(v)) {
        *ok = 0;
        return 0;
    }
    const float f = (v - mn) / (mx - mn);
    const float g = oracle_gamma_correction(f, ok);
    if (!*ok) return 0;
    const float r = roundf(g * 255.0f);   // f32::round: half away from zero; `as u8` saturates
    return r <= 0.0f ? 0 : (r >= 255.0f ? 255 : (int)r);
}

void oracle_band_merger(const float* red, const float* green, const float* blue, size_t n, const double* mm, uint8_t* rgba) {
    for (size_t i = 0; i < n; i++) {
        int ok;
        const bool all_nan = std::isnan(red[i]) && std::isnan(green[i]) && std::isnan(blue[i]);
        int r = oracle_f32_to_u8(red[i], (float)mm[0], (float)mm[1], &ok);
        if (!ok) r = 0;
        int g = oracle_f32_to_u8(green[i], (float)mm[2], (float)mm[3], &ok);
        if (!ok) g = 0;
        int b = oracle_f32_to_u8(blue[i], (float)mm[4], (float)mm[5], &ok);
        if (!ok) b = 0;
        rgba[4 * i + 0] = (uint8_t)r;
        rgba[4 * i + 1] = (uint8_t)g;
        rgba[4 * i + 2] = (uint8_t)b;
        rgba[4 * i + 3] = all_nan ? 0 : 255;
    }
}

// ---- warpPerspective, INTER_LINEAR, 8UC4, BORDER_CONSTANT ------------------------------------------------------
static inline int sat_int(double v) { return v <= (double)INT_MIN ? INT_MIN : (v >= (double)INT_MAX ? INT_MAX : (int)lrint(v)); }

static bool invert3x3(const double* m, double* inv) {
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

// 32x32 table of 2x2 fixed-point (15 bit) bilinear weights that sum to 32768
static const short* bilinear_tab() {
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
                        const float v = lin[fy][k1] * lin[fx][k2];
                        const int iv = (int)lrintf(v * 32768.f);
                        w[k1 * 2 + k2] = (short)(iv > 32767 ? 32767 : iv);
                        isum += w[k1 * 2 + k2];
                    }
                if (isum != 32768) {
                    // OpenCV nudges one tap so the four weights sum to 2^15. Its scan for that tap is written for the
                    // 4- and 8-tap kernels; for the 2x2 case the effective rule restated here is: correct the largest
                    // weight, except when that would leave the 16-bit range (the exact tap (32767,0,0,0)), where the
                    // diagonal tap takes the unit. Differences between variants are at most 1 LSB of one output byte.
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

int oracle_warp_perspective_8uc4(const uint8_t* src, int rows, int cols, const double* Min, int dst_rows, int dst_cols, uint8_t* dst) {
    double M[9];
    if (!invert3x3(Min, M)) return -1;
    const short* tab = bilinear_tab();
    const uint8_t border[4] = {1, 1, 1, 1};
    for (int y = 0; y < dst_rows; y++)
        for (int x = 0; x < dst_cols; x++) {
            const double X0 = M[0] * x + M[1] * y + M[2], Y0 = M[3] * x + M[4] * y + M[5];
            double W = M[6] * x + M[7] * y + M[8];
            W = W ? 32. / W : 0;
            const double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, X0 * W));
            const double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, Y0 * W));
            const int X = sat_int(fX), Y = sat_int(fY);
            const int sx = X >> 5, sy = Y >> 5;
            const short* w = &tab[((Y & 31) * 32 + (X & 31)) * 4];
            uint8_t* d = &dst[((size_t)y * dst_cols + x) * 4];
            const uint8_t* p[4];
            for (int k = 0; k < 4; k++) {
                const int xx = sx + (k & 1), yy = sy + (k >> 1);
                p[k] = (xx >= 0 && xx < cols && yy >= 0 && yy < rows) ? &src[((size_t)yy * cols + xx) * 4] : border;
            }
            for (int c = 0; c < 4; c++) {
                const int v = p[0][c] * w[0] + p[1][c] * w[1] + p[2][c] * w[2] + p[3][c] * w[3];
                d[c] = (uint8_t)((v + (1 << 14)) >> 15);
            }
        }
    return 0;
}

// ---- feature_database/src/elevationdb.rs:64-104 get_world_coordinates (first-party Rust over GDAL/PROJ) ---------------------
// pixel -> geotransform -> (optional) elevation lookup through the inverse elevation geotransform -> EPSG:4326 -> EPSG:4978.
// GDAL GeoTransform::apply / invert (GDALInvGeoTransform) and PROJ's geodetic -> geocentric conversion (pj_cart) are absent
// here and restated; sin/cos are the fixed polynomials shared with the GPU. PARITY UNPINNED (the reference's tests need Postgres).
static void fixed_sincos(double a, double* s, double* c) {
    const bool neg = a < 0;
    if (neg) a = -a;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const int k = (int)(a * two_over_pi + 0.5);
    const double r = (a - k * pio2_hi) - k * pio2_lo;
    const double r2 = r * r;
    double ps = -7.6471637318198164759e-13;
    ps = ps * r2 + 1.6059043836821614599e-10;
    ps = ps * r2 + -2.5052108385441718775e-08;
    ps = ps * r2 + 2.7557319223985890653e-06;
    ps = ps * r2 + -1.9841269841269841270e-04;
    ps = ps * r2 + 8.3333333333333333333e-03;
    ps = ps * r2 + -1.6666666666666666667e-01;
    const double sr = r + r * (r2 * ps);
    double pc = 4.7794773323873852974e-14;
    pc = pc * r2 + -1.1470745597729724714e-11;
    pc = pc * r2 + 2.0876756987868098979e-09;
    pc = pc * r2 + -2.7557319223985890653e-07;
    pc = pc * r2 + 2.4801587301587301587e-05;
    pc = pc * r2 + -1.3888888888888888889e-03;
    pc = pc * r2 + 4.1666666666666666667e-02;
    pc = pc * r2 + -0.5;
    const double cr = 1.0 + r2 * pc;
    double sv, cv;
    switch (k & 3) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    *s = neg ? -sv : sv;
    *c = cv;
}

int oracle_invert_geotransform(const double* gt, double* out) {   // GDALInvGeoTransform
    if (gt[2] == 0.0 && gt[4] == 0.0 && gt[1] != 0.0 && gt[5] != 0.0) {
        out[0] = -gt[0] / gt[1];
        out[1] = 1.0 / gt[1];
        out[2] = 0.0;
        out[3] = -gt[3] / gt[5];
        out[4] = 0.0;
        out[5] = 1.0 / gt[5];
        return 1;
    }
    const double det = gt[1] * gt[5] - gt[2] * gt[4];
    const double magnitude = std::max(std::max(std::fabs(gt[1]), std::fabs(gt[2])), std::max(std::fabs(gt[4]), std::fabs(gt[5])));
    if (std::fabs(det) <= 1e-10 * magnitude * magnitude) return 0;
    const double inv_det = 1.0 / det;
    out[1] = gt[5] * inv_det;
    out[4] = -gt[4] * inv_det;
    out[2] = -gt[2] * inv_det;
    out[5] = gt[1] * inv_det;
    out[0] = (gt[2] * gt[3] - gt[0] * gt[5]) * inv_det;
    out[3] = (-gt[1] * gt[3] + gt[0] * gt[4]) * inv_det;
    return 1;
}

int oracle_world_coordinates(const double* xy, int n, const double* dgt, const double* egt, const double* elev, int ew, int eh, double* xyz) {
    double inv[6] = {0, 0, 0, 0, 0, 0};
    const bool has_elev = egt != nullptr;
    if (has_elev && !oracle_invert_geotransform(egt, inv)) return -5;
    const double a = 6378137.0, f = 1.0 / 298.257223563, es = f * (2.0 - f), deg = 0.017453292519943296;
    int missing = 0;
    for (int i = 0; i < n; i++) {
        const double x = xy[2 * i], y = xy[2 * i + 1];
        const double gx = dgt[0] + x * dgt[1] + y * dgt[2];
        const double gy = dgt[3] + x * dgt[4] + y * dgt[5];
        double h = 0.0;
        if (has_elev) {
            const double px = inv[0] + gx * inv[1] + gy * inv[2];
            const double py = inv[3] + gx * inv[4] + gy * inv[5];
            // elevationdb.rs:240: id = y.round() as i32 * x_size + x.round() as i32 + 1 (1-based row id; no per-axis bounds check)
            const long long id0 = (long long)(int)std::round(py) * ew + (long long)(int)std::round(px);
            if (id0 < 0 || id0 >= (long long)ew * eh) {
                missing = 1;
                xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = NAN;
                continue;
            }
            h = elev[id0];
        }
        // convert_coordinates(coordinates.1, coordinates.0, height): (lat, lon, h) EPSG:4326 -> EPSG:4978
        double sp, cp, sl, cl;
        fixed_sincos(gy * deg, &sp, &cp);
        fixed_sincos(gx * deg, &sl, &cl);
        const double N = a / std::sqrt(1.0 - es * sp * sp);
        xyz[3 * i] = (N + h) * cp * cl;
        xyz[3 * i + 1] = (N + h) * cp * sl;
        xyz[3 * i + 2] = (N * (1.0 - es) + h) * sp;
    }
    return missing ? -211 : 0;
}

}  // extern "C"

// The same map for the other element types the generic reference function admits (warp_image_perspective<T: DataType>): `channels`
// interleaved u8 (the fixed-point path above) or f32 elements. f32: cv::remap's float bilinear path - the four weights are float products
// of the 1/32-step fractions (initInterTab2D), the value ((v0 w0 + v1 w1) + v2 w2) + v3 w3 in binary32, a destination pixel whose 2x2
// footprint lies wholly outside the source is the border value itself. Restated from memory (OpenCV's SIMD variants may fuse the
// multiply-adds: PARITY UNPINNED beyond the identity warp, mod.rs:683-707).
int oracle_warp_perspective_any(const void* src_, int rows, int cols, int channels, int elem_bytes, const double* Min, int dst_rows, int dst_cols, void* dst_) {
    double M[9];
    if (!invert3x3(Min, M)) return -1;
    if ((elem_bytes != 1 && elem_bytes != 4) || channels < 1 || channels > 4) return -2;
    const short* tab = bilinear_tab();
    for (int y = 0; y < dst_rows; y++)
        for (int x = 0; x < dst_cols; x++) {
            const double X0 = M[0] * x + M[1] * y + M[2], Y0 = M[3] * x + M[4] * y + M[5];
            double W = M[6] * x + M[7] * y + M[8];
            W = W ? 32. / W : 0;
            const double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, X0 * W));
            const double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, Y0 * W));
            const int X = sat_int(fX), Y = sat_int(fY);
            const int sx = X >> 5, sy = Y >> 5;
            bool in[4];
            size_t at[4];
            bool any = false;
            for (int k = 0; k < 4; k++) {
                const int xx = sx + (k & 1), yy = sy + (k >> 1);
                in[k] = xx >= 0 && xx < cols && yy >= 0 && yy < rows;
                at[k] = in[k] ? ((size_t)yy * cols + xx) * channels : 0;
                any = any || in[k];
            }
            const size_t o = ((size_t)y * dst_cols + x) * channels;
            if (elem_bytes == 1) {
                const uint8_t* src = static_cast<const uint8_t*>(src_);
                uint8_t* dst = static_cast<uint8_t*>(dst_);
                const short* w = &tab[((Y & 31) * 32 + (X & 31)) * 4];
                for (int c = 0; c < channels; c++) {
                    int v = 0;
                    for (int k = 0; k < 4; k++) v += (in[k] ? src[at[k] + c] : 1) * w[k];
                    dst[o + c] = (uint8_t)((v + (1 << 14)) >> 15);
                }
            } else {
                const float* src = static_cast<const float*>(src_);
                float* dst = static_cast<float*>(dst_);
                if (!any) {
                    for (int c = 0; c < channels; c++) dst[o + c] = 1.0f;
                    continue;
                }
                const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
                const float w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
                for (int c = 0; c < channels; c++) {
                    float acc = (in[0] ? src[at[0] + c] : 1.0f) * w[0];
                    acc += (in[1] ? src[at[1] + c] : 1.0f) * w[1];
                    acc += (in[2] ? src[at[2] + c] : 1.0f) * w[2];
                    acc += (in[3] ? src[at[3] + c] : 1.0f) * w[3];
                    dst[o + c] = acc;
                }
            }
        }
    return 0;
}
