// include/apds.hpp — C++17 host-side mirror of the reference crates `feature_extraction` and `homographier` over the C ABI (apds.h).
//
// The reference is Rust; there is no Rust toolchain in the build image, so the compiled-language host side above the C ABI is this
// header: the crates' public names, argument order, Option/Result shapes and error behaviour, with std::optional for Option and a
// small Result<T, E>. rust_shim/ holds the same bindings as (uncompiled) Rust for a maintainer of the original workspace.
// Every function cites the reference item it mirrors (/root/reference/<crate>/src/...). Header only; link with libapds_hip.so.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <type_traits>
#include <vector>

#include "apds.h"

namespace apds {

/// opencv::Error { code, message }
struct Error {
    int code = 0;
    std::string message;
};
inline Error last_error(int code) { return Error{code, apds_last_error()}; }

/// Rust's Result<T, E>, as much of it as the crates' callers and tests use.
template <class T, class E>
class Result {
public:
    static Result Ok(T v) {
        Result r;
        r.ok_ = std::move(v);
        return r;
    }
    static Result Err(E e) {
        Result r;
        r.err_ = std::move(e);
        return r;
    }
    bool is_ok() const { return ok_.has_value(); }
    bool is_err() const { return !ok_.has_value(); }
    template <class F>
    bool is_err_and(F&& f) const {
        return is_err() && f(*err_);
    }
    T& unwrap() {
        if (!ok_) throw std::runtime_error("called unwrap() on an Err value");
        return *ok_;
    }
    const T& unwrap() const {
        if (!ok_) throw std::runtime_error("called unwrap() on an Err value");
        return *ok_;
    }
    const E& unwrap_err() const {
        if (ok_) throw std::runtime_error("called unwrap_err() on an Ok value");
        return *err_;
    }

private:
    std::optional<T> ok_;
    std::optional<E> err_;
};

struct Point2f {
    float x, y;
};
struct Point2d {
    double x, y;
};
struct Point3d {
    double x, y, z;
};
using Vec4b = std::array<uint8_t, 4>;   // cv::Vec4b
struct RGBA8 {                          // rgb::RGBA8
    uint8_t r, g, b, a;
};
using KeyPoint = apds_keypoint;   // cv::KeyPoint, 28 bytes
using DMatch = apds_dmatch;       // cv::DMatch, 16 bytes

/// A dense row-major matrix standing in for cv::Mat (rows x cols elements of T); rows == 0 is the empty Mat.
template <class T>
struct Mat {
    int rows = 0, cols = 0;
    std::vector<T> data;
    Mat() = default;
    Mat(int r, int c, const T& fill = T{}) : rows(r), cols(c), data((size_t)r * c, fill) {}
    bool empty() const { return rows <= 0 || cols <= 0; }
    const T& at(int r, int c) const { return data[(size_t)r * cols + c]; }
    T& at(int r, int c) { return data[(size_t)r * cols + c]; }
};

namespace homographier {

/// homographier/src/homographier/mod.rs:25-31
enum class HomographyMethod { Default = 0, LMEDS = 4, RANSAC = 8, RHO = 16 };

/// mod.rs:33-44
struct MatError {
    enum Kind { Opencv, Empty, Jagged, Unknown } kind = Unknown;
    Error inner;   // for Opencv
    static MatError opencv(Error e) { return MatError{Opencv, std::move(e)}; }
};

/// mod.rs:66-146 — checked matrix: non-empty by construction.
template <class T>
class Cmat {
public:
    Mat<T> mat;

    /// mod.rs:109-114 (the element type is carried by T here, so only emptiness can fail)
    static Result<Cmat, MatError> new_(Mat<T> m) {
        if (m.empty()) return Result<Cmat, MatError>::Err(MatError{MatError::Empty, {}});   // mod.rs:100-106
        Cmat c;
        c.mat = std::move(m);
        return Result<Cmat, MatError>::Ok(std::move(c));
    }
    /// Mat::from_slice_2d behind Cmat::from_2d_slice: rows of equal length, else an OpenCV error
    static Result<Cmat, MatError> from_2d_slice(const std::vector<std::vector<T>>& rows) {
        Mat<T> m;
        m.rows = (int)rows.size();
        m.cols = rows.empty() ? 0 : (int)rows[0].size();
        for (const auto& r : rows) {
            if ((int)r.size() != m.cols) return Result<Cmat, MatError>::Err(MatError{MatError::Jagged, {}});
            m.data.insert(m.data.end(), r.begin(), r.end());
        }
        return new_(std::move(m));
    }
    /// mod.rs:136-145
    static Result<Cmat, MatError> zeros(int rows, int cols) { return new_(Mat<T>(rows, cols)); }
    /// mod.rs:127-134 — checked element access: anything out of range is Opencv(StsOutOfRange)
    Result<const T*, MatError> at_2d(int row, int col) const {
        if (row < 0 || col < 0 || row >= mat.rows || col >= mat.cols)
            return Result<const T*, MatError>::Err(MatError::opencv(Error{APDS_ERR_OUT_OF_RANGE, ""}));
        return Result<const T*, MatError>::Ok(&mat.at(row, col));
    }
};

inline MatError from_status(int rc) {
    if (rc == APDS_ERR_EMPTY) return MatError{MatError::Empty, {}};   // empty model -> Cmat::new(empty) failed (mod.rs:258,114-119)
    return MatError::opencv(last_error(rc));
}

/// mod.rs:183-197 — RGBA8 slice -> Cmat<Vec4b> in BGRA order; MatError::Unknown when pixels.len() != w*h
inline Result<Cmat<Vec4b>, MatError> raster_to_mat(const std::vector<RGBA8>& pixels, int w, int h) {
    using R = Result<Cmat<Vec4b>, MatError>;
    if (w <= 0 || h <= 0 || pixels.size() != (size_t)w * h) return R::Err(MatError{MatError::Unknown, {}});
    Mat<Vec4b> out(h, w);
    const int rc = apds_raster_to_mat(reinterpret_cast<const uint8_t*>(pixels.data()), pixels.size(), w, h, reinterpret_cast<uint8_t*>(out.data.data()));
    if (rc != 0) return R::Err(rc == APDS_ERR_BAD_ARG ? MatError{MatError::Unknown, {}} : from_status(rc));
    return Cmat<Vec4b>::new_(std::move(out));
}

/// mod.rs:231-259 — (homography 3x3, inlier mask n x 1 for RANSAC / LMEDS only)
inline Result<std::pair<Cmat<double>, std::optional<Cmat<uint8_t>>>, MatError> find_homography_mat(const std::vector<Point2f>& input,
                                                                                                  const std::vector<Point2f>& reference,
                                                                                                  std::optional<HomographyMethod> method,
                                                                                                  std::optional<double> reproj_threshold) {
    using Out = std::pair<Cmat<double>, std::optional<Cmat<uint8_t>>>;
    using R = Result<Out, MatError>;
    if (input.size() != reference.size()) return R::Err(MatError::opencv(Error{APDS_ERR_ASSERT, "point lists differ in length"}));
    const int n = (int)input.size();
    Mat<double> H(3, 3);
    Mat<uint8_t> mask(n > 0 ? n : 1, 1);
    const int rc = apds_find_homography(reinterpret_cast<const float*>(input.data()), reinterpret_cast<const float*>(reference.data()), n,
                                        (int)method.value_or(HomographyMethod::Default), reproj_threshold.value_or(3.0), H.data.data(), mask.data.data());
    if (rc != 0) return R::Err(from_status(rc));
    std::optional<Cmat<uint8_t>> out_mask;
    if (method && (*method == HomographyMethod::RANSAC || *method == HomographyMethod::LMEDS)) {   // mod.rs:253-257
        mask.rows = n;
        mask.data.resize(n);
        auto m = Cmat<uint8_t>::new_(std::move(mask));
        if (m.is_err()) return R::Err(m.unwrap_err());
        out_mask = std::move(m.unwrap());
    }
    auto h = Cmat<double>::new_(std::move(H));
    return R::Ok(Out{std::move(h.unwrap()), std::move(out_mask)});
}

/// mod.rs:271-300 — warpPerspective(INTER_LINEAR, BORDER_CONSTANT (1,1,1,1)); size = (width, height), None = the source size. Generic over
/// the element type like the reference (`warp_image_perspective<T: DataType>`): T = uint8_t, Vec3b, Vec4b, float, Vec3f, Vec4f.
using Vec3b = std::array<uint8_t, 3>;   // cv::Vec3b
using Vec3f = std::array<float, 3>;     // cv::Vec3f
using Vec4f = std::array<float, 4>;     // cv::Vec4f
template <class T>
inline Result<Cmat<T>, MatError> warp_image_perspective(const Cmat<T>& src, const Cmat<double>& m, std::optional<std::pair<int, int>> size) {
    using R = Result<Cmat<T>, MatError>;
    constexpr bool is_u8 = std::is_same_v<T, uint8_t> || std::is_same_v<T, Vec3b> || std::is_same_v<T, Vec4b>;
    constexpr bool is_f32 = std::is_same_v<T, float> || std::is_same_v<T, Vec3f> || std::is_same_v<T, Vec4f>;
    static_assert(is_u8 || is_f32, "warp_image_perspective: u8 or f32 elements with 1, 3 or 4 channels");
    constexpr int channels = (int)(sizeof(T) / (is_u8 ? 1 : 4));
    if (m.mat.rows != 3 || m.mat.cols != 3) return R::Err(MatError::opencv(Error{APDS_ERR_ASSERT, "m must be 3x3"}));
    const int dw = size ? size->first : src.mat.cols, dh = size ? size->second : src.mat.rows;
    Mat<T> out(dh, dw);
    int rc;
    if constexpr (is_u8)
        rc = apds_warp_perspective(reinterpret_cast<const uint8_t*>(src.mat.data.data()), src.mat.rows, src.mat.cols, channels, m.mat.data.data(), dh, dw,
                                   reinterpret_cast<uint8_t*>(out.data.data()));
    else
        rc = apds_warp_perspective_f32(reinterpret_cast<const float*>(src.mat.data.data()), src.mat.rows, src.mat.cols, channels, m.mat.data.data(), dh, dw,
                                       reinterpret_cast<float*>(out.data.data()));
    if (rc != 0) return R::Err(from_status(rc));
    return Cmat<T>::new_(std::move(out));
}

/// opencv::calib3d::SolvePnPMethod values the reference can pass (mod.rs:4,327)
enum class SolvePnPMethod { SOLVEPNP_ITERATIVE = 0, SOLVEPNP_EPNP = 1, SOLVEPNP_P3P = 2, SOLVEPNP_DLS = 3, SOLVEPNP_UPNP = 4, SOLVEPNP_AP3P = 5, SOLVEPNP_IPPE = 6, SOLVEPNP_IPPE_SQUARE = 7, SOLVEPNP_SQPNP = 8 };

/// mod.rs:52-65
struct ImgObjCorrespondence {
    Point3d obj_point;
    Point2d img_point;
    ImgObjCorrespondence(Point3d o, Point2d i) : obj_point(o), img_point(i) {}
};

/// mod.rs:46-51
struct PNPRANSACSolution {
    Cmat<double> rvec, tvec;
    Cmat<int32_t> inliers;
};

/// mod.rs:320-369 — solvePnPRansac(useExtrinsicGuess = false); dist_coeffs is accepted and ignored, as the reference shadows it with
/// zeros(4,1) (mod.rs:344). Ok(nullopt) when no pose was found.
inline Result<std::optional<PNPRANSACSolution>, MatError> pnp_solver_ransac(const std::vector<ImgObjCorrespondence>& point_correspondences,
                                                                            const Cmat<double>& camera_intrinsic, int iter_count, float reproj_thres,
                                                                            double confidence, std::optional<std::vector<double>> /*dist_coeffs*/,
                                                                            std::optional<SolvePnPMethod> method) {
    using R = Result<std::optional<PNPRANSACSolution>, MatError>;
    const int n = (int)point_correspondences.size();
    std::vector<double> obj, img;
    obj.reserve(3 * (size_t)n);
    img.reserve(2 * (size_t)n);
    for (const auto& p : point_correspondences) {   // mod.rs:329-335
        obj.insert(obj.end(), {p.obj_point.x, p.obj_point.y, p.obj_point.z});
        img.insert(img.end(), {p.img_point.x, p.img_point.y});
    }
    if (camera_intrinsic.mat.rows != 3 || camera_intrinsic.mat.cols != 3)
        return R::Err(MatError::opencv(Error{APDS_ERR_ASSERT, "camera_intrinsic must be 3x3"}));
    Mat<double> rvec(3, 1), tvec(3, 1);
    std::vector<int32_t> inl((size_t)(n > 0 ? n : 1));
    int n_inl = 0, found = 0;
    const int rc = apds_pnp_solver_ransac(obj.data(), img.data(), n, camera_intrinsic.mat.data.data(), iter_count, reproj_thres, confidence,
                                          (int)method.value_or(SolvePnPMethod::SOLVEPNP_EPNP),   // mod.rs:360
                                          rvec.data.data(), tvec.data.data(), inl.data(), &n_inl, &found);
    if (rc != 0) return R::Err(MatError::opencv(last_error(rc)));
    if (!found) return R::Ok(std::nullopt);   // res.then_some(solution), mod.rs:367
    Mat<int32_t> im(n_inl, 1);
    std::memcpy(im.data.data(), inl.data(), (size_t)n_inl * sizeof(int32_t));
    PNPRANSACSolution sol{Cmat<double>::new_(std::move(rvec)).unwrap(), Cmat<double>::new_(std::move(tvec)).unwrap(),
                          Cmat<int32_t>::new_(std::move(im)).unwrap()};
    return R::Ok(std::optional<PNPRANSACSolution>(std::move(sol)));
}

}  // namespace homographier

namespace feature_extraction {

/// feature_extraction/src/lib.rs:12-13
constexpr int MAX_POINTS_SHIFT = APDS_MAX_POINTS_SHIFT;
constexpr int MAX_POINTS = APDS_MAX_POINTS;

/// lib.rs:20-32
struct DbKeypoints {
    float x_coord, y_coord, size, angle, response;
    int32_t octave, class_id;
    std::vector<uint8_t> descriptor;
    int32_t image_id;
};

/// lib.rs:15-18, 34-58 — descriptors: one 61-byte row per keypoint
struct ExtractedKeyPoint {
    std::vector<KeyPoint> keypoints;
    Mat<uint8_t> descriptors;
    std::vector<DbKeypoints> to_db_type(int32_t image_id) const {
        std::vector<DbKeypoints> out;
        out.reserve(keypoints.size());
        for (size_t i = 0; i < keypoints.size(); i++) {
            const KeyPoint& k = keypoints[i];
            const uint8_t* d = &descriptors.data[i * (size_t)descriptors.cols];
            out.push_back(DbKeypoints{k.x, k.y, k.size, k.angle, k.response, k.octave, k.class_id, std::vector<uint8_t>(d, d + descriptors.cols), image_id});
        }
        return out;
    }
};

/// lib.rs:61-92 — AKAZE(MLDB, 0, 3, 0.001, 4 octaves, 4 layers, PM_G2, max_points or MAX_POINTS).detectAndCompute on a u8 image
/// (rows x cols x channels, channels in {1, 3, 4}; the preprocessor passes the BGRA Mat of raster_to_mat)
inline Result<ExtractedKeyPoint, Error> akaze_keypoint_descriptor_extraction_def(const uint8_t* img, int rows, int cols, int channels, size_t stride_bytes,
                                                                                std::optional<int> max_points) {
    using R = Result<ExtractedKeyPoint, Error>;
    apds_keypoint* kps = nullptr;
    uint8_t* desc = nullptr;
    int n = 0, nb = 0;
    const int rc = apds_akaze_extract(img, rows, cols, channels, stride_bytes, max_points.value_or(MAX_POINTS), &kps, &desc, &n, &nb);
    if (rc != 0) return R::Err(last_error(rc));
    ExtractedKeyPoint out;
    out.keypoints.assign(kps, kps + n);
    out.descriptors.rows = n;
    out.descriptors.cols = nb;
    out.descriptors.data.assign(desc, desc + (size_t)n * nb);
    apds_free(kps);
    apds_free(desc);
    return R::Ok(std::move(out));
}
inline Result<ExtractedKeyPoint, Error> akaze_keypoint_descriptor_extraction_def(const Mat<Vec4b>& img, std::optional<int> max_points) {
    return akaze_keypoint_descriptor_extraction_def(reinterpret_cast<const uint8_t*>(img.data.data()), img.rows, img.cols, 4, (size_t)img.cols * 4, max_points);
}

/// lib.rs:94-114 — BFMatcher(HAMMING).knnMatch + ratio filter m[0].distance < m[1].distance * filter_strength
inline Result<std::vector<DMatch>, Error> get_knn_matches(const Mat<uint8_t>& origin_desc, const Mat<uint8_t>& target_desc, int k, float filter_strength) {
    using R = Result<std::vector<DMatch>, Error>;
    apds_dmatch* m = nullptr;
    int n = 0;
    const int rc = apds_get_knn_matches(origin_desc.data.data(), origin_desc.rows, target_desc.data.data(), target_desc.rows, origin_desc.cols, k, filter_strength,
                                        &m, &n);
    if (rc != 0) return R::Err(last_error(rc));
    std::vector<DMatch> out(m, m + n);
    apds_free(m);
    return R::Ok(std::move(out));
}

/// lib.rs:116-126 — BFMatcher(HAMMING, crossCheck = true).match
inline Result<std::vector<DMatch>, Error> get_bruteforce_matches(const Mat<uint8_t>& origin_desc, const Mat<uint8_t>& target_desc) {
    using R = Result<std::vector<DMatch>, Error>;
    apds_dmatch* m = nullptr;
    int n = 0;
    const int rc = apds_get_bruteforce_matches(origin_desc.data.data(), origin_desc.rows, target_desc.data.data(), target_desc.rows, origin_desc.cols, &m, &n);
    if (rc != 0) return R::Err(last_error(rc));
    std::vector<DMatch> out(m, m + n);
    apds_free(m);
    return R::Ok(std::move(out));
}

/// lib.rs:161-180 — matched keypoint coordinates of both images (the intended gather; bug_compatible = true reproduces :169,:176-177)
inline Result<std::pair<std::vector<Point2f>, std::vector<Point2f>>, Error> get_points_from_matches(const std::vector<KeyPoint>& img1_keypoints,
                                                                                                    const std::vector<KeyPoint>& img2_keypoints,
                                                                                                    const std::vector<DMatch>& matches,
                                                                                                    bool bug_compatible = false) {
    using Out = std::pair<std::vector<Point2f>, std::vector<Point2f>>;
    using R = Result<Out, Error>;
    Out out;
    out.first.resize(matches.size());
    out.second.resize(matches.size());
    const int rc = apds_get_points_from_matches(img1_keypoints.data(), (int)img1_keypoints.size(), img2_keypoints.data(), (int)img2_keypoints.size(),
                                                matches.data(), (int)matches.size(), bug_compatible ? 1 : 0, reinterpret_cast<float*>(out.first.data()),
                                                reinterpret_cast<float*>(out.second.data()));
    if (rc != 0) return R::Err(last_error(rc));
    return R::Ok(std::move(out));
}

}  // namespace feature_extraction
}  // namespace apds
