// tests/cpp/reference_tests.cpp — the reference's own unit tests for the hot-path crates, restated against the C++ host mirror
// (include/apds.hpp). Each TEST cites the #[test] it restates; the pipeline test at the end chains the crates as the reference's
// feature_extraction tests do (lib.rs:197-249), on a synthetic tile instead of the git-ignored GeoTIFFs.
// Build + run: tests/test_cpp_mirror.py (needs a GPU to run; compiles anywhere).
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>

#include "apds.hpp"

using namespace apds;
using namespace apds::homographier;

static int g_failed = 0;
#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::printf("    CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            g_failed++;                                                                \
            return;                                                                    \
        }                                                                              \
    } while (0)

static Cmat<Vec4b> test_image(size_t size) {   // mod.rs:391-406
    std::vector<RGBA8> image(size * size, RGBA8{1, 1, 1, 1});
    for (size_t p = 0; p < image.size(); p++) {
        const size_t row = p / size + 1, col = p % size + 1;
        image[p] = RGBA8{image[p].r, (uint8_t)(image[p].g * col), (uint8_t)(image[p].b * row), image[p].a};
    }
    return raster_to_mat(image, (int)size, (int)size).unwrap();
}

static Cmat<double> empty_homography() {   // mod.rs:431-435
    return Cmat<double>::from_2d_slice({{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}).unwrap();
}

// mod.rs:437-472
static void homography_success() {
    std::vector<Point2f> points;
    for (int i = 1; i <= 10; i++)
        for (int j = 1; j <= 10; j++) points.push_back(Point2f{(float)i, (float)j});
    auto res = find_homography_mat(points, points, HomographyMethod::RANSAC, 1.0);
    CHECK(res.is_ok());
    const auto& homography = res.unwrap().first;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++) CHECK(std::round(*homography.at_2d(row, col).unwrap()) == (col == row ? 1.0 : 0.0));
    CHECK(res.unwrap().second.has_value() && res.unwrap().second->mat.rows == 100);
}

// mod.rs:475-477
static void cmat_init() { CHECK(Cmat<Vec4b>::new_(Mat<Vec4b>()).is_err()); }

// mod.rs:478-486
static void cmat_init_2d() {
    auto cmat = Cmat<Vec4b>::new_(Mat<Vec4b>(10, 10));
    CHECK(cmat.is_ok() && cmat.unwrap().mat.rows == 10 && cmat.unwrap().mat.cols == 10);
}

// mod.rs:515-553
static void cmat_from_slice() {
    const size_t IMG_SIZE = 4;
    std::vector<std::vector<Vec4b>> image(IMG_SIZE);
    for (size_t i = 0; i < IMG_SIZE; i++)
        for (size_t j = 0; j < IMG_SIZE; j++) {
            const uint8_t s = (uint8_t)(1 + j);
            image[i].push_back(Vec4b{(uint8_t)(1 * s), (uint8_t)(2 * s), (uint8_t)(3 * s), (uint8_t)(4 * s)});
        }
    auto cmat = Cmat<Vec4b>::from_2d_slice(image).unwrap();
    CHECK((cmat.mat.at(0, 0) == Vec4b{1, 2, 3, 4}));
    CHECK((cmat.mat.at((int)IMG_SIZE - 1, (int)IMG_SIZE - 1) == Vec4b{4, 8, 12, 16}));
}

// mod.rs:556-603
static void raster_to_mat_works() {
    const size_t IMG_SIZE = 4;
    std::vector<RGBA8> image(IMG_SIZE * IMG_SIZE, RGBA8{1, 1, 1, 1});
    for (size_t p = 0; p < image.size(); p++) {
        const size_t row = p / IMG_SIZE + 1, col = p % IMG_SIZE + 1;
        image[p] = RGBA8{1, (uint8_t)col, (uint8_t)row, 1};
    }
    auto res = raster_to_mat(image, (int)IMG_SIZE, (int)IMG_SIZE);
    CHECK(res.is_ok());
    const auto& m = res.unwrap().mat;   // BGRA, row major
    CHECK((m.at(0, 0) == Vec4b{1, 1, 1, 1}));
    CHECK((m.at(3, 3) == Vec4b{4, 4, 1, 1}));
    CHECK((m.at(0, 3) == Vec4b{1, 4, 1, 1}));
    CHECK((m.at(3, 0) == Vec4b{4, 1, 1, 1}));
    CHECK(raster_to_mat(image, 5, 4).is_err_and([](const MatError& e) { return e.kind == MatError::Unknown; }));   // mod.rs:185-187
}

// mod.rs:606-625
static void cmat_at_2d_works() {
    auto image = test_image(4);
    auto out_of_range = [](const MatError& x) { return x.kind == MatError::Opencv && x.inner.code == -211; };
    CHECK(image.at_2d(3, 5).is_err_and(out_of_range));
    CHECK(image.at_2d(5, 3).is_err_and(out_of_range));
    CHECK((*image.at_2d(3, 3).unwrap() == Vec4b{4, 4, 1, 1}));
}

// mod.rs:627-638
static void pnp_solver_ransac_no_work_lthan_3_points() {
    std::vector<ImgObjCorrespondence> corres_v{ImgObjCorrespondence(Point3d{1, 2, 3}, Point2d{1, 2}), ImgObjCorrespondence(Point3d{4, 5, 6}, Point2d{4, 5})};
    auto camera_intrinsic = Cmat<double>::zeros(3, 3).unwrap();
    auto res = pnp_solver_ransac(corres_v, camera_intrinsic, 50, 2.0f, 0.99, std::nullopt, std::nullopt);
    CHECK(res.is_err());
}

// mod.rs:640-682 (#[ignore]d upstream: "needs AKAZE keypoints"); here it runs: no errors, and a pose or Ok(None)
static void pnp_solver_works() {
    std::vector<ImgObjCorrespondence> corres_v{
        ImgObjCorrespondence(Point3d{0, 5, 1}, Point2d{-1.48, 0.39}), ImgObjCorrespondence(Point3d{5, 0, 0}, Point2d{2.14, -1.92}),
        ImgObjCorrespondence(Point3d{5, 5, 1.5}, Point2d{1.74, 0.56}), ImgObjCorrespondence(Point3d{0, 0, 1}, Point2d{-2, -1.62}),
        ImgObjCorrespondence(Point3d{2, 8, -2}, Point2d{-0.16, 0.3})};
    auto camera_intrinsic = Cmat<double>::from_2d_slice({{8.64, 0, 0}, {0, 8.64, 0}, {0, 0, 1}}).unwrap();   // camera_matrix(), mod.rs:408-427
    auto res = pnp_solver_ransac(corres_v, camera_intrinsic, 10000, 100.0f, 0.5, std::nullopt, SolvePnPMethod::SOLVEPNP_P3P);
    CHECK(res.is_ok());
}

// mod.rs:683-707
static void warp_image_empty() {
    const int SIZE = 4;
    auto image = test_image(SIZE);
    auto warped = warp_image_perspective(image, empty_homography(), std::nullopt);
    CHECK(warped.is_ok());
    for (int row = 0; row < SIZE; row++)
        for (int col = 0; col < SIZE; col++) CHECK(*image.at_2d(row, col).unwrap() == *warped.unwrap().at_2d(row, col).unwrap());
}

// the chain of feature_extraction's tests (lib.rs:197-249: extract two images, knn / brute-force match, matched points), then
// find_homography_mat on the result: a textured tile and a copy shifted by (7, 4) pixels
static void extract_match_homography_on_a_shifted_tile() {
    namespace fe = apds::feature_extraction;
    const int S = 384, dx = 7, dy = 4;
    std::vector<RGBA8> a((size_t)S * S), b((size_t)S * S);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return (uint32_t)((z ^ (z >> 31)) >> 40);
    };
    std::vector<float> field((size_t)(S + 16) * (S + 16), 128.f);
    for (int blob = 0; blob < 900; blob++) {   // random soft blobs: corner-rich texture
        const int cx = rnd() % (S + 16), cy = rnd() % (S + 16), r = 2 + rnd() % 6;
        const float amp = (float)(rnd() % 160) - 80.f;
        for (int y = std::max(0, cy - r); y < std::min(S + 16, cy + r); y++)
            for (int x = std::max(0, cx - r); x < std::min(S + 16, cx + r); x++) {
                const float d2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                field[(size_t)y * (S + 16) + x] += amp * std::exp(-d2 / (0.5f * r * r));
            }
    }
    auto px = [&](int x, int y) {
        const float v = std::min(255.f, std::max(0.f, field[(size_t)y * (S + 16) + x]));
        return RGBA8{(uint8_t)v, (uint8_t)(0.8f * v), (uint8_t)(255.f - v), 255};
    };
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++) {
            a[(size_t)y * S + x] = px(x, y);
            b[(size_t)y * S + x] = px(x + dx, y + dy);
        }
    auto ma = raster_to_mat(a, S, S).unwrap(), mb = raster_to_mat(b, S, S).unwrap();
    auto ea = fe::akaze_keypoint_descriptor_extraction_def(ma.mat, std::nullopt);
    auto eb = fe::akaze_keypoint_descriptor_extraction_def(mb.mat, std::nullopt);
    CHECK(ea.is_ok() && eb.is_ok());
    CHECK(ea.unwrap().keypoints.size() > 100 && ea.unwrap().descriptors.cols == 61);
    CHECK(ea.unwrap().to_db_type(7).size() == ea.unwrap().keypoints.size() && ea.unwrap().to_db_type(7)[0].image_id == 7);
    auto knn = fe::get_knn_matches(ea.unwrap().descriptors, eb.unwrap().descriptors, 2, 0.7f);
    auto bf = fe::get_bruteforce_matches(ea.unwrap().descriptors, eb.unwrap().descriptors);
    CHECK(knn.is_ok() && bf.is_ok() && knn.unwrap().size() > 40 && bf.unwrap().size() >= knn.unwrap().size() / 2);
    auto pts = fe::get_points_from_matches(ea.unwrap().keypoints, eb.unwrap().keypoints, knn.unwrap());
    CHECK(pts.is_ok());
    auto h = find_homography_mat(pts.unwrap().first, pts.unwrap().second, HomographyMethod::RANSAC, 2.0);
    CHECK(h.is_ok());
    const auto& H = h.unwrap().first.mat;   // image a -> image b: x' = x - dx, y' = y - dy
    CHECK(std::fabs(H.at(0, 0) - 1) < 0.02 && std::fabs(H.at(1, 1) - 1) < 0.02 && std::fabs(H.at(0, 1)) < 0.02 && std::fabs(H.at(1, 0)) < 0.02);
    CHECK(std::fabs(H.at(0, 2) + dx) < 0.75 && std::fabs(H.at(1, 2) + dy) < 0.75);
    // k = 1 cannot feed the ratio test: i.get(1)? fails in the reference (lib.rs:107-111)
    CHECK(fe::get_knn_matches(ea.unwrap().descriptors, eb.unwrap().descriptors, 1, 0.7f).is_err());
}

int main() {
    const std::pair<const char*, std::function<void()>> tests[] = {
        {"homography_success", homography_success},
        {"cmat_init", cmat_init},
        {"cmat_init_2d", cmat_init_2d},
        {"cmat_from_slice", cmat_from_slice},
        {"raster_to_mat_works", raster_to_mat_works},
        {"cmat_at_2d_works", cmat_at_2d_works},
        {"pnp_solver_ransac_no_work_lthan_3_points", pnp_solver_ransac_no_work_lthan_3_points},
        {"pnp_solver_works", pnp_solver_works},
        {"warp_image_empty", warp_image_empty},
        {"extract_match_homography_on_a_shifted_tile", extract_match_homography_on_a_shifted_tile},
    };
    for (const auto& t : tests) {
        const int before = g_failed;
        t.second();
        std::printf("test %s ... %s\n", t.first, g_failed == before ? "ok" : "FAILED");
    }
    std::printf("%d failed\n", g_failed);
    return g_failed ? 1 : 0;
}
