// tests/cpp/pipeline_test.cpp — a host that is NOT Python streams frames through the library's pipeline (apds_pipeline_create / _submit /
// _poll / _stats / _destroy, include/apds.h) and must get, frame by frame, what the one-call entry points give:
//   apds_dev_akaze_extract -> apds_dev_hamming_topk (k = 2) -> apds_dev_ratio_filter -> apds_dev_points_from_matches -> apds_dev_find_homography.
// Built by g++ against libapds_hip.so (tests/test_pipeline_native.py), no torch, no HIP headers; run on the GPU box. 24 frames (device
// frames and host frames alternating, a blank frame among them), then a second batch on the same pipeline, then the counters.
#include <apds.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

namespace {

int failures = 0;
#define CHECK(cond, ...)                                        \
    do {                                                        \
        if (!(cond)) {                                          \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                       \
            fprintf(stderr, "\n");                              \
            failures++;                                         \
        }                                                       \
    } while (0)
#define OK(call)                                                                                               \
    do {                                                                                                       \
        const int rc_ = (call);                                                                                \
        if (rc_ != 0) {                                                                                        \
            fprintf(stderr, "FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, apds_last_error()); \
            failures++;                                                                                        \
        }                                                                                                      \
    } while (0)

struct SplitMix {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// a grey BGRA image of Gaussian blobs (enough structure for a few thousand AKAZE keypoints)
std::vector<uint8_t> blob_frame(int T, uint64_t seed) {
    std::vector<float> v((size_t)T * T, 128.f);
    SplitMix g{seed};
    const int blobs = 2200 * T / 1024 * T / 1024 + 300;
    for (int b = 0; b < blobs; b++) {
        const double cx = g.uni() * T, cy = g.uni() * T, sg = 1.5 + g.uni() * 7.0, amp = -80 + g.uni() * 160;
        const int r = (int)(3 * sg) + 1;
        for (int y = std::max(0, (int)cy - r); y <= std::min(T - 1, (int)cy + r); y++)
            for (int x = std::max(0, (int)cx - r); x <= std::min(T - 1, (int)cx + r); x++) {
                const double d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
                v[(size_t)y * T + x] += (float)(amp * std::exp(-d2 / (2 * sg * sg)));
            }
    }
    std::vector<uint8_t> img((size_t)T * T * 4);
    for (size_t i = 0; i < v.size(); i++) {
        const uint8_t u = (uint8_t)std::min(255.f, std::max(0.f, v[i] + 0.5f));
        img[i * 4] = img[i * 4 + 1] = img[i * 4 + 2] = u;
        img[i * 4 + 3] = 255;
    }
    return img;
}

std::vector<uint8_t> rolled(const std::vector<uint8_t>& img, int T, int dy, int dx) {
    std::vector<uint8_t> out(img.size());
    for (int y = 0; y < T; y++)
        for (int x = 0; x < T; x++) std::memcpy(&out[((size_t)((y + dy) % T) * T + (x + dx) % T) * 4], &img[((size_t)y * T + x) * 4], 4);
    return out;
}

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { OK(apds_dev_alloc(bytes, &p)); }
    ~DevBuf() { apds_dev_release(p); }
    DevBuf(const DevBuf&) = delete;
};

struct Expect {
    int K = 0, M = 0, inliers = 0, found = 0;
    double H[9] = {0};
};

}  // namespace

int main() {
    if (apds_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    OK(apds_set_device(0));
    const int T = 512, NF = 4, NDB = 40000, CAP = 20000;
    const float ratio = 0.3f;
    std::vector<std::vector<uint8_t>> frames;
    for (int f = 0; f < NF - 1; f++) frames.push_back(blob_frame(T, 0xF00D + (uint64_t)f));
    frames.emplace_back((size_t)T * T * 4, 0);   // a blank frame: no keypoints, no queries, no homography
    for (size_t i = 3; i < frames.back().size(); i += 4) frames.back()[i] = 255;
    const size_t fbytes = (size_t)T * T * 4;
    std::vector<std::unique_ptr<DevBuf>> dframes;
    for (auto& f : frames) {
        dframes.emplace_back(new DevBuf(fbytes));
        OK(apds_dev_upload(dframes.back()->p, f.data(), fbytes, nullptr));
    }
    // the train set: descriptors (and keypoints) of a shifted copy of every frame, then random rows
    DevBuf kps((size_t)CAP * 28), desc((size_t)CAP * 64), tmp(fbytes), db((size_t)NDB * 64), dbk((size_t)NDB * 28);
    std::vector<uint8_t> db_rows((size_t)NDB * 64, 0);
    std::vector<apds_keypoint> db_kps((size_t)NDB);
    int P = 0;
    for (int f = 0; f < NF - 1; f++) {
        std::vector<uint8_t> r = rolled(frames[(size_t)f], T, 19, 23);
        OK(apds_dev_upload(tmp.p, r.data(), fbytes, nullptr));
        int n = 0;
        OK(apds_dev_akaze_extract(tmp.p, T, T, 4, (size_t)T * 4, CAP, kps.p, desc.p, CAP, &n, nullptr));
        CHECK(n > 300 && P + n < NDB, "shifted frame %d gives %d keypoints", f, n);
        OK(apds_dev_download(&db_rows[(size_t)P * 64], desc.p, (size_t)n * 64, nullptr));
        OK(apds_dev_download(&db_kps[(size_t)P], kps.p, (size_t)n * 28, nullptr));
        P += n;
    }
    SplitMix g{0xDB};
    for (int i = P; i < NDB; i++) {
        uint64_t* w = reinterpret_cast<uint64_t*>(&db_rows[(size_t)i * 64]);
        for (int j = 0; j < 8; j++) w[j] = g.next();
        db_rows[(size_t)i * 64 + 60] &= 0x3F;
        db_rows[(size_t)i * 64 + 61] = db_rows[(size_t)i * 64 + 62] = db_rows[(size_t)i * 64 + 63] = 0;
        db_kps[(size_t)i] = apds_keypoint{0, 0, 0, 0, 0, 0, 0};
    }
    OK(apds_dev_upload(db.p, db_rows.data(), db_rows.size(), nullptr));
    OK(apds_dev_upload(dbk.p, db_kps.data(), db_kps.size() * 28, nullptr));
    OK(apds_stream_synchronize(nullptr));

    // what the one-call entry points give per frame
    std::vector<Expect> want((size_t)NF);
    {
        DevBuf keys((size_t)CAP * 16), matches((size_t)CAP * 16), p1((size_t)CAP * 8), p2((size_t)CAP * 8), mask(CAP);
        for (int f = 0; f < NF; f++) {
            Expect& e = want[(size_t)f];
            OK(apds_dev_akaze_extract(dframes[(size_t)f]->p, T, T, 4, (size_t)T * 4, CAP, kps.p, desc.p, CAP, &e.K, nullptr));
            if (e.K == 0) continue;
            OK(apds_dev_hamming_topk(desc.p, e.K, db.p, NDB, 0, 2, keys.p, nullptr));
            OK(apds_dev_ratio_filter(keys.p, e.K, 2, ratio, matches.p, &e.M, nullptr));
            if (e.M < 4) continue;
            OK(apds_dev_points_from_matches(kps.p, e.K, dbk.p, NDB, matches.p, e.M, 0, p1.p, p2.p, nullptr));
            const int rc = apds_dev_find_homography(p1.p, p2.p, e.M, APDS_HOMOGRAPHY_RANSAC, 3.0, 2000, 0.995, e.H, mask.p, nullptr);
            CHECK(rc == 0 || rc == APDS_ERR_EMPTY, "find_homography -> %d", rc);
            if (rc != 0) continue;
            e.found = 1;
            std::vector<uint8_t> hm((size_t)e.M);
            OK(apds_dev_download(hm.data(), mask.p, (size_t)e.M, nullptr));
            for (uint8_t b : hm) e.inliers += b != 0;
        }
        CHECK(want[0].K > 300 && want[0].M > 50 && want[0].found && std::fabs(want[0].H[2] - 23) < 0.5 && std::fabs(want[0].H[5] - 19) < 0.5,
              "serial frame 0: K %d M %d found %d H02 %.3f H12 %.3f", want[0].K, want[0].M, want[0].found, want[0].H[2], want[0].H[5]);
        CHECK(want[(size_t)NF - 1].K == 0 && !want[(size_t)NF - 1].found, "the blank frame has %d keypoints", want[(size_t)NF - 1].K);
    }

    apds_pipeline_params pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.rows = pp.cols = T;
    pp.channels = 4;
    pp.max_points = CAP;
    pp.filter_strength = ratio;
    pp.homography_method = APDS_HOMOGRAPHY_RANSAC;
    pp.reproj_threshold = 3.0;
    pp.timing = 1;
    void* pipe = nullptr;
    OK(apds_pipeline_create(&pipe, db.p, NDB, 0, nullptr, dbk.p, NDB, &pp));
    if (!pipe) return 1;
    int64_t next = 0;
    for (int batch = 0; batch < 2; batch++) {
        const int count = batch == 0 ? 24 : 9;
        for (int i = 0; i < count; i++) {
            const int f = (int)((next + i) % NF);
            const bool host = ((next + i) % 3) == 1;   // every third frame comes from host memory (uploaded by the extraction worker)
            int64_t id = -1;
            OK(apds_pipeline_submit(pipe, host ? (const void*)frames[(size_t)f].data() : dframes[(size_t)f]->p, (size_t)T * 4, host ? 0 : 1, &id));
            CHECK(id == next + i, "frame id %lld, expected %lld", (long long)id, (long long)(next + i));
        }
        for (int i = 0; i < count; i++) {
            apds_frame_result r;
            OK(apds_pipeline_poll(pipe, &r, 1));
            const Expect& e = want[(size_t)((next + i) % NF)];
            CHECK(r.frame == next + i && r.status == 0, "result %d: frame %lld status %d", i, (long long)r.frame, r.status);
            CHECK(r.n_keypoints == e.K && r.n_matches == e.M && r.n_inliers == e.inliers && r.homography_found == e.found,
                  "frame %lld: pipeline (K %d, M %d, inliers %d, found %d) != one-call entry points (%d, %d, %d, %d)", (long long)r.frame, r.n_keypoints, r.n_matches,
                  r.n_inliers, r.homography_found, e.K, e.M, e.inliers, e.found);
            if (e.found) CHECK(std::memcmp(r.H, e.H, sizeof(e.H)) == 0, "frame %lld: H differs from the one-call result", (long long)r.frame);
        }
        apds_frame_result none;
        CHECK(apds_pipeline_poll(pipe, &none, 1) == APDS_PIPELINE_NOT_READY, "a poll with nothing in flight must not block");
        next += count;
        apds_pipeline_counters st;
        OK(apds_pipeline_stats(pipe, &st, 1));
        CHECK(st.frames_done == next && st.frames_submitted == next, "counters: %lld done of %lld", (long long)st.frames_done, (long long)st.frames_submitted);
        CHECK(st.akaze_extract_calls == count && st.akaze_extract_ms > 0, "timed extractions: %d", st.akaze_extract_calls);
        int matrix_cores = 0;   // the matrix-core matcher runs on one stream; the vector-ALU one splits pre-pass / scan / merge over three
        OK(apds_dev_match_backend(&matrix_cores));
        CHECK(st.hamming_topk_launches >= count / 2 && st.hamming_topk_ms > 0 && st.extract_workers == 2 && st.split_scan == (matrix_cores ? 0 : 1) && st.world == 1,
              "timed scans: %d (%.3f ms), workers %d, split %d", st.hamming_topk_launches, st.hamming_topk_ms, st.extract_workers, st.split_scan);
        printf("batch %d: %d frames ... %s  (main scan %.3f ms x %d, extraction %.3f ms x %d)\n", batch, count, failures ? "FAILED" : "ok", st.hamming_topk_ms,
               st.hamming_topk_launches, st.akaze_extract_ms, st.akaze_extract_calls);
    }
    // a frame geometry the pipeline was not created for is refused by submit, not by a worker
    CHECK(apds_pipeline_submit(pipe, dframes[0]->p, 16, 1, nullptr) == APDS_ERR_ASSERT, "a short stride must be refused");
    OK(apds_pipeline_destroy(pipe));
    // frames in flight at destroy are drained, not dropped mid-kernel
    OK(apds_pipeline_create(&pipe, db.p, NDB, 0, nullptr, dbk.p, NDB, &pp));
    for (int i = 0; i < 5; i++) OK(apds_pipeline_submit(pipe, dframes[(size_t)(i % NF)]->p, (size_t)T * 4, 1, nullptr));
    OK(apds_pipeline_destroy(pipe));
    OK(apds_thread_release());
    printf("%d failed\n", failures);
    return failures ? 1 : 0;
}
