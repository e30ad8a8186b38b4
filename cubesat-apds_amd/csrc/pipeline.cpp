// csrc/pipeline.cpp — the streamed frame pipeline behind the C ABI (apds_pipeline_*, include/apds.h).
//
// frame -> AKAZE -> Hamming top-2 against the resident descriptor DB -> ratio test -> matched points -> homography, software-pipelined
// over a stream of frames by host threads that live INSIDE the library, so that a host which is not Python (the reference's host is
// Rust: preprocessor/src/main.rs:227-245 calls the path from a rayon pool, no lock) reaches the same frames/s bench.py reports:
//
//   extraction workers (2)   frame i+1, i+2: upload (host frames), AKAZE; HBM-bound stencils at raised wave priority
//   match worker (1)         frame i: threshold pre-pass | main scan | record merge on three streams, so the main scans of consecutive
//                            frames follow each other without the pre-pass, the merge and the dependent-launch gaps between them;
//                            with a row-sharded DB (apds_shard_*): query gather of frame i+1 before the key exchange of frame i
//   homography worker (1)    frame i-1: ratio filter, point gather, RANSAC (its host round trips hide behind the other two stages)
//
// Every worker owns a HIP stream and a device workspace (the per-thread context every entry point uses), stages hand frames over
// through HIP events, results leave in frame order. The reference only chains these steps inside unit tests
// (feature_extraction/src/lib.rs:197-249) and never calls find_homography_mat on the result; this is the composed path the
// north-star metric (frames/s) is measured on. The Python class cubesat-apds_amd/pipeline.py:StreamedFramePipeline is a front of this.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "config.h"
#include "kernels.h"

using namespace apds;

namespace {

// a blocking queue; close() wakes every waiter (pop then returns false once the queue is empty)
template <class T>
class Chan {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;

public:
    void push(T v) {
        {
            std::lock_guard<std::mutex> g(m);
            q.push_back(std::move(v));
        }
        cv.notify_one();
    }
    bool pop(T& out) {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        out = std::move(q.front());
        q.pop_front();
        return true;
    }
    void close() {
        {
            std::lock_guard<std::mutex> g(m);
            closed = true;
        }
        cv.notify_all();
    }
};

struct StageError {
    int code;
    std::string msg;
};
[[noreturn]] void stage_fail(int rc) { throw StageError{rc, apds_last_error()}; }
#define PIPE_OK(call)                       \
    do {                                    \
        const int rc_ = (call);             \
        if (rc_ != APDS_OK) stage_fail(rc_); \
    } while (0)

struct Slot {
    // device buffers of one frame in flight
    apds_keypoint* kps = nullptr;
    uint8_t* desc = nullptr;
    uint64_t* keys = nullptr;
    apds_dmatch* matches = nullptr;
    float *p1 = nullptr, *p2 = nullptr;
    uint8_t* mask = nullptr;
    uint8_t* frame_dev = nullptr;   // staging buffer of a host frame
    size_t frame_dev_bytes = 0;
    hipEvent_t ev_extract = nullptr, ev_pre = nullptr, ev_scan = nullptr, ev_match = nullptr;
    hipEvent_t ev_mstart = nullptr, ev_mend = nullptr;   // (timing enabled) the match stream's idle gaps: starvation watch
    void* topk_state = nullptr;   // split scan (one GPU)
    void* shard_slot = nullptr;   // exchange slot (sharded DB)
    int owner = 0;
    // the job
    const void* src = nullptr;
    size_t stride = 0;
    bool on_device = false;
    int64_t index = 0;
    int K = 0, status = APDS_OK;
    std::string error;
    std::vector<int> counts;
};

struct Pipeline {
    apds_pipeline_params p{};
    int device = 0;
    const void* db_rows = nullptr;
    void* db_expanded = nullptr;   // the train rows as matrix-core operands (hm_train_create), made once: the DB is resident for the pipeline's life
    int64_t n_rows = 0;
    uint32_t index_base = 0;
    void* shard = nullptr;
    int world = 1;
    const apds_keypoint* db_kps = nullptr;
    int64_t n_db_total = 0;
    int cap = 0, E = 2;
    bool split = true, adaptive_cap = true, own_match_stream = true;
    double extract_delay_s = 0;

    std::vector<std::unique_ptr<Slot>> slots;
    std::vector<std::unique_ptr<Chan<Slot*>>> q_free, q_jobs, q_ext;
    Chan<int> q_any;          // which worker finished a frame (arrival order; one GPU)
    Chan<Slot*> q_homography;
    std::vector<std::thread> threads;
    hipStream_t st_extract[8] = {}, st_match = nullptr, st_pre = nullptr, st_merge = nullptr, st_homography = nullptr, st_gather = nullptr, st_counts = nullptr;

    std::mutex m;   // results, counters, errors
    std::condition_variable cv;
    std::map<int64_t, apds_frame_result> results;
    std::map<int64_t, std::string> result_errors;
    int64_t next_submit = 0, next_result = 0, done = 0;
    bool failed = false, closing = false;
    int fail_code = 0;
    std::string fail_msg;
    // statistics (timing enabled)
    std::map<std::string, std::pair<double, int>> timers;
    std::vector<float> gaps;
    int cap_bytes = 0, cap_frame = -1;
    std::vector<float> cap_gaps;   // the six gaps that triggered the cap
    int harvest_acks = 0;
    std::mutex submit_m;   // one submitter at a time keeps frame numbers and slot hand-out in step

    void fail(int code, const std::string& msg) {
        {
            std::lock_guard<std::mutex> g(m);
            if (!failed) {
                failed = true;
                fail_code = code;
                fail_msg = msg;
            }
        }
        close_all();
        cv.notify_all();
    }
    void close_all() {
        for (auto& q : q_free) q->close();
        for (auto& q : q_jobs) q->close();
        for (auto& q : q_ext) q->close();
        q_any.close();
        q_homography.close();
    }
    void add_timer(const char* name, double ms, int n) {
        std::lock_guard<std::mutex> g(m);
        auto& t = timers[name];
        t.first += ms;
        t.second += n;
    }
    // this worker thread's event timers of the named kernels -> the pipeline's totals (the events were recorded on the launch streams)
    void collect(std::initializer_list<const char*> names) {
        if (!p.timing) return;
        for (const char* n : names) {
            float ms = 0;
            int k = 0;
            if (apds_dev_last_kernel_ms(n, &ms, &k) == APDS_OK && k > 0) add_timer(n, ms, k);
        }
    }

    void harvest(std::initializer_list<const char*> names) {
        collect(names);
        {
            std::lock_guard<std::mutex> g(m);
            harvest_acks++;
        }
        cv.notify_all();
    }

    template <class F>
    void worker(const char* what, F body) {
        try {
            PIPE_OK(apds_set_device(device));
            PIPE_OK(apds_dev_timing_enable(p.timing ? 1 : 0));
            body();
        } catch (const StageError& e) {
            fail(e.code, std::string(what) + ": " + e.msg);
        } catch (const Error& e) {
            fail(e.code, std::string(what) + ": " + e.msg);
        } catch (const std::exception& e) {
            fail(APDS_ERR_INTERNAL, std::string(what) + ": " + e.what());
        }
        (void)apds_dev_timing_enable(0);
        (void)apds_thread_release();   // this worker's stream + workspace go back to the process-wide cache
    }

    void finish(Slot* s, const apds_frame_result& r, const std::string& why) {
        {
            std::lock_guard<std::mutex> g(m);
            results[s->index] = r;
            if (r.status != APDS_OK) result_errors[s->index] = why;
            done++;
        }
        cv.notify_all();
        q_free[(size_t)s->owner]->push(s);
    }

    // ---- stage 1: extraction ----------------------------------------------------------------------------------------------------------
    void extract_worker(int e) {
        hipStream_t st = st_extract[e];
        Slot* s = nullptr;
        while (q_jobs[(size_t)e]->pop(s)) {
            if (!s) {   // apds_pipeline_stats on the idle pipeline: this thread's event timers -> the totals
                harvest({"akaze_extract"});
                continue;
            }
            const void* img = s->src;
            s->status = APDS_OK;
            s->K = 0;
            if (!s->on_device) {
                // a host frame: uploaded on THIS worker's stream in front of the extraction, into the slot's own buffer; the other
                // extraction worker and the match run meanwhile, so the PCIe copy is hidden (pinned memory makes it a true async copy)
                const size_t bytes = s->stride * (size_t)p.rows;
                if (s->frame_dev_bytes < bytes) {
                    if (s->frame_dev) HIP_CHECK(hipFree(s->frame_dev));
                    s->frame_dev = nullptr;
                    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->frame_dev), bytes));
                    s->frame_dev_bytes = bytes;
                }
                HIP_CHECK(hipMemcpyAsync(s->frame_dev, s->src, bytes, hipMemcpyHostToDevice, st));
                img = s->frame_dev;
            }
            int n = 0;
            const int rc = apds_dev_akaze_extract(img, p.rows, p.cols, p.channels, s->stride, cap, s->kps, s->desc, cap, &n, st);
            if (rc != APDS_OK) {   // this frame fails, the pipeline goes on (a sharded run contributes zero queries for it)
                s->status = rc;
                s->error = apds_last_error();
                n = 0;
            }
            s->K = n;
            HIP_CHECK(hipEventRecord(s->ev_extract, st));
            if (extract_delay_s > 0) std::this_thread::sleep_for(std::chrono::duration<double>(extract_delay_s));   // test hook: a box on which extraction cannot keep up
            q_ext[(size_t)e]->push(s);
            if (world == 1) q_any.push(e);
        }
        collect({"akaze_extract"});
    }

    // ---- stage 2: match (one GPU) ------------------------------------------------------------------------------------------------------
    // Starvation watch: the match stream should never wait for a frame. On some boxes the short extraction kernels are dispatched so late
    // under the match kernel, which owns every wave slot, that extraction paces the pipeline. The gap on the match stream between one
    // frame's last match kernel and the next frame's first is measured with events; if three of six consecutive gaps exceed 4 ms (healthy:
    // 0.01 / 1.2 ms alternating) the main scan's occupancy is capped at two workgroups per CU (apds_dev_match_lds_cap), which leaves
    // wave slots for the other stages at ~1.5 % of match throughput. The cap is scoped to THIS thread's scan launches (set_thread_scan_cap):
    // other matchers of the process are not touched and nothing has to be restored.
    void match_worker() {
        bool watch = adaptive_cap;
        if (p.match_lds_cap > 0) set_thread_scan_cap(p.match_lds_cap), watch = false;   // the caller asked for a cap from the first frame on
        std::deque<std::pair<Slot*, Slot*>> pending;   // (previous frame, this frame): gap = prev.ev_mend -> this.ev_mstart
        std::vector<float> recent;
        Slot* prev = nullptr;
        int e = 0;
        int64_t seen = 0;
        while (q_any.pop(e)) {
            if (e < 0) {
                harvest({"hamming_topk", "hamming_topk_sample"});
                continue;
            }
            // One GPU: nothing requires frame order on the match stream (results are stored by frame index), so frames are matched first come,
            // first served: one of the two extraction workers tends to finish just after a match ends, and in frame order the match stream
            // then waited ~1.2 ms for every second frame
            Slot* s = nullptr;
            if (!q_ext[(size_t)e]->pop(s)) break;
            const int K = s->K;
            if (K > 0 && split) {
                HIP_CHECK(hipStreamWaitEvent(st_pre, s->ev_extract, 0));   // threshold pre-pass: beside the previous frame's main scan
                PIPE_OK(apds_dev_topk_prepass(s->topk_state, s->desc, K, db_rows, n_rows, index_base, 2, st_pre));
                HIP_CHECK(hipEventRecord(s->ev_pre, st_pre));
                HIP_CHECK(hipStreamWaitEvent(st_match, s->ev_pre, 0));
                HIP_CHECK(hipEventRecord(s->ev_mstart, st_match));
                PIPE_OK(apds_dev_topk_scan(s->topk_state, s->desc, db_rows, st_match));
                HIP_CHECK(hipEventRecord(s->ev_scan, st_match));
                HIP_CHECK(hipEventRecord(s->ev_mend, st_match));
                HIP_CHECK(hipStreamWaitEvent(st_merge, s->ev_scan, 0));    // record merge: beside the next frame's main scan
                PIPE_OK(apds_dev_topk_merge(s->topk_state, index_base, s->keys, st_merge));
                HIP_CHECK(hipEventRecord(s->ev_match, st_merge));
            } else {
                HIP_CHECK(hipStreamWaitEvent(st_match, s->ev_extract, 0));
                HIP_CHECK(hipEventRecord(s->ev_mstart, st_match));
                if (K > 0 && db_expanded) {   // the matrix-core matcher on the DB's expanded copy
                    PIPE_OK(guarded([&] {
                        ctx().ws_reset();   // (as every entry point does: the scan's scratch is this thread's workspace from its start; one stream orders the frames)
                        hamming_mfma_topk_train_device(s->desc, K, db_expanded, index_base, 2, static_cast<uint64_t*>(s->keys), st_match);
                    }));
                } else if (K > 0) {
                    PIPE_OK(apds_dev_hamming_topk(s->desc, K, db_rows, n_rows, index_base, 2, s->keys, st_match));
                }
                HIP_CHECK(hipEventRecord(s->ev_match, st_match));
                HIP_CHECK(hipEventRecord(s->ev_mend, st_match));
            }
            if (watch) {
                if (prev && seen >= 4) pending.emplace_back(prev, s);   // the first frames are the pipeline filling up
                while (!pending.empty() && hipEventQuery(pending.front().second->ev_mstart) == hipSuccess) {
                    float g = -1;
                    // (a slot that was re-used in the meantime re-recorded its events: the sample is skipped)
                    if (hipEventElapsedTime(&g, pending.front().first->ev_mend, pending.front().second->ev_mstart) != hipSuccess) (void)hipGetLastError(), g = -1;
                    pending.pop_front();
                    if (g >= 0 && g < 1000) {
                        recent.push_back(g);
                        std::lock_guard<std::mutex> lk(m);
                        gaps.push_back(g);
                    }
                }
                (void)hipGetLastError();   // hipEventQuery's hipErrorNotReady is not an error
                int late = 0;
                for (size_t i = recent.size() >= 6 ? recent.size() - 6 : 0; i < recent.size(); i++) late += recent[i] > 4.0f;
                if (recent.size() >= 6 && late >= 3 && cap_bytes == 0) {
                    set_thread_scan_cap(55000);
                    std::lock_guard<std::mutex> lk(m);
                    cap_bytes = 55000;
                    cap_frame = (int)s->index;
                    for (size_t i = recent.size() - 6; i < recent.size(); i++) cap_gaps.push_back(recent[i]);
                    watch = false;
                }
            }
            prev = s;
            seen++;
            q_homography.push(s);
        }
        collect({"hamming_topk", "hamming_topk_sample"});
        q_homography.close();
    }

    // ---- stage 2: match (row-sharded DB) -----------------------------------------------------------------------------------------------
    // This thread issues ALL collectives of the pipeline, in frame order, the same order on every rank: counts(i) [a tiny all-gather on its
    // own stream; the only host synchronisation, and frame i-1's scan is already queued behind it], gather(i) [its own stream, after that
    // frame's extraction] BEFORE exchange(i-1) [match stream, after scan(i-1)]: the query all-gather of a frame travels under the previous
    // frame's scan. Deferring exchange(i-1) until frame i has arrived is only safe when frame i is certain to arrive on EVERY rank, and the
    // choice must be the same everywhere (the collectives' order is at stake): each rank adds to its count a flag "frame i+1 is already
    // submitted here", and frame i's exchange is deferred iff every rank set it. A host that streams (submits ahead of the results) gets
    // the overlap on all frames but its last; a host that submits one frame and waits gets that frame's result without a successor.
    void sharded_match_worker() {
        constexpr int NEXT_FLAG = 1 << 30;
        Slot* prev = nullptr;
        for (int64_t i = 0;; i++) {
            Slot* s = nullptr;
            if (!q_ext[(size_t)(i % E)]->pop(s)) break;
            if (!s) {   // apds_pipeline_stats on the idle pipeline (no frame is between gather and exchange)
                harvest({"hamming_topk", "hamming_topk_sample"});
                i--;
                continue;
            }
            bool next_here;
            {
                std::lock_guard<std::mutex> g(m);
                next_here = next_submit > i + 1;
            }
            s->counts.assign((size_t)world, 0);
            PIPE_OK(apds_shard_counts(shard, s->K | (next_here ? NEXT_FLAG : 0), s->counts.data(), st_counts));
            bool next_everywhere = true;
            for (int& c : s->counts) {
                next_everywhere = next_everywhere && (c & NEXT_FLAG);
                c &= NEXT_FLAG - 1;
            }
            HIP_CHECK(hipStreamWaitEvent(st_gather, s->ev_extract, 0));
            PIPE_OK(apds_shard_gather(shard, s->shard_slot, s->K ? s->desc : nullptr, s->K, s->counts.data(), st_gather));
            auto finish_exchange = [&](Slot* f) {
                PIPE_OK(apds_shard_exchange_merge(shard, f->shard_slot, 2, f->keys, st_match));
                HIP_CHECK(hipEventRecord(f->ev_match, st_match));
                q_homography.push(f);
            };
            if (prev) finish_exchange(prev);
            prev = nullptr;
            PIPE_OK(apds_shard_scan(shard, s->shard_slot, 2, st_match));
            if (next_everywhere) prev = s;   // frame i+1 is on its way on every rank: its gather goes out first
            else finish_exchange(s);
        }
        collect({"hamming_topk", "hamming_topk_sample"});
        q_homography.close();
    }

    // ---- stage 3: ratio filter, matched points, homography ---------------------------------------------------------------------------
    void homography_worker() {
        hipStream_t st = st_homography;
        Slot* s = nullptr;
        int* count_dev = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&count_dev), 256));
        while (q_homography.pop(s)) {
            if (!s) {
                harvest({"ransac_score"});
                continue;
            }
            apds_frame_result r{};
            r.frame = s->index;
            r.status = s->status;
            r.n_keypoints = s->K;
            std::string why = s->status != APDS_OK ? s->error : std::string();
            try {
                HIP_CHECK(hipStreamWaitEvent(st, s->ev_match, 0));
                int M = 0;
                if (s->K > 0) PIPE_OK(apds_dev_ratio_filter(s->keys, s->K, 2, p.filter_strength, s->matches, &M, st));
                r.n_matches = M;
                if (M >= 4) {
                    // query_idx -> this frame's keypoints, train_idx -> the DB rows' keypoints (the intended gather: lib.rs:161-180 has two bugs)
                    PIPE_OK(apds_dev_points_from_matches(s->kps, s->K, db_kps, (int)n_db_total, s->matches, M, 0, s->p1, s->p2, st));
                    const int rc = apds_dev_find_homography(s->p1, s->p2, M, p.homography_method, p.reproj_threshold, p.max_iters, p.confidence, r.H, s->mask, st);
                    if (rc == APDS_OK) {
                        r.homography_found = 1;
                        r.n_inliers = count_nonzero_device(s->mask, M, count_dev, st);
                    } else if (rc != APDS_ERR_EMPTY) {
                        stage_fail(rc);
                    }
                }
            } catch (const StageError& e) {   // this frame's failure: reported with the frame, the pipeline goes on
                r.status = e.code;
                why = e.msg;
            }
            finish(s, r, why);
        }
        (void)hipFree(count_dev);
        collect({"ransac_score"});
    }
};

void destroy_pipeline(Pipeline* P) {
    {
        std::lock_guard<std::mutex> g(P->m);
        P->closing = true;
    }
    // the stages drain in order: no more jobs -> extraction ends -> match ends (closes the homography queue) -> homography ends
    for (auto& q : P->q_jobs) q->close();
    for (auto& q : P->q_free) q->close();
    for (size_t e = 0; e < P->threads.size() && e < (size_t)P->E; e++)
        if (P->threads[e].joinable()) P->threads[e].join();
    for (auto& q : P->q_ext) q->close();
    P->q_any.close();
    for (auto& t : P->threads)
        if (t.joinable()) t.join();
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    (void)hipSetDevice(P->device);
    (void)hipDeviceSynchronize();
    for (auto& up : P->slots) {
        Slot* s = up.get();
        for (void* ptr : {(void*)s->kps, (void*)s->desc, (void*)s->keys, (void*)s->matches, (void*)s->p1, (void*)s->p2, (void*)s->mask, (void*)s->frame_dev})
            if (ptr) (void)hipFree(ptr);
        for (hipEvent_t ev : {s->ev_extract, s->ev_pre, s->ev_scan, s->ev_match, s->ev_mstart, s->ev_mend})
            if (ev) (void)hipEventDestroy(ev);
        if (s->topk_state) topk_split_destroy(s->topk_state);
        if (s->shard_slot && P->shard) (void)apds_shard_slot_destroy(P->shard, s->shard_slot);
    }
    if (P->db_expanded) hm_train_destroy(P->db_expanded);
    P->db_expanded = nullptr;
    if (!P->own_match_stream) P->st_match = nullptr;
    for (hipStream_t st : {P->st_match, P->st_pre, P->st_merge, P->st_homography, P->st_gather, P->st_counts})
        if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : P->st_extract)
        if (st) (void)hipStreamDestroy(st);
    if (previous >= 0) (void)hipSetDevice(previous);
    delete P;
}

Pipeline* pipe_handle(void* h) {
    APDS_REQUIRE(h, APDS_ERR_BAD_ARG, "null pipeline handle");
    return static_cast<Pipeline*>(h);
}

}  // namespace

extern "C" {

int apds_pipeline_create(void** pipe, const void* db_rows64_dev, int64_t n_rows, uint32_t index_base, void* shard, const void* db_kps_dev, int64_t n_db_total,
                         const apds_pipeline_params* params) {
    return guarded([&] {
        APDS_REQUIRE(pipe && params, APDS_ERR_BAD_ARG, "null argument");
        *pipe = nullptr;
        const apds_pipeline_params& in = *params;
        APDS_REQUIRE(in.rows > 0 && in.cols > 0 && (in.channels == 1 || in.channels == 3 || in.channels == 4), APDS_ERR_ASSERT, "bad frame geometry");
        APDS_REQUIRE(shard || (db_rows64_dev && n_rows >= 2), APDS_ERR_ASSERT, "a train set of at least two rows (or a shard handle) is required");
        APDS_REQUIRE(db_kps_dev && n_db_total > 0 && n_db_total < (1ll << 31), APDS_ERR_ASSERT, "the train rows' keypoints are required (n_db_total x 28 bytes)");
        ThreadCtx& c = ctx();
        std::unique_ptr<Pipeline, void (*)(Pipeline*)> P(new Pipeline(), destroy_pipeline);
        P->p = in;
        P->device = c.device;
        P->db_rows = db_rows64_dev;
        P->n_rows = n_rows;
        P->index_base = index_base;
        P->shard = shard;
        P->db_kps = static_cast<const apds_keypoint*>(db_kps_dev);
        P->n_db_total = n_db_total;
        const Config& cfg = config();
        if (shard) {
            int w = 1;
            int64_t nr = 0;
            uint32_t base = 0;
            const int rc = apds_shard_info(shard, nullptr, &w, &nr, &base, nullptr, nullptr);
            if (rc != APDS_OK) fail(rc, apds_last_error());
            P->world = w;
            P->n_rows = nr;
            P->index_base = base;
        }
        P->cap = in.max_points > 0 ? std::min(in.max_points, APDS_MAX_POINTS) : APDS_MAX_POINTS;
        P->E = std::max(1, std::min(8, in.extract_workers > 0 ? in.extract_workers : cfg.pipe_extract_workers));
        const int n_slots = std::max(in.n_slots > 0 ? in.n_slots : 6, 2 * P->E);
        // three streams for the pre-pass, main scan and merge of consecutive frames: the vector-ALU matcher's pipeline (its pre-pass and record
        // merge are 0.8 ms of small kernels per frame). The matrix-core matcher has one main launch and two tiny ones: on one stream 131.6
        // frames/s, split over three 126.8 (profiles/r04/ab_bench_env.txt); APDS_MATCH_SPLIT=2 forces the split for it as well.
        P->split = P->world == 1 && !shard && (cfg.match_mfma ? cfg.pipe_match_split == 2 : cfg.pipe_match_split != 0);
        // (the starvation watch caps hamming_topk_kernel's occupancy: the matrix-core matcher has no such knob and is not watched)
        P->adaptive_cap = cfg.pipe_adaptive_cap != 0 && P->split && in.match_lds_cap == 0 && !cfg.match_mfma;
        P->extract_delay_s = in.debug_extract_delay_ms > 0 ? in.debug_extract_delay_ms * 1e-3 : 0.0;
        if (P->p.reproj_threshold <= 0) P->p.reproj_threshold = 3.0;
        if (P->p.max_iters <= 0) P->p.max_iters = 2000;
        if (!(P->p.confidence > 0 && P->p.confidence < 1)) P->p.confidence = 0.995;
        // the short kernels of extraction / homography / query gather get high-priority queues: their blocks are dispatched as soon as match
        // workgroups retire (the match kernel alone fills every CU for ~22 ms)
        const int hp = cfg.pipe_prio;
        for (int e = 0; e < P->E; e++) HIP_CHECK(hipStreamCreateWithPriority(&P->st_extract[e], hipStreamNonBlocking, hp));
        if (in.match_stream) P->st_match = static_cast<hipStream_t>(in.match_stream), P->own_match_stream = false;
        else HIP_CHECK(hipStreamCreateWithPriority(&P->st_match, hipStreamNonBlocking, 0));
        HIP_CHECK(hipStreamCreateWithPriority(&P->st_homography, hipStreamNonBlocking, hp));
        if (P->split) {
            HIP_CHECK(hipStreamCreateWithPriority(&P->st_pre, hipStreamNonBlocking, 0));
            HIP_CHECK(hipStreamCreateWithPriority(&P->st_merge, hipStreamNonBlocking, hp));
        }
        if (shard) {
            HIP_CHECK(hipStreamCreateWithPriority(&P->st_gather, hipStreamNonBlocking, hp));
            HIP_CHECK(hipStreamCreateWithPriority(&P->st_counts, hipStreamNonBlocking, hp));
        }
        if (P->world == 1 && !shard && cfg.match_mfma) P->db_expanded = hm_train_create(P->db_rows, P->n_rows, P->st_match);
        const size_t cap = (size_t)P->cap;
        for (int i = 0; i < n_slots; i++) {
            auto s = std::make_unique<Slot>();
            s->owner = i % P->E;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->kps), cap * sizeof(apds_keypoint)));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->desc), cap * 64));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->keys), cap * 2 * 8));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->matches), cap * sizeof(apds_dmatch)));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->p1), cap * 8));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->p2), cap * 8));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->mask), cap));
            for (hipEvent_t* ev : {&s->ev_extract, &s->ev_pre, &s->ev_scan, &s->ev_match}) HIP_CHECK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
            HIP_CHECK(hipEventCreate(&s->ev_mstart));
            HIP_CHECK(hipEventCreate(&s->ev_mend));
            if (P->split) {
                s->topk_state = topk_split_create();
                if (P->db_expanded) topk_split_use_train(s->topk_state, P->db_expanded);
            }
            if (shard) {
                const int rc = apds_shard_slot_create(shard, P->cap, 2, &s->shard_slot);
                if (rc != APDS_OK) fail(rc, apds_last_error());
            }
            P->slots.push_back(std::move(s));
        }
        for (int e = 0; e < P->E; e++) {
            P->q_free.push_back(std::make_unique<Chan<Slot*>>());
            P->q_jobs.push_back(std::make_unique<Chan<Slot*>>());
            P->q_ext.push_back(std::make_unique<Chan<Slot*>>());
        }
        for (auto& s : P->slots) P->q_free[(size_t)s->owner]->push(s.get());
        HIP_CHECK(hipDeviceSynchronize());
        Pipeline* raw = P.get();
        for (int e = 0; e < P->E; e++)   // (the first E threads are the extraction workers: destroy joins them first)
            P->threads.emplace_back([raw, e] { raw->worker("extraction", [&] { raw->extract_worker(e); }); });
        if (P->world > 1 || shard) P->threads.emplace_back([raw] { raw->worker("sharded match", [&] { raw->sharded_match_worker(); }); });
        else P->threads.emplace_back([raw] { raw->worker("match", [&] { raw->match_worker(); }); });
        P->threads.emplace_back([raw] { raw->worker("homography", [&] { raw->homography_worker(); }); });
        *pipe = P.release();
    });
}

int apds_pipeline_submit(void* pipe, const void* frame, size_t stride_bytes, int on_device, int64_t* frame_id) {
    return guarded([&] {
        Pipeline* P = pipe_handle(pipe);
        APDS_REQUIRE(frame, APDS_ERR_BAD_ARG, "null frame");
        APDS_REQUIRE(stride_bytes >= (size_t)P->p.cols * (size_t)P->p.channels, APDS_ERR_ASSERT, "row stride shorter than a row");
        std::lock_guard<std::mutex> one(P->submit_m);
        int64_t index;
        {
            std::lock_guard<std::mutex> g(P->m);
            if (P->failed) fail(P->fail_code, P->fail_msg);
            index = P->next_submit;
        }
        Slot* s = nullptr;
        const size_t e = (size_t)(index % P->E);
        if (!P->q_free[e]->pop(s)) {   // blocks while every slot of this worker is in flight
            std::lock_guard<std::mutex> g(P->m);
            fail(P->failed ? P->fail_code : APDS_ERR_INTERNAL, P->failed ? P->fail_msg : "the pipeline is shutting down");
        }
        s->src = frame;
        s->stride = stride_bytes;
        s->on_device = on_device != 0;
        s->index = index;
        {
            std::lock_guard<std::mutex> g(P->m);
            P->next_submit = index + 1;
        }
        if (frame_id) *frame_id = index;
        P->q_jobs[e]->push(s);
    });
}

int apds_pipeline_poll(void* pipe, apds_frame_result* result, int wait) {
    int ready = 0;
    const int rc = guarded([&] {
        Pipeline* P = pipe_handle(pipe);
        APDS_REQUIRE(result, APDS_ERR_BAD_ARG, "null output");
        std::unique_lock<std::mutex> g(P->m);
        auto have = [&] { return P->results.count(P->next_result) > 0; };
        if (wait) P->cv.wait(g, [&] { return have() || P->failed || P->next_result >= P->next_submit; });
        if (have()) {
            *result = P->results[P->next_result];
            P->results.erase(P->next_result);
            auto why = P->result_errors.find(P->next_result);
            if (why != P->result_errors.end()) {   // (the frame's own status is in result->status; the text goes where every entry point leaves it)
                set_last_error(why->second);
                P->result_errors.erase(why);
            }
            P->next_result++;
            ready = 1;
            return;
        }
        if (P->failed) fail(P->fail_code, P->fail_msg);
    });
    return rc != APDS_OK ? rc : (ready ? APDS_OK : APDS_PIPELINE_NOT_READY);
}

int apds_pipeline_stats(void* pipe, apds_pipeline_counters* out, int reset) {
    return guarded([&] {
        Pipeline* P = pipe_handle(pipe);
        APDS_REQUIRE(out, APDS_ERR_BAD_ARG, "null output");
        std::memset(out, 0, sizeof(*out));
        if (P->p.timing) {
            // the kernel timers are HIP events held by the worker threads: once the pipeline is idle every worker is asked (a null job through
            // its own queue) to add its finished timers to the totals
            std::unique_lock<std::mutex> g(P->m);
            P->cv.wait(g, [&] { return P->done >= P->next_submit || P->failed; });
            if (!P->failed && !P->closing) {
                P->harvest_acks = 0;
                const int64_t consumed = P->next_submit;
                g.unlock();
                for (auto& q : P->q_jobs) q->push(nullptr);
                if (P->world > 1 || P->shard) P->q_ext[(size_t)(consumed % P->E)]->push(nullptr);
                else P->q_any.push(-1);
                P->q_homography.push(nullptr);
                g.lock();
                P->cv.wait(g, [&] { return P->harvest_acks >= P->E + 2 || P->failed; });
            }
        }
        std::lock_guard<std::mutex> g(P->m);
        out->frames_submitted = P->next_submit;
        out->frames_done = P->done;
        auto get = [&](const char* n, double& ms, int& k) {
            auto it = P->timers.find(n);
            if (it != P->timers.end()) ms = it->second.first, k = it->second.second;
        };
        get("hamming_topk", out->hamming_topk_ms, out->hamming_topk_launches);
        get("hamming_topk_sample", out->hamming_topk_sample_ms, out->hamming_topk_sample_launches);
        get("akaze_extract", out->akaze_extract_ms, out->akaze_extract_calls);
        get("ransac_score", out->ransac_score_ms, out->ransac_score_launches);
        out->match_gaps = (int)P->gaps.size();
        double sum = 0;
        for (size_t i = 0; i < P->gaps.size(); i++) {
            sum += P->gaps[i];
            if (i < 16) out->match_gaps_first_ms[i] = P->gaps[i];
        }
        out->match_gap_mean_ms = P->gaps.empty() ? 0.0 : sum / (double)P->gaps.size();
        for (size_t i = 0; i < P->cap_gaps.size() && i < 6; i++) out->match_lds_cap_gaps_ms[i] = P->cap_gaps[i];
        out->match_lds_cap_bytes = P->p.match_lds_cap > 0 ? P->p.match_lds_cap : P->cap_bytes;
        out->match_lds_cap_set_at_frame = P->cap_frame;
        out->extract_workers = P->E;
        out->slots = (int)P->slots.size();
        out->split_scan = P->split ? 1 : 0;
        out->world = P->world;
        if (reset) {
            P->timers.clear();
            P->gaps.clear();
        }
    });
}

int apds_pipeline_destroy(void* pipe) {
    return guarded([&] {
        if (!pipe) return;
        destroy_pipeline(static_cast<Pipeline*>(pipe));
    });
}

}  // extern "C"
