// csrc/runtime.cpp — library-level entry points and the per-thread runtime (stream, workspace, timing).
#include <chrono>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>

#include <mutex>
#include <vector>

#include <atomic>

#include "common.h"
#include "config.h"

namespace apds {

const Config& config() {
    static const Config c = [] {
        auto env = [](const char* name, int dflt) {
            const char* v = getenv(name);
            return v && *v ? atoi(v) : dflt;
        };
        Config k;
        k.nld_strip = env("APDS_NLD_STRIP", 1);
        k.sf_strip = env("APDS_SF_STRIP", 1);
        k.base_strip = env("APDS_BASE_STRIP", 1);
        k.level_strip = env("APDS_LEVEL_STRIP", 1);
        k.level_fuse = env("APDS_LEVEL_FUSE", 1);
        k.level_stream = env("APDS_LEVEL_STREAM", 1);
        k.level_stream_rows = env("APDS_LEVEL_STREAM_ROWS", 0);
        k.doh_strip = env("APDS_DOH_STRIP", 1);
        k.doh_strip_rows = env("APDS_DOH_STRIP_ROWS", 0);
        k.kp_ranked = env("APDS_KP_RANKED", 1);
        k.kp_xcd = env("APDS_KP_XCD", 1);
        k.fed_shrink = env("APDS_FED_SHRINK", 1);
        k.half_fuse = env("APDS_HALF_FUSE", 1);
        k.early_fork = env("APDS_EARLY_FORK", 0);
        k.akaze_fork = env("APDS_AKAZE_FORK", 1);
        k.side_probe = env("APDS_SIDE_PROBE", 1);
        k.event_scope = env("APDS_EVENT_SCOPE", 2);
        k.flag_fork = env("APDS_FLAG_FORK", 0);
        k.match_mfma = env("APDS_MATCH_MFMA", 1);
        k.match_mfma_xcd = env("APDS_MATCH_MFMA_XCD", 1);
        k.match_mfma_splits = env("APDS_MATCH_MFMA_SPLITS", 0);
        k.match_mfma_sample = env("APDS_MATCH_MFMA_SAMPLE", 16384);
        k.match_mfma_prio = env("APDS_MATCH_MFMA_PRIO", 2);
        k.l2_prio = env("APDS_L2_PRIO", 1);
        k.early_count = env("APDS_EARLY_COUNT", 1);
        k.match_mfma_lds_pad = env("APDS_MATCH_MFMA_LDS_PAD", 0);
        k.debug_host_time = env("APDS_DEBUG_HOST_TIME", 0);
        k.match_lds_cap = env("APDS_MATCH_LDS_CAP", 0);
        k.match_sample = env("APDS_MATCH_SAMPLE", 16384);
        k.ransac_batch = env("APDS_RANSAC_BATCH", 512);
        k.pnp_batch = env("APDS_PNP_BATCH", 2048);
        k.l2_sample_div = std::max(1, env("APDS_L2_SAMPLE_DIV", 12));
        k.pipe_extract_workers = std::max(1, env("APDS_EXTRACT_WORKERS", 2));
        k.pipe_match_split = env("APDS_MATCH_SPLIT", 1);
        k.pipe_adaptive_cap = env("APDS_ADAPTIVE_CAP", 1);
        k.pipe_prio = env("APDS_PIPE_PRIO", -1);
        k.loopback_lag_rank = -1;
        k.loopback_lag_ms = 0;
        if (const char* lag = getenv("APDS_TEST_LOOPBACK_LAG")) {
            int r = -1, ms = 0;
            if (sscanf(lag, "%d:%d", &r, &ms) == 2 && r >= 0 && ms > 0) k.loopback_lag_rank = r, k.loopback_lag_ms = ms;
        }
        return k;
    }();
    return c;
}

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}
}  // namespace
TraceRange::TraceRange(const char* name) {
    if (roctx().push) roctx().push(name);
}
TraceRange::~TraceRange() {
    if (roctx().pop) roctx().pop();
}

static thread_local std::string g_last_error;
static thread_local ThreadCtx g_ctx;

void set_last_error(const std::string& m) { g_last_error = m; }
ThreadCtx& ctx() {
    g_ctx.ensure();
    return g_ctx;
}

void ThreadCtx::ensure() {
    if (ready) return;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        fail(APDS_ERR_NO_DEVICE, "no HIP device available: libapds_hip has no CPU fallback");
    }
    if (device >= n) fail(APDS_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    ready = true;
    live_contexts().fetch_add(1);
}

// host threads that currently own a stream + workspace (created by their first call, dropped by apds_thread_release)
std::atomic<int>& live_contexts() {
    static std::atomic<int> n{0};
    return n;
}

hipStream_t ThreadCtx::side_stream() {
    if (!side) {
        // normal priority: a high-priority stream, even an idle one, halves the throughput of OTHER host threads' streams on this
        // runtime (4 threads extracting 1024^2 tiles: 1320 /s with one high-priority side stream around, 1880 /s with normal ones)
        HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, 0));
    }
    return side;
}

namespace {
std::mutex g_side_cache_mutex;
std::vector<std::pair<int, hipStream_t>> g_side_cache;
}  // namespace
bool take_cached_side_stream(int device, hipStream_t& out) {
    std::lock_guard<std::mutex> g(g_side_cache_mutex);
    for (size_t i = 0; i < g_side_cache.size(); i++)
        if (g_side_cache[i].first == device) {
            out = g_side_cache[i].second;
            g_side_cache.erase(g_side_cache.begin() + i);
            return true;
        }
    return false;
}
void cache_side_stream(int device, hipStream_t st) {
    std::lock_guard<std::mutex> g(g_side_cache_mutex);
    g_side_cache.emplace_back(device, st);
}

int* ThreadCtx::pinned_ints(size_t n) {
    if (n > host_ints_cap) {
        if (host_ints) (void)hipHostFree(host_ints);
        host_ints = nullptr;
        host_ints_cap = 0;
        const size_t cap = std::max<size_t>(64, n * 2);
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host_ints), cap * sizeof(int), hipHostMallocDefault));
        host_ints_cap = cap;
    }
    return host_ints;
}

// side streams, their events and the join event: dropped with the thread's stream (release, device change)
void ThreadCtx::drop_side() {
    if (host_ints) (void)hipHostFree(host_ints);
    host_ints = nullptr;
    host_ints_cap = 0;
    side_probe_caller = side_probe_choice = nullptr;
    for (hipStream_t& st : side_pool)
        if (st) {
            (void)hipStreamSynchronize(st);
            cache_side_stream(device, st);
            st = nullptr;
        }
    if (side) {
        (void)hipStreamSynchronize(side);
        (void)hipStreamDestroy(side);
        side = nullptr;
    }
    for (hipEvent_t e : fork_events) (void)hipEventDestroy(e);
    fork_events.clear();
    if (join_event) (void)hipEventDestroy(join_event);
    join_event = nullptr;
    fork_open = false;
    if (tail_event) (void)hipEventSynchronize(tail_event), (void)hipEventDestroy(tail_event);
    if (count_event) (void)hipEventDestroy(count_event);
    tail_event = count_event = nullptr;
    tail_pending = false;
    if (fork_flag) (void)hipFree(fork_flag);
    fork_flag = nullptr;
    fork_flag_tried = false;
    fork_seq = 0;
    fork_pending = ForkSignal{};
}

bool ThreadCtx::fork_flag_ready() {
    if (!fork_flag_tried) {
        fork_flag_tried = true;
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) == hipSuccess && p) {
            fork_flag = static_cast<unsigned*>(p);
            *reinterpret_cast<volatile unsigned long long*>(p) = 0;   // signal memory is host-visible
            fork_seq = 0;
        } else {
            (void)hipGetLastError();
        }
    }
    return fork_flag != nullptr;
}

// Events that order one GPU stream after another of the same device: no timing and no system-scope fence when the event completes.
// The default event makes the producer's data host-visible (cache write-back + invalidate in front of the next kernel: ~3 us per
// event, 40 us per 4096^2 frame with its 17 fork / join events); a stream-to-stream dependency on one device is covered by the
// agent-scope release / acquire every kernel dispatch carries (the parity tests at every tile size run with this setting: a plane
// that was not written back would be stale as a whole for the small tiles). APDS_EVENT_SCOPE=1: the runtime's default event.
unsigned stream_event_flags() {
    const int scope = config().event_scope;
    return hipEventDisableTiming | (scope == 2 ? (unsigned)hipEventDisableSystemFence : scope == 0 ? (unsigned)hipEventReleaseToDevice : 0u);
}

hipEvent_t ThreadCtx::fork_event(size_t i) {
    while (fork_events.size() <= i) {
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreateWithFlags(&e, stream_event_flags()));
        fork_events.push_back(e);
    }
    return fork_events[i];
}

// Device slabs that released threads leave behind, per device. A pipeline that starts fresh worker threads for every run would
// otherwise pay hipMalloc / hipFree (gigabytes, with an implicit device-wide synchronisation) at the start of every run; that showed
// up as sporadic stalls of 0.1 - 0.5 s inside timed regions. A new slab request is served from here first, largest slab first (the
// first threads to ask are the extraction workers, which are also the ones that grew the largest workspaces).
namespace {
struct SlabCache {
    std::mutex m;
    std::vector<std::pair<int, std::pair<char*, size_t>>> free;   // (device, (pointer, bytes))
};
SlabCache& slab_cache() {
    static SlabCache c;
    return c;
}
bool take_cached_slab(int device, size_t want, std::pair<char*, size_t>& out) {
    SlabCache& c = slab_cache();
    std::lock_guard<std::mutex> g(c.m);
    int best = -1;
    for (int i = 0; i < (int)c.free.size(); i++)
        if (c.free[i].first == device && c.free[i].second.second >= want && (best < 0 || c.free[i].second.second > c.free[best].second.second)) best = i;
    if (best < 0) return false;
    out = c.free[best].second;
    c.free.erase(c.free.begin() + best);
    return true;
}
void cache_slab(int device, std::pair<char*, size_t> s) {
    SlabCache& c = slab_cache();
    std::lock_guard<std::mutex> g(c.m);
    c.free.emplace_back(device, s);
}
}  // namespace

void* ThreadCtx::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    call_bytes += bytes;
    if (slabs.empty() || slab_used + bytes > slabs.back().second) {
        // an additional slab lives until the next ws_reset(), which replaces all of them by one that fits the whole call: it only has to
        // hold this request and the small ones that usually follow it (doubling here turned the 9 MB request behind an 8192^2 frame's
        // 5 GB arena into a 10 GB slab, and the next call's consolidation into 1.3 s of hipFree / hipMalloc)
        const size_t want = std::max(bytes + bytes / 8, size_t(1) << 24);
        std::pair<char*, size_t> got;
        if (take_cached_slab(device, want, got)) {
            slabs.push_back(got);
        } else {
            char* p = nullptr;
            HIP_CHECK(hipMalloc(&p, want));
            slabs.emplace_back(p, want);
        }
        slab_used = 0;
    }
    void* r = slabs.back().first + slab_used;
    slab_used += bytes;
    return r;
}

void ThreadCtx::mark_tail(hipStream_t s) {
    if (!tail_event) HIP_CHECK(hipEventCreateWithFlags(&tail_event, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(tail_event, s));
    tail_stream = s;
    tail_pending = true;
}

void ThreadCtx::ws_reset(hipStream_t for_stream) {
    if (tail_pending && !(for_stream && for_stream == tail_stream)) {   // (same stream: its order protects the workspace, the tail stays pending for others)
        HIP_CHECK(hipEventSynchronize(tail_event));
        tail_pending = false;
    }
    if (slabs.size() > 1) {
        // one slab for everything the last call asked for (+ 1/16), not the sum of the slabs it happened to open
        const size_t total = call_bytes + call_bytes / 16 + (size_t(1) << 20);
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipDeviceSynchronize());
        size_t big = 0;
        for (size_t i = 1; i < slabs.size(); i++)
            if (slabs[i].second > slabs[big].second) big = i;
        const std::pair<char*, size_t> keep = slabs[big];
        const bool fits = keep.second >= total;   // the usual case: the slab opened for the call's one large request
        for (size_t i = 0; i < slabs.size(); i++)
            if (!fits || i != big) (void)hipFree(slabs[i].first);
        slabs.clear();
        if (fits) {
            slabs.push_back(keep);
        } else {
            char* p = nullptr;
            HIP_CHECK(hipMalloc(&p, total));
            slabs.emplace_back(p, total);
        }
    }
    slab_used = 0;
    call_bytes = 0;
}

}  // namespace apds

using namespace apds;

extern "C" {

const char* apds_last_error(void) { return g_last_error.c_str(); }

void apds_free(void* p) { std::free(p); }

int apds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int apds_set_device(int ordinal) {
    return guarded([&] {
        APDS_REQUIRE(ordinal >= 0, APDS_ERR_BAD_ARG, "negative device ordinal");
        if (g_ctx.ready && g_ctx.device != ordinal) {
            // drop this thread's stream/workspace on the old device
            HIP_CHECK(hipSetDevice(g_ctx.device));
            (void)hipDeviceSynchronize();   // apds_dev_* callers may have used the workspace on streams of their own
            g_ctx.drop_side();
            for (auto& s : g_ctx.slabs) (void)hipFree(s.first);
            g_ctx.slabs.clear();
            g_ctx.slab_used = 0;
            (void)hipStreamDestroy(g_ctx.stream);
            g_ctx.ready = false;
            live_contexts().fetch_sub(1);
        }
        g_ctx.device = ordinal;
        g_ctx.ensure();
    });
}

const char* apds_build_info(void) { return "libapds_hip gfx950 (CDNA4) hand-written HIP kernels; -ffp-contract=off"; }

// Streams for callers that overlap stages: optional CU mask (bit i = compute unit i may run this stream's kernels)
// and priority (0 normal, -1 high). The 30 ms match grid otherwise occupies every CU and starves the short kernels of
// the other stages; masking a few CUs out of the MATCH stream keeps them free for those.
int apds_stream_create(int priority, const uint32_t* cu_mask, int cu_mask_words, void** stream) {
    return guarded([&] {
        APDS_REQUIRE(stream, APDS_ERR_BAD_ARG, "null output");
        ctx();   // device selected
        hipStream_t s = nullptr;
        if (cu_mask && cu_mask_words > 0) {
            HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)cu_mask_words, cu_mask));
        } else {
            HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
        }
        *stream = s;
    });
}

int apds_stream_destroy(void* stream) {
    return guarded([&] {
        if (stream) HIP_CHECK(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    });
}

// Device memory and copies for hosts that keep buffers resident (the apds_dev_* / apds_shard_* entry points) without binding the HIP
// runtime themselves: a Rust or C host needs nothing but this header.
int apds_dev_alloc(size_t bytes, void** ptr) {
    return guarded([&] {
        APDS_REQUIRE(ptr, APDS_ERR_BAD_ARG, "null output");
        *ptr = nullptr;
        ctx();   // the calling thread's device
        HIP_CHECK(hipMalloc(ptr, std::max<size_t>(bytes, 256)));
    });
}

int apds_dev_release(void* ptr) {
    return guarded([&] {
        if (ptr) HIP_CHECK(hipFree(ptr));
    });
}

int apds_dev_upload(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    return guarded([&] {
        APDS_REQUIRE((dst_dev && src_host) || bytes == 0, APDS_ERR_BAD_ARG, "null pointer");
        if (bytes) HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, pick_stream(stream)));
    });
}

int apds_dev_download(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
    return guarded([&] {
        APDS_REQUIRE((dst_host && src_dev) || bytes == 0, APDS_ERR_BAD_ARG, "null pointer");
        hipStream_t s = pick_stream(stream);
        if (bytes) HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int apds_stream_synchronize(void* stream) {
    return guarded([&] { HIP_CHECK(hipStreamSynchronize(pick_stream(stream))); });
}

int apds_dev_timing_enable(int on) {
    return guarded([&] { ctx().timing = on != 0; });
}

int apds_release_cached_memory(void) {
    return guarded([&] {
        SlabCache& c = slab_cache();
        std::lock_guard<std::mutex> g(c.m);
        for (auto& e : c.free) {
            (void)hipSetDevice(e.first);
            (void)hipFree(e.second.first);
        }
        c.free.clear();
    });
}

int apds_live_contexts(void) { return live_contexts().load(); }

int apds_thread_release(void) {
    return guarded([&] {
        ThreadCtx& c = g_ctx;
        if (!c.ready) return;
        HIP_CHECK(hipSetDevice(c.device));
        // apds_dev_* entry points launch kernels that use this thread's workspace on CALLER-supplied streams: wait for the whole
        // device, not only for the thread's own streams, before the slabs go to the cache where another thread may take them
        (void)hipDeviceSynchronize();
        for (auto& kv : c.events)
            for (auto& ev : kv.second) {
                (void)hipEventDestroy(ev.a);
                (void)hipEventDestroy(ev.b);
            }
        c.events.clear();
        c.drop_side();
        // the stream(s) were synchronised above: nothing uses the slabs any more. A thread that ends before its second call still
        // holds the chain of doubling slabs of its first one: hand ONE slab of the total size to the cache (the next thread would
        // otherwise start from a fragment, grow and consolidate inside somebody's timed region, with a device-wide sync).
        if (c.slabs.size() > 1) {
            size_t total = 0;
            for (auto& s : c.slabs) total += s.second;
            for (auto& s : c.slabs) (void)hipFree(s.first);
            c.slabs.clear();
            char* p = nullptr;
            if (hipMalloc(&p, total) == hipSuccess) c.slabs.emplace_back(p, total);
            else (void)hipGetLastError();
        }
        for (auto& s : c.slabs) cache_slab(c.device, s);
        c.slabs.clear();
        c.slab_used = 0;
        (void)hipStreamDestroy(c.stream);
        c.stream = nullptr;
        c.ready = false;
        live_contexts().fetch_sub(1);
    });
}

int apds_dev_last_kernel_ms(const char* which, float* ms, int* launches) {
    return guarded([&] {
        APDS_REQUIRE(which && ms, APDS_ERR_BAD_ARG, "null argument");
        ThreadCtx& c = ctx();
        auto it = c.events.find(which);
        float total = 0;
        int n = 0;
        if (it != c.events.end()) {
            for (auto& ev : it->second) {
                HIP_CHECK(hipEventSynchronize(ev.b));
                float t = 0;
                HIP_CHECK(hipEventElapsedTime(&t, ev.a, ev.b));
                total += t;
                n++;
                (void)hipEventDestroy(ev.a);
                (void)hipEventDestroy(ev.b);
            }
            c.events.erase(it);
        }
        *ms = total;
        if (launches) *launches = n;
    });
}

}  // extern "C"
