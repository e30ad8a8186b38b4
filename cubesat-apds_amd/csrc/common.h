// csrc/common.h — host-side plumbing shared by the entry points: error mapping, the per-thread HIP
// stream + grow-only device workspace, and hipEvent kernel timing for bench.py.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/apds.h"

// The match kernel keeps every SIMD's issue slots booked for milliseconds per frame and the hardware arbitrates issue by priority, then
// age, so co-resident waves of the short, memory-bound kernels of the other stages (running on other streams) would starve. They raise
// their own wave priority; they need few issue slots, so the match loses almost nothing and the stages overlap. Level 1 since the matrix-core
// matcher: above its ranking code (0), below its MFMA bursts (2) - 149.0 frames/s against 147.3 at level 3 or 0 (round 4, three runs each).
#ifndef APDS_STAGE_WAVE_PRIO
#define APDS_STAGE_WAVE_PRIO 1
#endif
#define APDS_RAISE_WAVE_PRIORITY() __builtin_amdgcn_s_setprio(APDS_STAGE_WAVE_PRIO)
// First act of a main-chain kernel when a fork is armed (ForkSignal below): its first block reports that the kernel has STARTED, i.e. that
// everything in front of it on its stream is done and written back. The side stream's hipStreamWaitValue32 waits for that value.
#define APDS_FORK_SIGNAL(sig)                                                                                              \
    do {                                                                                                                   \
        if ((sig).flag && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0)                                       \
            __hip_atomic_store((sig).flag, (sig).value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                      \
    } while (0)

namespace apds {

struct Error {
    int code;
    std::string msg;
};

void set_last_error(const std::string& m);

[[noreturn]] inline void fail(int code, const std::string& m) { throw Error{code, m}; }

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e == hipSuccess) return;
    int code = APDS_ERR_INTERNAL;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu || e == hipErrorInsufficientDriver ||
        e == hipErrorNotInitialized)
        code = APDS_ERR_NO_DEVICE;
    if (e == hipErrorOutOfMemory) code = APDS_ERR_NOMEM;
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    (void)hipGetLastError();
    throw Error{code, buf};
}
#define HIP_CHECK(x) ::apds::hip_check((x), #x, __FILE__, __LINE__)
#define APDS_REQUIRE(cond, code, msg) \
    do {                               \
        if (!(cond)) ::apds::fail((code), (msg)); \
    } while (0)

// "my predecessor on this stream is done": what a kernel of the main chain stores as its first act when a fork is armed
struct ForkSignal {
    unsigned* flag = nullptr;
    unsigned value = 0;
};
// Per host thread: device ordinal, a private stream, a bump-allocated workspace that only grows.
struct ThreadCtx {
    int device = 0;
    bool ready = false;
    hipStream_t stream = nullptr;
    std::vector<std::pair<char*, size_t>> slabs;
    size_t slab_used = 0;   // in slabs.back()
    size_t call_bytes = 0;  // bytes handed out since the last ws_reset() (what one slab has to hold for the call to need no second one)
    bool timing = false;
    struct Ev {
        hipEvent_t a, b;
        hipStream_t s;
    };
    std::map<std::string, std::vector<Ev>> events;
    // fork / join inside one call: a second stream for work that hangs off the main chain (akaze: the per-level Hessian kernels)
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork_events;
    hipEvent_t join_event = nullptr;
    int akaze_first_batch[2] = {8, 8};          // suppression rounds to launch before the first host check, per phase (adaptive)
    int akaze_batch_streak[2] = {0, 0};
    // fork without an event (round 4): the first block of the NEXT kernel on the main stream stores a sequence number to `fork_flag`
    // (signal memory), the side stream waits for that value (hipStreamWaitValue32). A launcher that supports it takes the armed signal.
    unsigned* fork_flag = nullptr;              // 8 bytes of signal memory, allocated on first use; null if the runtime refused
    bool fork_flag_tried = false;
    unsigned fork_seq = 0;
    ForkSignal fork_pending{};                  // armed and not yet given to a kernel
    bool fork_flag_ready();                     // allocate on first use; false -> use events
    ForkSignal arm_fork_signal() { return fork_pending = ForkSignal{fork_flag, ++fork_seq}; }
    ForkSignal take_fork_signal() {
        const ForkSignal r = fork_pending;
        fork_pending = ForkSignal{};
        return r;
    }
    bool fork_open = false;                     // side-stream work was issued and not yet joined (only after an error in between)
    int akaze_kp_estimate = 0;                  // keypoints of this thread's previous image (grid size of the per-keypoint kernels)
    int* host_ints = nullptr;                   // pinned host memory for count read-backs (a pageable target makes the copy a staged, blocking one)
    size_t host_ints_cap = 0;
    int* pinned_ints(size_t n);                 // valid until the next call with a larger n
    hipStream_t side_stream();                  // created on first use
    // a side stream that really runs beside `caller` (see side_stream_beside in misc.hip); cached per caller stream
    hipStream_t side_pool[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t side_probe_caller = nullptr, side_probe_choice = nullptr;
    void drop_side();
    hipEvent_t fork_event(size_t i);            // i-th reusable event (no timing)

    // A call that hands its result count to the host before its last kernels are done (akaze: the count is final ~0.3 ms before the
    // descriptors) leaves a TAIL on its stream: the workspace it used is still being read. The next call of this thread re-uses that memory,
    // so ws_reset() waits for the tail first - unless the next call goes to the same stream, whose order already protects it.
    hipEvent_t tail_event = nullptr, count_event = nullptr;
    hipStream_t tail_stream = nullptr;
    bool tail_pending = false;
    void mark_tail(hipStream_t s);   // everything queued on s so far is this thread's tail

    void ensure();
    void* alloc(size_t bytes);   // valid until the next ws_reset()
    void ws_reset(hipStream_t for_stream = nullptr);   // frees all but one slab sized to the high-water mark; for_stream: the stream the coming call uses, if known
    template <class T>
    T* alloc_n(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
};
ThreadCtx& ctx();
std::atomic<int>& live_contexts();   // host threads that currently own a stream + workspace

// A stream of the calling thread whose kernels run CONCURRENTLY with kernels on `caller`. The runtime maps streams onto a few
// hardware queues in creation order; two streams on one queue serialise, and which queue the caller's stream sits on cannot be
// asked. So: four candidates (consecutive creations: different queues), each timed once with a spinning one-wave kernel on it and one
// on `caller`; the candidate whose pair finishes first does not share the caller's queue. ~0.3 ms, once per (thread, caller stream).
hipStream_t side_stream_beside(hipStream_t caller);
// a one-thread kernel that stores an armed fork signal no chain kernel has taken (the launchers that cannot carry one)
void launch_fork_signal(ForkSignal sig, hipStream_t s);
// the candidate streams of released threads, per device (creating and destroying four streams per short-lived thread costs milliseconds)
bool take_cached_side_stream(int device, hipStream_t& out);
void cache_side_stream(int device, hipStream_t st);
unsigned stream_event_flags();   // flags for events that only order GPU streams of one device

inline hipStream_t pick_stream(void* s) { return s ? static_cast<hipStream_t>(s) : ctx().stream; }

// RAII: time one named kernel with hipEvents on the stream it is launched on (only when enabled).
struct KernelTimer {
    ThreadCtx::Ev ev{};
    bool on;
    const char* name;
    KernelTimer(const char* nm, hipStream_t s) : on(ctx().timing), name(nm) {
        if (!on) return;
        ev.s = s;
        HIP_CHECK(hipEventCreate(&ev.a));
        HIP_CHECK(hipEventCreate(&ev.b));
        HIP_CHECK(hipEventRecord(ev.a, s));
    }
    ~KernelTimer() {
        if (!on) return;
        (void)hipEventRecord(ev.b, ev.s);
        ctx().events[name].push_back(ev);
    }
};

// A named ROCTx range around a stage of the path (visible in `rocprofv3 --marker-trace` timelines; SURVEY section 5 asks for them). The
// marker library (librocprofiler-sdk-roctx.so, or the older libroctx64.so) is looked up once with dlopen: the library does not depend on
// a profiler being installed, and without one a range is a null-pointer test.
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};
#define APDS_RANGE(name) ::apds::TraceRange apds_trace_range_(name)

// Wrap an entry-point body: map exceptions to status codes + last_error text.
template <class F>
int guarded(F&& f) {
    try {
        f();
        return APDS_OK;
    } catch (const Error& e) {
        set_last_error(e.msg);
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return APDS_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return APDS_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown failure");
        return APDS_ERR_INTERNAL;
    }
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace apds
