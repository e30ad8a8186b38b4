// csrc/shard.cpp — the row-sharded Hamming matcher behind the C ABI (apds_shard_*, include/apds.h): shard_core.h's choreography on
// HIP buffers and kernels, with three transports for its one exchange step:
//   RCCL      one rank per GPU (process or thread), ncclAllGather for the query rows, grouped ncclSend / ncclRecv for the keys, over xGMI;
//   loopback  threads of ONE process as ranks (any number per GPU): pointer + event exchange through a hub, device-to-device copies;
//             runs the identical choreography on a one-GPU box (tests/cpp/shard_loopback_test.cpp);
//   host      the host program's own communicator as two callbacks on host buffers (gloo / MPI); device data is staged around them.
// Reference: there is no multi-device code upstream; what is sharded is the train set that keypointdb.rs:50-90 returns, and the only
// parallel caller is preprocessor/src/main.rs:86-89,227-245.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <thread>

#include "config.h"
#include "kernels.h"
#include "shard_core.h"

using namespace apds;
using namespace apds::shard;

namespace {

#define NCCL_CHECK(x)                                                                                                                   \
    do {                                                                                                                                \
        ncclResult_t r_ = (x);                                                                                                          \
        if (r_ != ncclSuccess && r_ != ncclInProgress) throw ShardError(APDS_ERR_INTERNAL, std::string(#x " failed: ") + ncclGetErrorString(r_)); \
    } while (0)

// hipSetDevice(handle's device) for the length of a destroy path, then the caller's device again: a thread bound to GPU A may destroy a
// handle living on GPU B (a thread-per-GPU host; a finaliser running on an arbitrary thread) and must keep launching on A afterwards.
struct DeviceGuard {
    int previous = -1;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&previous) != hipSuccess) previous = -1;
        (void)hipSetDevice(device);
    }
    ~DeviceGuard() {
        if (previous >= 0) (void)hipSetDevice(previous);
    }
};

struct HipDevice final : Device {
    static hipStream_t st(void* s) { return pick_stream(s); }
    void* alloc(size_t bytes) override {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
        return p;
    }
    void release(void* p) override { (void)hipFree(p); }   // (synchronises the device: nothing in flight uses p afterwards)
    void copy(void* dst, const void* src, size_t bytes, void* s) override { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st(s))); }
    void to_host(void* host, const void* dev, size_t bytes, void* s) override {
        HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st(s)));
        HIP_CHECK(hipStreamSynchronize(st(s)));
    }
    void from_host(void* dev, const void* host, size_t bytes, void* s) override {
        HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st(s)));
        HIP_CHECK(hipStreamSynchronize(st(s)));
    }
    // the shard's rows as matrix-core operands: made at the first scan and kept (the shard is resident for the handle's life: shard_core.h)
    std::mutex expanded_m;
    void* expanded = nullptr;
    const void* expanded_rows = nullptr;
    int64_t expanded_n = 0;
    ~HipDevice() override {
        if (expanded) hm_train_destroy(expanded);
    }
    void topk(const void* q, int nq, const void* rows, int64_t n_rows, uint32_t base, int k, void* out, void* s) override {
        // the scan's scratch is the calling thread's workspace, from its start: scans issued by one thread must be ordered on the GPU
        // (one stream, or events), exactly as for apds_dev_hamming_topk
        ctx().ws_reset();
        if (k <= 2 && config().match_mfma && nq > 0 && n_rows > 0) {
            void* train = nullptr;
            {
                std::lock_guard<std::mutex> g(expanded_m);
                if (expanded && (expanded_rows != rows || expanded_n != n_rows)) {
                    hm_train_destroy(expanded);
                    expanded = nullptr;
                }
                if (!expanded) {
                    expanded = hm_train_create(rows, n_rows, st(s));
                    expanded_rows = rows;
                    expanded_n = n_rows;
                }
                train = expanded;
            }
            hamming_mfma_topk_train_device(q, nq, train, base, k, static_cast<uint64_t*>(out), st(s));
            return;
        }
        hamming_topk_device(q, nq, rows, n_rows, base, k, static_cast<uint64_t*>(out), st(s));
    }
    void merge(const void* parts, int nparts, int nq, int k, void* out, void* s) override {
        merge_topk_device(static_cast<const uint64_t*>(parts), nparts, nq, k, static_cast<uint64_t*>(out), st(s));
    }
    void* event_create() override {
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreateWithFlags(&e, stream_event_flags()));
        return e;
    }
    void event_destroy(void* ev) override { (void)hipEventDestroy(static_cast<hipEvent_t>(ev)); }
    void event_record(void* ev, void* s) override { HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(ev), st(s))); }
    void stream_wait(void* s, void* ev) override { HIP_CHECK(hipStreamWaitEvent(st(s), static_cast<hipEvent_t>(ev), 0)); }
};

// ---- RCCL ------------------------------------------------------------------------------------------------------------------------------
struct RcclTransport final : Transport {
    ncclComm_t comm = nullptr;
    int* counts_dev = nullptr;
    bool broken = false;   // a collective failed half-way: the communicator's state is unknown, every later call refuses
    // ncclGroupEnd on every path out of a group, and the communicator marked broken when a call inside it failed
    struct Group {
        RcclTransport& t;
        bool open = false, ok = false;
        explicit Group(RcclTransport& t_) : t(t_) {
            NCCL_CHECK(ncclGroupStart());
            open = true;
        }
        void end() {
            open = false;
            NCCL_CHECK(ncclGroupEnd());
            ok = true;
        }
        ~Group() {
            if (open) (void)ncclGroupEnd();
            if (!ok) t.broken = true;
        }
    };
    void usable() const {
        if (broken) throw ShardError(APDS_ERR_INTERNAL, "the RCCL communicator is broken (an earlier collective failed): destroy the shard and create a new one");
    }
    RcclTransport(int rank_, int world_, const apds_comm_id& id) {
        rank = rank_;
        world = world_;
        ncclUniqueId uid;
        static_assert(sizeof(uid) == APDS_COMM_ID_BYTES, "apds_comm_id must hold an ncclUniqueId");
        std::memcpy(&uid, id.bytes, sizeof(uid));
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&counts_dev), (size_t)(world + 1) * sizeof(int)));
        try {
            NCCL_CHECK(ncclCommInitRank(&comm, world, uid, rank));
        } catch (...) {   // (a constructor that throws runs no destructor)
            (void)hipFree(counts_dev);
            throw;
        }
    }
    ~RcclTransport() override {
        if (counts_dev) (void)hipFree(counts_dev);
        if (comm) (void)ncclCommDestroy(comm);
    }
    const char* name() const override { return "rccl"; }
    void counts(int mine, int* all, void* s) override {
        usable();
        hipStream_t st = pick_stream(s);
        HIP_CHECK(hipMemcpyAsync(counts_dev + world, &mine, sizeof(int), hipMemcpyHostToDevice, st));
        collective_call([&] { NCCL_CHECK(ncclAllGather(counts_dev + world, counts_dev, 1, ncclInt32, comm, st)); });
        HIP_CHECK(hipMemcpyAsync(all, counts_dev, (size_t)world * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    template <class F>
    void collective_call(F&& f) {
        try {
            f();
        } catch (...) {
            broken = true;
            throw;
        }
    }
    void all_gather(const void* send, void* recv, size_t bytes, void* s) override {
        usable();
        collective_call([&] { NCCL_CHECK(ncclAllGather(send, recv, bytes, ncclUint8, comm, pick_stream(s))); });
    }
    void broadcast(void* buf, size_t bytes, int root, void*, Device&, void* s) override {
        usable();
        collective_call([&] { NCCL_CHECK(ncclBroadcast(buf, buf, bytes, ncclUint8, root, comm, pick_stream(s))); });
    }
    void all_to_all(const void* send, const size_t* soff, const size_t* sbytes, void* recv, const size_t* roff, const size_t* rbytes, void* s) override {
        usable();
        hipStream_t st = pick_stream(s);
        const char* sp = static_cast<const char*>(send);
        char* rp = static_cast<char*>(recv);
        // this rank's own block never touches the network
        if (sbytes[rank]) HIP_CHECK(hipMemcpyAsync(rp + roff[rank], sp + soff[rank], sbytes[rank], hipMemcpyDeviceToDevice, st));
        Group g(*this);   // (its destructor closes the group and marks the communicator broken if a call below throws)
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            if (sbytes[p]) NCCL_CHECK(ncclSend(sp + soff[p], sbytes[p], ncclUint8, p, comm, st));
            if (rbytes[p]) NCCL_CHECK(ncclRecv(rp + roff[p], rbytes[p], ncclUint8, p, comm, st));
        }
        g.end();
    }
};

// ---- the host program's communicator on device buffers (e.g. torch.distributed over its own RCCL communicator) ---------------------------------
struct DeviceCallbackTransport final : Transport {
    apds_device_transport cb;
    int* counts_dev = nullptr;
    DeviceCallbackTransport(const apds_device_transport& c, int rank_, int world_) : cb(c) {
        rank = rank_;
        world = world_;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&counts_dev), (size_t)(world + 1) * sizeof(int)));
    }
    ~DeviceCallbackTransport() override {
        if (counts_dev) (void)hipFree(counts_dev);
    }
    const char* name() const override { return "device-callbacks"; }
    void counts(int mine, int* all, void* s) override {
        hipStream_t st = pick_stream(s);
        HIP_CHECK(hipMemcpyAsync(counts_dev + world, &mine, sizeof(int), hipMemcpyHostToDevice, st));
        if (cb.all_gather(cb.user, counts_dev + world, counts_dev, sizeof(int), st)) throw ShardError(APDS_ERR_INTERNAL, "device all_gather callback failed");
        HIP_CHECK(hipMemcpyAsync(all, counts_dev, (size_t)world * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    void all_gather(const void* send, void* recv, size_t bytes, void* s) override {
        if (cb.all_gather(cb.user, send, recv, bytes, pick_stream(s))) throw ShardError(APDS_ERR_INTERNAL, "device all_gather callback failed");
    }
    void all_to_all(const void* send, const size_t* soff, const size_t* sbytes, void* recv, const size_t* roff, const size_t* rbytes, void* s) override {
        if (cb.all_to_all(cb.user, send, soff, sbytes, recv, roff, rbytes, pick_stream(s))) throw ShardError(APDS_ERR_INTERNAL, "device all_to_all callback failed");
    }
};

// ---- loopback: the ranks are threads of this process -------------------------------------------------------------------------------------
struct Hub {
    const int world;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, attached = 0;
    uint64_t generation = 0;
    bool broken = false;
    struct Post {
        const void* send = nullptr;
        std::vector<size_t> soff, sbytes;
        hipEvent_t ready = nullptr, done = nullptr;
        int count = 0;
        bool owned = false;   // a transport is attached as this rank
    };
    std::vector<Post> posts;
    explicit Hub(int w) : world(w), posts((size_t)w) {}
    // The events belong to the hub, not to the rank that created them: a rank that has left its last collective may destroy its transport
    // while a slower peer is still issuing that collective's closing hipStreamWaitEvent on this rank's `done` (found by the randomised
    // run of tests/cpp/shard_loopback_test.cpp: "invalid resource handle" once in ~900 worlds). They go when the last rank has detached.
    ~Hub() {
        for (Post& p : posts) {
            if (p.ready) (void)hipEventDestroy(p.ready);
            if (p.done) (void)hipEventDestroy(p.done);
        }
    }
    // all `world` threads meet here; a rank that never arrives (it failed) must not hang the others for ever
    void barrier() {
        std::unique_lock<std::mutex> g(m);
        if (broken) throw ShardError(APDS_ERR_INTERNAL, "loopback communicator is broken (a rank failed or timed out)");
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return;
        }
        if (!cv.wait_for(g, std::chrono::seconds(120), [&] { return generation != gen || broken; })) {
            broken = true;
            cv.notify_all();
            throw ShardError(APDS_ERR_INTERNAL, "loopback barrier timed out: not every rank issued the collective");
        }
        if (generation == gen) throw ShardError(APDS_ERR_INTERNAL, "loopback communicator is broken (a rank failed or timed out)");
    }
};
std::mutex g_hubs_mutex;
std::map<std::string, std::shared_ptr<Hub>> g_hubs;

struct LoopbackTransport final : Transport {
    std::shared_ptr<Hub> hub;
    std::string key;
    LoopbackTransport(int rank_, int world_, const apds_comm_id& id) : key(id.bytes, sizeof(id.bytes)) {
        rank = rank_;
        world = world_;
        {
            std::lock_guard<std::mutex> g(g_hubs_mutex);
            auto& h = g_hubs[key];
            if (!h) h = std::make_shared<Hub>(world);
            if (h->world != world) throw ShardError(APDS_ERR_BAD_ARG, "loopback communicator id already used with another world size");
            // a second attachment of a rank is refused BEFORE anything of the hub is touched: the post, its events and the attachment
            // count belong to the first, healthy attachment
            if (h->posts[(size_t)rank].owned) throw ShardError(APDS_ERR_BAD_ARG, "loopback rank attached twice");
            hub = h;
            hub->posts[(size_t)rank].owned = true;
            hub->attached++;
        }
        bool made_events = false;   // only events THIS constructor created are taken back when it fails
        try {
            Hub::Post& me = hub->posts[(size_t)rank];
            if (!me.ready) {        // (a rank slot that was attached, detached and is attached again keeps its hub-owned events)
                made_events = true;
                HIP_CHECK(hipEventCreateWithFlags(&me.ready, hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&me.done, hipEventDisableTiming));
            }
            hub->barrier();   // every rank's events exist before the first collective reads them
        } catch (...) {       // (a constructor that throws runs no destructor: leave the hub as it was found)
            Hub::Post& me = hub->posts[(size_t)rank];
            if (made_events) {
                if (me.ready) (void)hipEventDestroy(me.ready);   // no collective has used them yet
                if (me.done) (void)hipEventDestroy(me.done);
                me.ready = me.done = nullptr;
            }
            std::lock_guard<std::mutex> g(g_hubs_mutex);
            me.owned = false;
            if (--hub->attached == 0) g_hubs.erase(key);
            throw;
        }
    }
    ~LoopbackTransport() override {
        std::lock_guard<std::mutex> g(g_hubs_mutex);
        hub->posts[(size_t)rank].owned = false;
        if (--hub->attached == 0) g_hubs.erase(key);   // (the hub itself, and with it every rank's events, lives until the last transport lets go of it)
    }
    const char* name() const override { return "loopback"; }
    void counts(int mine, int* all, void*) override {
        hub->posts[(size_t)rank].count = mine;
        hub->barrier();
        for (int p = 0; p < world; p++) all[p] = hub->posts[(size_t)p].count;
        hub->barrier();
    }
    // post my send buffer, pull from every peer on MY stream behind their `ready` events, then hold my later work behind their `done`
    template <class Pull>
    void collective(const void* send, const size_t* soff, const size_t* sbytes, hipStream_t st, Pull pull) {
        Hub::Post& me = hub->posts[(size_t)rank];
        me.send = send;
        if (soff) me.soff.assign(soff, soff + world), me.sbytes.assign(sbytes, sbytes + world);
        HIP_CHECK(hipEventRecord(me.ready, st));
        hub->barrier();
        for (int p = 0; p < world; p++) {
            const Hub::Post& peer = hub->posts[(size_t)p];
            if (p != rank) HIP_CHECK(hipStreamWaitEvent(st, peer.ready, 0));
            pull(p, peer);
        }
        HIP_CHECK(hipEventRecord(me.done, st));
        hub->barrier();
        // test hook (APDS_TEST_LOOPBACK_LAG="rank:ms"): this rank dawdles between the closing barrier and its closing waits, the window in
        // which a faster peer may already be destroying its shard (the event-lifetime race of round 3, reproduced deterministically by
        // tests/cpp/shard_loopback_test.cpp `lag`)
        if (config().loopback_lag_ms > 0 && config().loopback_lag_rank == rank) std::this_thread::sleep_for(std::chrono::milliseconds(config().loopback_lag_ms));
        for (int p = 0; p < world; p++)
            if (p != rank) HIP_CHECK(hipStreamWaitEvent(st, hub->posts[(size_t)p].done, 0));   // my send buffer is free again only after every peer has read it
    }
    void all_gather(const void* send, void* recv, size_t bytes, void* s) override {
        hipStream_t st = pick_stream(s);
        collective(send, nullptr, nullptr, st, [&](int p, const Hub::Post& peer) {
            HIP_CHECK(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * bytes, peer.send, bytes, hipMemcpyDeviceToDevice, st));
        });
    }
    void broadcast(void* buf, size_t bytes, int root, void*, Device&, void* s) override {
        hipStream_t st = pick_stream(s);
        collective(buf, nullptr, nullptr, st, [&](int p, const Hub::Post& peer) {
            if (p == root && p != rank && bytes) HIP_CHECK(hipMemcpyAsync(buf, peer.send, bytes, hipMemcpyDeviceToDevice, st));
        });
    }
    void all_to_all(const void* send, const size_t* soff, const size_t* sbytes, void* recv, const size_t* roff, const size_t* rbytes, void* s) override {
        hipStream_t st = pick_stream(s);
        collective(send, soff, sbytes, st, [&](int p, const Hub::Post& peer) {
            if (peer.sbytes[(size_t)rank] != rbytes[p]) throw ShardError(APDS_ERR_INTERNAL, "loopback all_to_all: send / receive sizes disagree");
            if (rbytes[p])
                HIP_CHECK(hipMemcpyAsync(static_cast<char*>(recv) + roff[p], static_cast<const char*>(peer.send) + peer.soff[(size_t)rank], rbytes[p],
                                         hipMemcpyDeviceToDevice, st));
        });
    }
};

struct ShardHandle {
    HipDevice dev;
    std::unique_ptr<Transport> tr;
    std::unique_ptr<Matcher> m;
    int device = 0, transport = 0;
};

template <class F>
int shard_guarded(F&& f) {
    return guarded([&] {
        try {
            f();
        } catch (const ShardError& e) {
            throw Error{e.code, e.what()};
        }
    });
}

ShardHandle* handle(void* p) {
    APDS_REQUIRE(p, APDS_ERR_BAD_ARG, "null shard handle");
    return static_cast<ShardHandle*>(p);
}

}  // namespace

extern "C" {

int apds_comm_id_create(int transport, apds_comm_id* id) {
    return shard_guarded([&] {
        APDS_REQUIRE(id, APDS_ERR_BAD_ARG, "null output");
        std::memset(id->bytes, 0, sizeof(id->bytes));
        if (transport == APDS_TRANSPORT_RCCL) {
            ncclUniqueId uid;
            NCCL_CHECK(ncclGetUniqueId(&uid));
            std::memcpy(id->bytes, &uid, sizeof(uid));
        } else if (transport == APDS_TRANSPORT_LOOPBACK) {
            static std::atomic<uint64_t> serial{0};
            std::random_device rd;
            snprintf(id->bytes, sizeof(id->bytes), "apds-loopback-%llu-%08x%08x", (unsigned long long)serial.fetch_add(1), (unsigned)rd(), (unsigned)rd());
        } else {
            APDS_REQUIRE(transport == APDS_TRANSPORT_HOST, APDS_ERR_BAD_ARG, "unknown transport");
        }
    });
}

int apds_shard_create(void** shard, int rank, int world, int transport, const apds_comm_id* id, const apds_host_transport* host, const void* rows64_dev,
                      int64_t n_rows, uint32_t index_base) {
    return shard_guarded([&] {
        APDS_REQUIRE(shard, APDS_ERR_BAD_ARG, "null output");
        *shard = nullptr;
        APDS_REQUIRE(world >= 1 && rank >= 0 && rank < world, APDS_ERR_BAD_ARG, "rank outside [0, world)");
        APDS_REQUIRE(n_rows >= 0 && (rows64_dev || n_rows == 0), APDS_ERR_ASSERT, "bad shard rows");
        APDS_REQUIRE((uint64_t)index_base + (uint64_t)n_rows <= (1ull << 32), APDS_ERR_OUT_OF_RANGE, "global row indices must fit 32 bits");
        ThreadCtx& c = ctx();   // selects the calling thread's device: the communicator and every buffer of the shard live there
        auto h = std::make_unique<ShardHandle>();
        h->device = c.device;
        h->transport = transport;
        if (world == 1 && transport != APDS_TRANSPORT_RCCL && transport != APDS_TRANSPORT_DEVICE) {
            struct Solo final : Transport {
                const char* name() const override { return "none (one rank)"; }
                void counts(int mine, int* all, void*) override { all[0] = mine; }
                void all_gather(const void*, void*, size_t, void*) override {}
                void all_to_all(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, void*) override {}
            };
            h->tr = std::make_unique<Solo>();
        } else if (transport == APDS_TRANSPORT_RCCL) {
            APDS_REQUIRE(id, APDS_ERR_BAD_ARG, "the RCCL transport needs the communicator id rank 0 created (apds_comm_id_create)");
            h->tr = std::make_unique<RcclTransport>(rank, world, *id);
        } else if (transport == APDS_TRANSPORT_LOOPBACK) {
            APDS_REQUIRE(id, APDS_ERR_BAD_ARG, "the loopback transport needs a communicator id shared by the rank threads");
            h->tr = std::make_unique<LoopbackTransport>(rank, world, *id);
        } else if (transport == APDS_TRANSPORT_HOST) {
            APDS_REQUIRE(host && host->all_gather && host->all_to_all, APDS_ERR_BAD_ARG, "the host transport needs both callbacks");
            h->tr = std::make_unique<HostTransport>(h->dev, HostCallbacks{host->user, host->all_gather, host->all_to_all}, rank, world);
        } else if (transport == APDS_TRANSPORT_DEVICE) {
            const apds_device_transport* dt = reinterpret_cast<const apds_device_transport*>(host);
            APDS_REQUIRE(dt && dt->all_gather && dt->all_to_all, APDS_ERR_BAD_ARG, "the device-callback transport needs both callbacks");
            h->tr = std::make_unique<DeviceCallbackTransport>(*dt, rank, world);
        } else {
            fail(APDS_ERR_BAD_ARG, "unknown transport");
        }
        h->tr->rank = rank;
        h->tr->world = world;
        h->m = std::make_unique<Matcher>(h->dev, *h->tr, rows64_dev, n_rows, index_base, /*force_exchange=*/transport == APDS_TRANSPORT_RCCL || transport == APDS_TRANSPORT_DEVICE);
        *shard = h.release();
    });
}

int apds_shard_destroy(void* shard) {
    return shard_guarded([&] {
        if (!shard) return;
        ShardHandle* h = static_cast<ShardHandle*>(shard);
        DeviceGuard on(h->device);
        (void)hipDeviceSynchronize();
        h->m.reset();     // slots first (device buffers), then the communicator
        h->tr.reset();
        delete h;
    });
}

int apds_shard_info(const void* shard, int* rank, int* world, int64_t* n_rows, uint32_t* index_base, const char** transport_name, int* rccl_version) {
    return shard_guarded([&] {
        const ShardHandle* h = handle(const_cast<void*>(shard));
        if (rank) *rank = h->m->rank();
        if (world) *world = h->m->world();
        if (n_rows) *n_rows = h->m->shard_rows();
        if (index_base) *index_base = h->m->shard_base();
        if (transport_name) *transport_name = h->m->transport_name();
        if (rccl_version) {
            *rccl_version = 0;
            (void)ncclGetVersion(rccl_version);
        }
    });
}

int apds_shard_counts(void* shard, int n_query, int* counts, void* stream) {
    return shard_guarded([&] {
        APDS_REQUIRE(counts, APDS_ERR_BAD_ARG, "null output");
        handle(shard)->m->exchange_counts(n_query, counts, stream);
    });
}

int apds_shard_knn(void* shard, const void* q_rows64_dev, int n_query, const int* counts, int k, void* out_keys_dev, void* stream) {
    APDS_RANGE("apds_shard_knn");
    return shard_guarded([&] {
        APDS_REQUIRE(n_query >= 0 && (q_rows64_dev || n_query == 0), APDS_ERR_ASSERT, "bad query rows");
        APDS_REQUIRE(out_keys_dev || n_query == 0, APDS_ERR_BAD_ARG, "null output");
        APDS_REQUIRE(k >= 1 && k <= KMAX, APDS_ERR_ASSERT, "1 <= k <= 4096");
        handle(shard)->m->knn(q_rows64_dev, n_query, counts, k, out_keys_dev, stream);
    });
}

int apds_shard_knn_replicated(void* shard, const void* q_rows64_dev, int n_query, int root, int k, void* out_keys_dev, void* stream) {
    APDS_RANGE("apds_shard_knn_replicated");
    return shard_guarded([&] {
        ShardHandle* h = handle(shard);
        APDS_REQUIRE(n_query >= 0, APDS_ERR_ASSERT, "bad query count");
        APDS_REQUIRE(root < h->m->world(), APDS_ERR_BAD_ARG, "root outside [0, world)");
        APDS_REQUIRE(q_rows64_dev || n_query == 0 || (root >= 0 && root != h->m->rank()), APDS_ERR_ASSERT, "this rank has to bring the query rows");
        APDS_REQUIRE(out_keys_dev || n_query == 0, APDS_ERR_BAD_ARG, "null output");
        APDS_REQUIRE(k >= 1 && k <= KMAX, APDS_ERR_ASSERT, "1 <= k <= 4096");
        h->m->knn_replicated(q_rows64_dev, n_query, root, k, out_keys_dev, stream);
    });
}

int apds_shard_slot_create(void* shard, int max_queries, int kmax, void** slot) {
    return shard_guarded([&] {
        APDS_REQUIRE(slot, APDS_ERR_BAD_ARG, "null output");
        *slot = handle(shard)->m->slot_create(max_queries, kmax);
    });
}

int apds_shard_slot_destroy(void* shard, void* slot) {
    return shard_guarded([&] {
        if (!slot) return;
        ShardHandle* h = handle(shard);
        DeviceGuard on(h->device);
        (void)hipDeviceSynchronize();
        h->m->slot_destroy(static_cast<Slot*>(slot));
    });
}

int apds_shard_gather(void* shard, void* slot, const void* q_rows64_dev, int n_query, const int* counts, void* stream) {
    APDS_RANGE("apds_shard_gather");
    return shard_guarded([&] {
        APDS_REQUIRE(slot && counts, APDS_ERR_BAD_ARG, "null slot / counts");
        APDS_REQUIRE(n_query >= 0 && (q_rows64_dev || n_query == 0), APDS_ERR_ASSERT, "bad query rows");
        handle(shard)->m->gather(*static_cast<Slot*>(slot), q_rows64_dev, n_query, counts, stream);
    });
}

int apds_shard_scan(void* shard, void* slot, int k, void* stream) {
    APDS_RANGE("apds_shard_scan");
    return shard_guarded([&] {
        APDS_REQUIRE(slot, APDS_ERR_BAD_ARG, "null slot");
        handle(shard)->m->scan(*static_cast<Slot*>(slot), k, stream);
    });
}

int apds_shard_exchange_merge(void* shard, void* slot, int k, void* out_keys_dev, void* stream) {
    APDS_RANGE("apds_shard_exchange_merge");
    return shard_guarded([&] {
        APDS_REQUIRE(slot, APDS_ERR_BAD_ARG, "null slot");
        handle(shard)->m->exchange_merge(*static_cast<Slot*>(slot), k, out_keys_dev, stream);
    });
}

int apds_db_shard(void* db, int rank, int world, int transport, const apds_comm_id* id, const apds_host_transport* host, void** shard) {
    return shard_guarded([&] {
        APDS_REQUIRE(db && shard, APDS_ERR_BAD_ARG, "null argument");
        APDS_REQUIRE(world >= 1 && rank >= 0 && rank < world, APDS_ERR_BAD_ARG, "rank outside [0, world)");
        void *rows = nullptr, *kps = nullptr, *ids = nullptr, *img = nullptr;
        int n = 0;
        const int rc = apds_db_view(db, &rows, &kps, &ids, &img, &n);
        if (rc != APDS_OK) fail(rc, apds_last_error());
        // block partition of the view's rows: rank r holds [r * n / world, (r + 1) * n / world); trainIdx stays the position in the view
        const int64_t lo = (int64_t)rank * n / world, hi = (int64_t)(rank + 1) * n / world;
        const int rc2 = apds_shard_create(shard, rank, world, transport, id, host, static_cast<const char*>(rows) + lo * 64, hi - lo, (uint32_t)lo);
        if (rc2 != APDS_OK) fail(rc2, apds_last_error());
    });
}

}  // extern "C"
