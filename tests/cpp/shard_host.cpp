// tests/cpp/shard_host.cpp — TEST BUILD of the multi-GPU match choreography for machines without a GPU.
//
// csrc/shard_core.h (the text the product compiles into libapds_hip.so behind apds_shard_*) instantiated with a Device whose memory is
// host memory and whose local compute (top-k of a shard, u64-min merge) is INJECTED by the test, and with the host-callback transport.
// tests/test_sharded_matcher_cpu.py builds it with g++ and drives it from world-2 / world-4 gloo process groups: what is under test is
// the sharding / collective logic - message sizes, offsets, ordering of the steps, tie-break through the merge - which then runs
// unchanged over RCCL. Not part of the product; nothing outside tests/ builds or loads it.
#include <cstdlib>
#include <string>

#include "../../cubesat-apds_amd/csrc/shard_core.h"

using namespace apds::shard;

extern "C" {
typedef void (*shardhost_topk_fn)(void* user, const void* q_rows64, int nq, const void* rows64, int64_t n_rows, uint32_t index_base, int k, void* out_keys);
typedef void (*shardhost_merge_fn)(void* user, const void* parts, int nparts, int nq, int k, void* out_keys);
}

namespace {

thread_local std::string g_error;

struct HostDevice final : Device {
    shardhost_topk_fn topk_cb;
    shardhost_merge_fn merge_cb;
    void* user;
    void* alloc(size_t bytes) override {
        void* p = std::calloc(bytes ? bytes : 1, 1);
        if (!p) throw std::bad_alloc();
        return p;
    }
    void release(void* p) override { std::free(p); }
    void copy(void* dst, const void* src, size_t bytes, void*) override { std::memmove(dst, src, bytes); }
    void to_host(void* host, const void* dev, size_t bytes, void*) override { std::memcpy(host, dev, bytes); }
    void from_host(void* dev, const void* host, size_t bytes, void*) override { std::memcpy(dev, host, bytes); }
    void topk(const void* q, int nq, const void* rows, int64_t n_rows, uint32_t base, int k, void* out, void*) override {
        topk_cb(user, q, nq, rows, n_rows, base, k, out);
    }
    void merge(const void* parts, int nparts, int nq, int k, void* out, void*) override { merge_cb(user, parts, nparts, nq, k, out); }
    void* event_create() override { return this; }
    void event_destroy(void*) override {}
    void event_record(void*, void*) override {}
    void stream_wait(void*, void*) override {}
};

struct Handle {
    HostDevice dev;
    HostTransport* tr = nullptr;
    Matcher* m = nullptr;
    ~Handle() {
        delete m;
        delete tr;
    }
};

template <class F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const ShardError& e) {
        g_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -2;
    }
}

}  // namespace

extern "C" {

const char* shardhost_last_error(void) { return g_error.c_str(); }

int shardhost_create(void** out, int rank, int world, const HostCallbacks* cb, const void* rows64, int64_t n_rows, uint32_t index_base, shardhost_topk_fn topk,
                     shardhost_merge_fn merge, void* user) {
    return guarded([&] {
        Handle* h = new Handle();
        h->dev.topk_cb = topk;
        h->dev.merge_cb = merge;
        h->dev.user = user;
        try {
            h->tr = new HostTransport(h->dev, *cb, rank, world);
            h->m = new Matcher(h->dev, *h->tr, rows64, n_rows, index_base, /*force_exchange=*/false);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
    });
}
int shardhost_destroy(void* h) {
    return guarded([&] { delete static_cast<Handle*>(h); });
}
int shardhost_counts(void* h, int nq, int* counts) {
    return guarded([&] { static_cast<Handle*>(h)->m->exchange_counts(nq, counts, nullptr); });
}
int shardhost_knn(void* h, const void* q, int nq, const int* counts, int k, void* out_keys) {
    return guarded([&] { static_cast<Handle*>(h)->m->knn(q, nq, counts, k, out_keys, nullptr); });
}
int shardhost_knn_replicated(void* h, const void* q, int nq, int root, int k, void* out_keys) {
    return guarded([&] { static_cast<Handle*>(h)->m->knn_replicated(q, nq, root, k, out_keys, nullptr); });
}
int shardhost_slot_create(void* h, int max_queries, int kmax, void** slot) {
    return guarded([&] { *slot = static_cast<Handle*>(h)->m->slot_create(max_queries, kmax); });
}
int shardhost_slot_destroy(void* h, void* slot) {
    return guarded([&] { static_cast<Handle*>(h)->m->slot_destroy(static_cast<Slot*>(slot)); });
}
int shardhost_slot_pad(void* slot) { return static_cast<Slot*>(slot)->pad; }
int shardhost_gather(void* h, void* slot, const void* q, int nq, const int* counts) {
    return guarded([&] { static_cast<Handle*>(h)->m->gather(*static_cast<Slot*>(slot), q, nq, counts, nullptr); });
}
int shardhost_scan(void* h, void* slot, int k) {
    return guarded([&] { static_cast<Handle*>(h)->m->scan(*static_cast<Slot*>(slot), k, nullptr); });
}
int shardhost_exchange_merge(void* h, void* slot, int k, void* out_keys) {
    return guarded([&] { static_cast<Handle*>(h)->m->exchange_merge(*static_cast<Slot*>(slot), k, out_keys, nullptr); });
}

}  // extern "C"
