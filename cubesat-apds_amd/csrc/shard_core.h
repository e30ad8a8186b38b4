// csrc/shard_core.h — the multi-GPU match choreography (BASELINE config 5), free of HIP and RCCL headers.
//
// The reference has no multi-device code: its only parallelism is the rayon pool of the preprocessor
// (/root/reference/preprocessor/src/main.rs:86-89,227-245) and its train set is whatever keypointdb.rs:50-90 returns. Here the train
// set is row-sharded over the GPUs of one node: rank r holds rows [index_base, index_base + n_rows) resident in its HBM, frames
// are data-parallel (every rank matches its own queries against the WHOLE database), and the match step has one exchange:
//
//   (1) all-gather of the ranks' query rows (message = the frame's largest count rounded up to MSG_ROUND rows: a function of the
//       counts alone, so every rank posts the same size);
//   (2) local top-k of ALL gathered queries against the local shard; keys are (distance << 32 | global row);
//   (3) all-to-all of the keys: rank r sends rank p the [counts[p], k] block of p's queries, receives [counts[r], k] from every shard
//       (1/world of an all-gather's bytes; the result is what north_star's "all-gather of per-shard top-k" gives);
//   (4) u64-min merge of the `world` candidate lists of every query = the single-GPU answer, lowest-index tie-break included.
//
// The strong-scaling (latency) form, SURVEY 8e's literal shape, is Matcher::knn_replicated: ONE frame, the same queries on every rank
// (each rank extracted the frame itself, or one rank's rows are broadcast first), local top-k per shard, ALL-GATHER of the [nq, k] key
// lists, u64-min merge on every rank: every rank ends with the whole frame's answer after 1/world of the scan.
//
// Matcher is written against two small interfaces so that the SAME text runs in three places: the product (Device = HIP kernels and
// hipMemcpyAsync, Transport = RCCL), the g++-built loopback test (threads as ranks on one GPU, Transport = pointer exchange inside
// one process), and the CPU tests (tests/cpp/shard_host.cpp: Device = host memory with the local compute injected by the test,
// Transport = the host callbacks, world-2/4 gloo process groups).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace apds {
namespace shard {

constexpr int MSG_ROUND = 1024;   // the query all-gather moves multiples of this many rows per rank
constexpr int KMAX = 4096;   // per-query neighbours a slot may carry: a bound on the exchange buffers, not on the scan (which serves any k, above 16 in pages of 16)

struct ShardError : std::runtime_error {
    int code;
    ShardError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// What the choreography needs from "the device". `stream` is opaque (hipStream_t in the product, ignored on the host).
struct Device {
    virtual ~Device() {}
    virtual void* alloc(size_t bytes) = 0;
    virtual void release(void* p) = 0;
    virtual void copy(void* dst, const void* src, size_t bytes, void* stream) = 0;        // device -> device, ordered on `stream`
    virtual void to_host(void* host, const void* dev, size_t bytes, void* stream) = 0;    // ordered on `stream`, complete on return
    virtual void from_host(void* dev, const void* host, size_t bytes, void* stream) = 0;  // ordered on `stream`, host buffer reusable on return
    virtual void topk(const void* q_rows64, int nq, const void* rows64, int64_t n_rows, uint32_t index_base, int k, void* out_keys, void* stream) = 0;
    virtual void merge(const void* parts, int nparts, int nq, int k, void* out_keys, void* stream) = 0;
    virtual void* event_create() = 0;
    virtual void event_destroy(void* ev) = 0;
    virtual void event_record(void* ev, void* stream) = 0;
    virtual void stream_wait(void* stream, void* ev) = 0;
};

// The exchange step. Every method is a collective: all ranks call it, in the same order.
struct Transport {
    int rank = 0, world = 1;
    virtual ~Transport() {}
    virtual const char* name() const = 0;
    // every rank's query count, as host ints
    virtual void counts(int mine, int* all, void* stream) = 0;
    // recv = the ranks' `bytes` back to back, rank-major
    virtual void all_gather(const void* send, void* recv, size_t bytes, void* stream) = 0;
    // to rank p: send[soff[p] .. +sbytes[p]); from rank p: rbytes[p] bytes to recv + roff[p]
    virtual void all_to_all(const void* send, const size_t* soff, const size_t* sbytes, void* recv, const size_t* roff, const size_t* rbytes, void* stream) = 0;
    // buf[0 .. bytes) of rank `root` replaces buf on every rank. `scratch` holds world * bytes (used by the default form only: an
    // all-gather of which block `root` is kept - what a transport without a native broadcast does).
    virtual void broadcast(void* buf, size_t bytes, int root, void* scratch, Device& dev, void* stream) {
        all_gather(buf, scratch, bytes, stream);
        if (bytes) dev.copy(buf, static_cast<char*>(scratch) + (size_t)root * bytes, bytes, stream);
    }
};

// A communicator supplied by the host program as two callbacks on HOST buffers (gloo, MPI, a test harness): the device data is
// staged through host memory around them. Synchronous with the host by construction; the rehearsal / test transport.
struct HostCallbacks {
    void* user;
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank);
    int (*all_to_all)(void* user, const void* send, const size_t* send_off, const size_t* send_bytes, void* recv, const size_t* recv_off,
                      const size_t* recv_bytes);
};

class HostTransport : public Transport {
    Device& dev;
    HostCallbacks cb;
    std::vector<char> hs, hr;

public:
    HostTransport(Device& d, const HostCallbacks& c, int rank_, int world_) : dev(d), cb(c) {
        rank = rank_;
        world = world_;
        if (!cb.all_gather || !cb.all_to_all) throw ShardError(-5, "host transport needs both callbacks");
    }
    const char* name() const override { return "host-callbacks"; }
    void counts(int mine, int* all, void*) override {
        int32_t v = mine;
        std::vector<int32_t> r((size_t)world);
        if (cb.all_gather(cb.user, &v, r.data(), sizeof(int32_t))) throw ShardError(-2, "host all_gather callback failed");
        for (int p = 0; p < world; p++) all[p] = r[(size_t)p];
    }
    void all_gather(const void* send, void* recv, size_t bytes, void* stream) override {
        hs.resize(bytes);
        hr.resize(bytes * (size_t)world);
        dev.to_host(hs.data(), send, bytes, stream);
        if (cb.all_gather(cb.user, hs.data(), hr.data(), bytes)) throw ShardError(-2, "host all_gather callback failed");
        dev.from_host(recv, hr.data(), hr.size(), stream);
    }
    void all_to_all(const void* send, const size_t* soff, const size_t* sbytes, void* recv, const size_t* roff, const size_t* rbytes, void* stream) override {
        size_t send_end = 0, recv_end = 0;
        for (int p = 0; p < world; p++) {
            send_end = std::max(send_end, soff[p] + sbytes[p]);
            recv_end = std::max(recv_end, roff[p] + rbytes[p]);
        }
        hs.resize(std::max<size_t>(send_end, 1));
        hr.resize(std::max<size_t>(recv_end, 1));
        if (send_end) dev.to_host(hs.data(), send, send_end, stream);
        if (cb.all_to_all(cb.user, hs.data(), soff, sbytes, hr.data(), roff, rbytes)) throw ShardError(-2, "host all_to_all callback failed");
        if (recv_end) dev.from_host(recv, hr.data(), recv_end, stream);
    }
};

// One frame's exchange buffers: allocated once, reused by every frame that goes through the slot.
struct Slot {
    int pad = 0;                 // capacity in query rows per rank (a multiple of MSG_ROUND)
    char* mine = nullptr;        // [pad, 64]            this rank's rows, zero padded to the message size
    char* gathered = nullptr;    // [world, pad, 64]     every rank's padded rows
    char* all_q = nullptr;       // [world * pad, 64]    the ranks' rows back to back without padding (what the shard is scanned with)
    uint64_t* local = nullptr;   // [world * pad, KMAX]  this shard's top-k of all of them
    uint64_t* recv = nullptr;    // [world, pad, KMAX]   every shard's top-k of THIS rank's queries
    uint64_t* merged = nullptr;  // [pad, KMAX]
    std::vector<int> counts;
    int total = 0, nq = 0, kmax = 2;
    void* gathered_ev = nullptr;   // recorded when all_q is complete: a scan on another stream waits for it
};

class Matcher {
    Device& dev;
    Transport& tr;
    const void* rows;
    int64_t n_rows;
    uint32_t index_base;
    Slot* own = nullptr;   // buffers of the one-call form
    bool force_exchange;   // run the collectives even with one rank (the RCCL self-test a one-GPU box allows)

    static int round_up(int v) { return std::max(MSG_ROUND, (v + MSG_ROUND - 1) / MSG_ROUND * MSG_ROUND); }

public:
    Matcher(Device& d, Transport& t, const void* rows64, int64_t n, uint32_t base, bool force_exchange_ = false)
        : dev(d), tr(t), rows(rows64), n_rows(n), index_base(base), force_exchange(force_exchange_) {
        if (n < 0) throw ShardError(-215, "negative shard row count");
        if (t.world < 1 || t.rank < 0 || t.rank >= t.world) throw ShardError(-5, "rank outside [0, world)");
    }
    ~Matcher() {
        if (own) slot_destroy(own);
    }
    int rank() const { return tr.rank; }
    int world() const { return tr.world; }
    int64_t shard_rows() const { return n_rows; }
    uint32_t shard_base() const { return index_base; }
    const char* transport_name() const { return tr.name(); }

    Slot* slot_create(int max_queries, int kmax) {
        if (kmax < 1 || kmax > KMAX) throw ShardError(-215, "1 <= k <= 4096");
        Slot* s = new Slot();
        try {
            const size_t w = (size_t)tr.world;
            s->pad = round_up(std::max(max_queries, 1));
            s->kmax = kmax;
            const size_t pad = (size_t)s->pad;
            s->mine = static_cast<char*>(dev.alloc(pad * 64));
            s->gathered = static_cast<char*>(dev.alloc(w * pad * 64));
            s->all_q = static_cast<char*>(dev.alloc(w * pad * 64));
            s->local = static_cast<uint64_t*>(dev.alloc(w * pad * kmax * 8));
            s->recv = static_cast<uint64_t*>(dev.alloc(w * pad * kmax * 8));
            s->merged = static_cast<uint64_t*>(dev.alloc(pad * kmax * 8));
            s->gathered_ev = dev.event_create();
        } catch (...) {
            slot_destroy(s);
            throw;
        }
        return s;
    }
    void slot_destroy(Slot* s) {
        if (!s) return;
        for (void* p : {(void*)s->mine, (void*)s->gathered, (void*)s->all_q, (void*)s->local, (void*)s->recv, (void*)s->merged})
            if (p) dev.release(p);
        if (s->gathered_ev) dev.event_destroy(s->gathered_ev);
        delete s;
    }

    // every rank's query count of one frame (collective)
    void exchange_counts(int nq, int* all, void* stream) {
        if (nq < 0) throw ShardError(-215, "negative query count");
        if (tr.world == 1 && !force_exchange) {
            all[0] = nq;
            return;
        }
        tr.counts(nq, all, stream);
    }

    // step (1) on `stream`: afterwards s.all_q[0 .. total) holds every rank's queries, rank-major
    void gather(Slot& s, const void* q_rows64, int nq, const int* counts, void* stream) {
        const int w = tr.world;
        int maxc = 0;
        long long total = 0;
        for (int p = 0; p < w; p++) {
            if (counts[p] < 0) throw ShardError(-215, "negative query count");
            maxc = std::max(maxc, counts[p]);
            total += counts[p];
        }
        if (counts[tr.rank] != nq) throw ShardError(-215, "counts[rank] differs from this rank's query count");
        const int msg_rows = round_up(maxc);
        if (msg_rows > s.pad)
            throw ShardError(-215, "exchange slot holds " + std::to_string(s.pad) + " query rows per rank, this frame needs " + std::to_string(msg_rows));
        s.counts.assign(counts, counts + w);
        s.total = (int)total;
        s.nq = nq;
        if (w == 1 && !force_exchange) {   // nothing to exchange: the scan reads the caller's rows through all_q all the same (one copy, one code path)
            if (nq) dev.copy(s.all_q, q_rows64, (size_t)nq * 64, stream);
        } else {
            if (nq) dev.copy(s.mine, q_rows64, (size_t)nq * 64, stream);
            tr.all_gather(s.mine, s.gathered, (size_t)msg_rows * 64, stream);
            size_t off = 0;
            for (int p = 0; p < w; p++) {   // drop the padding: `world` slice copies
                if (counts[p]) dev.copy(s.all_q + off * 64, s.gathered + (size_t)p * msg_rows * 64, (size_t)counts[p] * 64, stream);
                off += (size_t)counts[p];
            }
        }
        dev.event_record(s.gathered_ev, stream);
    }

    // step (2) on `stream` (may differ from the gather's stream): no collective
    void scan(Slot& s, int k, void* stream) {
        if (k < 1 || k > s.kmax) throw ShardError(-215, "k outside the slot's range");
        dev.stream_wait(stream, s.gathered_ev);
        if (s.total) dev.topk(s.all_q, s.total, rows, n_rows, index_base, k, s.local, stream);
    }

    // steps (3) + (4) on the stream the scan ran on; out_keys: [nq, k] (nullptr: the slot's own buffer). Returns where the keys are.
    uint64_t* exchange_merge(Slot& s, int k, void* out_keys, void* stream) {
        if (k < 1 || k > s.kmax) throw ShardError(-215, "k outside the slot's range");
        const int w = tr.world, nq = s.nq;
        uint64_t* dst = out_keys ? static_cast<uint64_t*>(out_keys) : s.merged;
        if (w == 1 && !force_exchange) {
            if (nq) dev.copy(dst, s.local, (size_t)nq * k * 8, stream);
            return dst;
        }
        std::vector<size_t> soff((size_t)w), sbytes((size_t)w), roff((size_t)w), rbytes((size_t)w);
        size_t off = 0;
        for (int p = 0; p < w; p++) {
            soff[(size_t)p] = off * k * 8;
            sbytes[(size_t)p] = (size_t)s.counts[(size_t)p] * k * 8;
            off += (size_t)s.counts[(size_t)p];
            roff[(size_t)p] = (size_t)p * nq * k * 8;
            rbytes[(size_t)p] = (size_t)nq * k * 8;
        }
        tr.all_to_all(s.local, soff.data(), sbytes.data(), s.recv, roff.data(), rbytes.data(), stream);
        if (nq) dev.merge(s.recv, w, nq, k, dst, stream);
        return dst;
    }

    // Strong-scaling form on slot `s`: every rank holds the SAME nq queries (root < 0), or rank `root`'s rows are broadcast first (the other
    // ranks' q_rows64 is ignored; nq must be the same number on every rank either way). Every rank scans its own shard, the [nq, k] key lists
    // are all-gathered and merged: out_keys [nq, k] on EVERY rank = the single-device answer for the whole frame.
    uint64_t* replicated(Slot& s, const void* q_rows64, int nq, int root, int k, void* out_keys, void* stream) {
        if (k < 1 || k > s.kmax) throw ShardError(-215, "k outside the slot's range");
        if (nq < 0 || root >= tr.world) throw ShardError(-215, "bad query count / root");
        if (round_up(nq) > s.pad) throw ShardError(-215, "exchange slot holds " + std::to_string(s.pad) + " query rows, this frame has " + std::to_string(nq));
        const int w = tr.world;
        uint64_t* dst = out_keys ? static_cast<uint64_t*>(out_keys) : s.merged;
        s.nq = s.total = nq;
        if (!nq) return dst;
        const bool exchange = w > 1 || force_exchange;
        if (root < 0 || tr.rank == root || !exchange) dev.copy(s.all_q, q_rows64, (size_t)nq * 64, stream);
        if (root >= 0 && exchange) tr.broadcast(s.all_q, (size_t)round_up(nq) * 64, root, s.gathered, dev, stream);
        dev.topk(s.all_q, nq, rows, n_rows, index_base, k, s.local, stream);
        if (!exchange) {
            dev.copy(dst, s.local, (size_t)nq * k * 8, stream);
            return dst;
        }
        tr.all_gather(s.local, s.recv, (size_t)nq * k * 8, stream);   // recv = [world, nq, k]: the layout the merge reads
        dev.merge(s.recv, w, nq, k, dst, stream);
        return dst;
    }
    uint64_t* knn_replicated(const void* q_rows64, int nq, int root, int k, void* out_keys, void* stream) {
        if (!own || own->pad < round_up(nq) || own->kmax < k) {
            if (own) slot_destroy(own);
            own = nullptr;
            own = slot_create(nq, std::max(k, 2));
        }
        return replicated(*own, q_rows64, nq, root, k, out_keys, stream);
    }

    // the one-call form: counts == nullptr exchanges them first (a host synchronisation on most transports)
    uint64_t* knn(const void* q_rows64, int nq, const int* counts, int k, void* out_keys, void* stream) {
        std::vector<int> c((size_t)tr.world);
        if (counts) c.assign(counts, counts + tr.world);
        else exchange_counts(nq, c.data(), stream);
        int maxc = 0;
        for (int v : c) maxc = std::max(maxc, v);
        if (!own || own->pad < round_up(maxc) || own->kmax < k) {
            if (own) slot_destroy(own);
            own = nullptr;
            own = slot_create(maxc, std::max(k, 2));
        }
        gather(*own, q_rows64, nq, c.data(), stream);
        scan(*own, k, stream);
        return exchange_merge(*own, k, out_keys, stream);
    }
};

}  // namespace shard
}  // namespace apds
