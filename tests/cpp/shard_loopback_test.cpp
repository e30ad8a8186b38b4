// tests/cpp/shard_loopback_test.cpp — a host that is NOT Python drives the row-sharded matcher through the C ABI alone (include/apds.h).
// Built by g++ against libapds_hip.so (tests/test_shard_native.py), run on the GPU box.
//
// Ranks are THREADS of this process sharing one GPU (APDS_TRANSPORT_LOOPBACK): every rank keeps a block of the train rows resident,
// brings its own queries and must receive, for ITS queries, exactly the keys apds_dev_hamming_topk gives over the unsharded rows -
// bit for bit, a cross-shard tie resolved towards the lower global row - at world 2 and 4, in the one-call form, with the counts
// exchanged ahead, and in the split form with two frames in flight. Then the RCCL transport at world size 1 (all a one-GPU box
// allows: communicator set-up, ncclAllGather and the send/recv group really run), and the shard-of-a-table entry (apds_db_shard).
#include <apds.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

std::atomic<int> failures{0};
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, "\n");                         \
            failures++;                                    \
        }                                                  \
    } while (0)
#define OK(call)                                                                              \
    do {                                                                                      \
        const int rc_ = (call);                                                               \
        if (rc_ != 0) {                                                                       \
            fprintf(stderr, "FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, apds_last_error()); \
            failures++;                                                                       \
        }                                                                                     \
    } while (0)

struct SplitMix {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};

// n rows of 64 bytes: 486 random bits, the rest zero (the layout of an M-LDB descriptor row)
std::vector<uint8_t> random_rows(int n, uint64_t seed) {
    std::vector<uint8_t> r((size_t)n * 64, 0);
    SplitMix g{seed};
    for (int i = 0; i < n; i++) {
        uint64_t* w = reinterpret_cast<uint64_t*>(&r[(size_t)i * 64]);
        for (int j = 0; j < 8; j++) w[j] = g.next();
        r[(size_t)i * 64 + 60] &= 0x3F;
        r[(size_t)i * 64 + 61] = r[(size_t)i * 64 + 62] = r[(size_t)i * 64 + 63] = 0;
    }
    return r;
}

// queries: every third one is a train row with a few bits flipped, the rest random
std::vector<uint8_t> make_queries(const std::vector<uint8_t>& db, int nt, int nq, uint64_t seed) {
    std::vector<uint8_t> q = random_rows(nq, seed);
    SplitMix g{seed ^ 0xABCDEF};
    for (int i = 0; i < nq; i += 3) {
        const int src = (int)(g.next() % (uint64_t)nt);
        std::memcpy(&q[(size_t)i * 64], &db[(size_t)src * 64], 64);
        for (int f = 0; f < 12; f++) {
            const int bit = (int)(g.next() % 486);
            q[(size_t)i * 64 + bit / 8] ^= (uint8_t)(1u << (bit % 8));
        }
    }
    return q;
}

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { OK(apds_dev_alloc(bytes, &p)); }
    ~DevBuf() { apds_dev_release(p); }
    DevBuf(const DevBuf&) = delete;
};

std::vector<uint64_t> single_device_keys(const std::vector<uint8_t>& q, int nq, const std::vector<uint8_t>& db, int nt, int k) {
    std::vector<uint64_t> keys((size_t)nq * k);
    if (nq == 0) return keys;
    DevBuf dq(q.size()), dt(db.size()), dk(keys.size() * 8);
    OK(apds_dev_upload(dq.p, q.data(), q.size(), nullptr));
    OK(apds_dev_upload(dt.p, db.data(), db.size(), nullptr));
    OK(apds_dev_hamming_topk(dq.p, nq, dt.p, nt, 0, k, dk.p, nullptr));
    OK(apds_dev_download(keys.data(), dk.p, keys.size() * 8, nullptr));
    return keys;
}

void rank_main(int rank, int world, int transport, const apds_comm_id* id, const std::vector<uint8_t>* db, int nt, const std::vector<std::vector<uint8_t>>* queries,
               const std::vector<std::vector<uint64_t>>* want, int k) {
    OK(apds_set_device(0));
    const int lo = (int)((int64_t)rank * nt / world), hi = (int)((int64_t)(rank + 1) * nt / world);
    DevBuf rows((size_t)(hi - lo) * 64 + 64);
    OK(apds_dev_upload(rows.p, db->data() + (size_t)lo * 64, (size_t)(hi - lo) * 64, nullptr));
    OK(apds_stream_synchronize(nullptr));
    void* shard = nullptr;
    OK(apds_shard_create(&shard, rank, world, transport, id, nullptr, rows.p, hi - lo, (uint32_t)lo));
    if (!shard) return;
    int r = -1, w = -1, ver = 0;
    int64_t nr = -1;
    uint32_t base = 0;
    const char* tname = nullptr;
    OK(apds_shard_info(shard, &r, &w, &nr, &base, &tname, &ver));
    CHECK(r == rank && w == world && nr == hi - lo && base == (uint32_t)lo && tname, "shard info");
    const std::vector<uint8_t>& q = (*queries)[(size_t)rank];
    const int nq = (int)(q.size() / 64);
    const std::vector<uint64_t>& expect = (*want)[(size_t)rank];
    DevBuf dq(q.size() + 64), dk((size_t)nq * k * 8 + 64), dk2((size_t)nq * k * 8 + 64);
    OK(apds_dev_upload(dq.p, q.data(), q.size(), nullptr));
    std::vector<uint64_t> got((size_t)nq * k);
    // (1) one call, counts exchanged inside
    OK(apds_shard_knn(shard, dq.p, nq, nullptr, k, dk.p, nullptr));
    OK(apds_dev_download(got.data(), dk.p, got.size() * 8, nullptr));
    CHECK(got == expect, "rank %d/%d: one-call keys differ from the single-device keys", rank, world);
    // (2) counts ahead
    std::vector<int> counts((size_t)world, -1);
    OK(apds_shard_counts(shard, nq, counts.data(), nullptr));
    for (int p = 0; p < world; p++) CHECK(counts[(size_t)p] == (int)((*queries)[(size_t)p].size() / 64), "rank %d: counts[%d] = %d", rank, p, counts[(size_t)p]);
    std::fill(got.begin(), got.end(), 0);
    OK(apds_shard_knn(shard, dq.p, nq, counts.data(), k, dk.p, nullptr));
    OK(apds_dev_download(got.data(), dk.p, got.size() * 8, nullptr));
    CHECK(got == expect, "rank %d/%d: keys with counts ahead differ", rank, world);
    // (3) split form, two frames in flight: frame B = the same queries in reverse order; B's gather goes out (on a second stream) before
    // A's key exchange, as a pipeline issues them
    std::vector<uint8_t> qb(q.size());
    for (int i = 0; i < nq; i++) std::memcpy(&qb[(size_t)i * 64], &q[(size_t)(nq - 1 - i) * 64], 64);
    DevBuf dqb(qb.size() + 64);
    OK(apds_dev_upload(dqb.p, qb.data(), qb.size(), nullptr));
    OK(apds_stream_synchronize(nullptr));
    int maxc = 1;
    for (int c : counts) maxc = c > maxc ? c : maxc;
    void *slotA = nullptr, *slotB = nullptr, *side = nullptr;
    OK(apds_shard_slot_create(shard, maxc, k, &slotA));
    OK(apds_shard_slot_create(shard, maxc, k, &slotB));
    OK(apds_stream_create(0, nullptr, 0, &side));
    OK(apds_shard_gather(shard, slotA, dq.p, nq, counts.data(), nullptr));
    OK(apds_shard_scan(shard, slotA, k, nullptr));
    OK(apds_shard_gather(shard, slotB, dqb.p, nq, counts.data(), side));
    OK(apds_shard_exchange_merge(shard, slotA, k, dk.p, nullptr));
    OK(apds_shard_scan(shard, slotB, k, nullptr));
    OK(apds_shard_exchange_merge(shard, slotB, k, dk2.p, nullptr));
    std::vector<uint64_t> gotB((size_t)nq * k);
    OK(apds_dev_download(got.data(), dk.p, got.size() * 8, nullptr));
    OK(apds_dev_download(gotB.data(), dk2.p, gotB.size() * 8, nullptr));
    CHECK(got == expect, "rank %d/%d: split-form keys of frame A differ", rank, world);
    bool rev = true;
    for (int i = 0; i < nq && rev; i++)
        for (int j = 0; j < k; j++) rev = rev && gotB[(size_t)i * k + j] == expect[(size_t)(nq - 1 - i) * k + j];
    CHECK(rev, "rank %d/%d: split-form keys of frame B differ", rank, world);
    OK(apds_stream_synchronize(side));
    OK(apds_shard_slot_destroy(shard, slotA));
    OK(apds_shard_slot_destroy(shard, slotB));
    OK(apds_stream_destroy(side));
    // (4) the strong-scaling form: ONE frame (rank 0's queries) known to every rank, or brought by rank `world - 1` alone and broadcast;
    // every rank must end with the whole frame's single-device keys
    {
        const std::vector<uint8_t>& q0 = (*queries)[0];
        const int n0 = (int)(q0.size() / 64);
        const std::vector<uint64_t>& expect0 = (*want)[0];
        DevBuf dq0(q0.size() + 64), dr((size_t)n0 * k * 8 + 64);
        OK(apds_dev_upload(dq0.p, q0.data(), q0.size(), nullptr));
        std::vector<uint64_t> rep((size_t)n0 * k);
        OK(apds_shard_knn_replicated(shard, dq0.p, n0, -1, k, dr.p, nullptr));
        OK(apds_dev_download(rep.data(), dr.p, rep.size() * 8, nullptr));
        CHECK(rep == expect0, "rank %d/%d: replicated-form keys differ from the single-device keys", rank, world);
        std::fill(rep.begin(), rep.end(), 0);
        const int root = world - 1;
        OK(apds_shard_knn_replicated(shard, rank == root ? dq0.p : nullptr, n0, root, k, dr.p, nullptr));
        OK(apds_dev_download(rep.data(), dr.p, rep.size() * 8, nullptr));
        CHECK(rep == expect0, "rank %d/%d: broadcast-form keys differ from the single-device keys", rank, world);
    }
    OK(apds_shard_destroy(shard));
    OK(apds_thread_release());
}

void run_world(int world, int transport, int nt, const std::vector<int>& nqs, int k, const char* label) {
    std::vector<uint8_t> db = random_rows(nt, 0x44420001ull + (uint64_t)nt);
    // a cross-shard tie: a row of the LAST shard equals row 7, and rank 0's first query is that row: (7, copy) at distance 0, in this order
    const bool tie = nt >= 64 * world;   // (the randomised mode also runs worlds with a handful of rows, shards of zero rows included)
    const int twin = nt - nt / (2 * world) - 3;
    if (tie) std::memcpy(&db[(size_t)twin * 64], &db[(size_t)7 * 64], 64);
    std::vector<std::vector<uint8_t>> queries((size_t)world);
    std::vector<std::vector<uint64_t>> want((size_t)world);
    for (int r = 0; r < world; r++) {
        queries[(size_t)r] = make_queries(db, nt, nqs[(size_t)r], 0x51550001ull + (uint64_t)r * 977);
        if (tie && r == 0 && nqs[0] > 0) std::memcpy(queries[0].data(), &db[(size_t)7 * 64], 64);
        want[(size_t)r] = single_device_keys(queries[(size_t)r], nqs[(size_t)r], db, nt, k);
    }
    if (tie && nqs[0] > 0 && k >= 2)
        CHECK(want[0][0] == 7ull && want[0][1] == (uint64_t)twin, "tie case: single-device keys are (%llu, %llu), expected (7, %d)", (unsigned long long)want[0][0],
              (unsigned long long)want[0][1], twin);
    apds_comm_id id;
    OK(apds_comm_id_create(transport, &id));
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, r, world, transport, &id, &db, nt, &queries, &want, k);
    for (auto& t : th) t.join();
    printf("%s: world %d, %d rows, k %d ... %s\n", label, world, nt, k, failures.load() ? "FAILED" : "ok");
    fflush(stdout);
}

void table_shard_case() {
    // apds_db_shard: two thread-ranks, each with its own copy of the resident table and the same selection; keys == apds_db_knn_match order
    const int n = 3000, world = 2;
    std::vector<uint8_t> rows = random_rows(n, 0x7AB1E);
    std::vector<uint8_t> d61((size_t)n * 61);
    std::vector<apds_keypoint> kps((size_t)n);
    SplitMix g{99};
    for (int i = 0; i < n; i++) {
        std::memcpy(&d61[(size_t)i * 61], &rows[(size_t)i * 64], 61);
        kps[(size_t)i] = apds_keypoint{(float)(g.next() % 1000), (float)(g.next() % 1000), 4.f, 0.f, (float)(1 + g.next() % 100000) * 1e-3f, 0, (int)(g.next() % 4)};
    }
    std::vector<uint8_t> q = make_queries(rows, n, 200, 0xBEEF);
    std::vector<uint8_t> q61((size_t)200 * 61);
    for (int i = 0; i < 200; i++) std::memcpy(&q61[(size_t)i * 61], &q[(size_t)i * 64], 61);
    apds_comm_id id;
    OK(apds_comm_id_create(APDS_TRANSPORT_LOOPBACK, &id));
    auto body = [&](int rank) {
        OK(apds_set_device(0));
        void* db = nullptr;
        OK(apds_db_create(&db, n));
        OK(apds_db_insert_image(db, kps.data(), d61.data(), n, 1, 0, 0, 0, 1024, 1024));
        int nsel = 0;
        OK(apds_db_select(db, 0, 1, 0, 0, 0, 0, &nsel));
        CHECK(nsel == n, "selection holds %d rows", nsel);
        std::vector<int32_t> idx(400), dist(400);
        OK(apds_db_knn_match(db, q61.data(), 200, 61, 2, idx.data(), dist.data()));
        void* shard = nullptr;
        OK(apds_db_shard(db, rank, world, APDS_TRANSPORT_LOOPBACK, &id, nullptr, &shard));
        DevBuf dq(q.size()), dk(400 * 8);
        OK(apds_dev_upload(dq.p, q.data(), q.size(), nullptr));
        OK(apds_shard_knn(shard, dq.p, 200, nullptr, 2, dk.p, nullptr));
        std::vector<uint64_t> keys(400);
        OK(apds_dev_download(keys.data(), dk.p, 400 * 8, nullptr));
        bool same = true;
        for (int i = 0; i < 400; i++) same = same && (int32_t)(uint32_t)keys[(size_t)i] == idx[(size_t)i] && (int32_t)(keys[(size_t)i] >> 32) == dist[(size_t)i];
        CHECK(same, "rank %d: sharded keys over the table view differ from apds_db_knn_match", rank);
        OK(apds_shard_destroy(shard));
        OK(apds_db_destroy(db));
        OK(apds_thread_release());
    };
    std::thread a(body, 0), b(body, 1);
    a.join();
    b.join();
    printf("table shard (apds_db_shard): world 2 ... %s\n", failures.load() ? "FAILED" : "ok");
}

// `shard_loopback_test lag`: the event-lifetime race of round 3, deterministically. Started with APDS_TEST_LOOPBACK_LAG="0:300" rank 0
// sleeps 300 ms between the closing barrier of every collective and its closing hipStreamWaitEvent on the peers' `done` events; rank 1
// leaves its last collective at once, destroys its shard and releases its thread context while rank 0 is still asleep. With the events
// owned by the rank's transport (the round-3 bug) rank 0 then waits on destroyed events ("invalid resource handle"); they belong to the
// hub, which lives until the last rank detaches.
int lag_case() {
    const int world = 2, nt = 4000, k = 2, nq = 300;
    std::vector<uint8_t> db = random_rows(nt, 0x1A6);
    std::vector<std::vector<uint8_t>> queries((size_t)world);
    std::vector<std::vector<uint64_t>> want((size_t)world);
    for (int r = 0; r < world; r++) {
        queries[(size_t)r] = make_queries(db, nt, nq, 0x1A60 + (uint64_t)r);
        want[(size_t)r] = single_device_keys(queries[(size_t)r], nq, db, nt, k);
    }
    apds_comm_id id;
    OK(apds_comm_id_create(APDS_TRANSPORT_LOOPBACK, &id));
    auto body = [&](int rank) {
        OK(apds_set_device(0));
        const int lo = rank * nt / world, hi = (rank + 1) * nt / world;
        DevBuf rows((size_t)(hi - lo) * 64), dq((size_t)nq * 64), dk((size_t)nq * k * 8);
        OK(apds_dev_upload(rows.p, db.data() + (size_t)lo * 64, (size_t)(hi - lo) * 64, nullptr));
        OK(apds_dev_upload(dq.p, queries[(size_t)rank].data(), (size_t)nq * 64, nullptr));
        OK(apds_stream_synchronize(nullptr));
        void* shard = nullptr;
        OK(apds_shard_create(&shard, rank, world, APDS_TRANSPORT_LOOPBACK, &id, nullptr, rows.p, hi - lo, (uint32_t)lo));
        std::vector<int> counts = {nq, nq};
        for (int rep = 0; rep < 3; rep++) OK(apds_shard_knn(shard, dq.p, nq, counts.data(), k, dk.p, nullptr));
        std::vector<uint64_t> got((size_t)nq * k);
        OK(apds_dev_download(got.data(), dk.p, got.size() * 8, nullptr));
        CHECK(got == want[(size_t)rank], "lag case: rank %d keys differ", rank);
        OK(apds_shard_destroy(shard));   // rank 1 gets here while rank 0 still sleeps in front of its closing waits
        OK(apds_thread_release());
    };
    std::thread a(body, 0), b(body, 1);
    a.join();
    b.join();
    // a second attachment of a live rank is refused and must leave the first attachment usable
    {
        apds_comm_id id2;
        OK(apds_comm_id_create(APDS_TRANSPORT_LOOPBACK, &id2));
        OK(apds_set_device(0));
        void *first = nullptr, *second = nullptr;
        std::thread peer([&] {
            OK(apds_set_device(0));
            void* sh = nullptr;
            OK(apds_shard_create(&sh, 1, 2, APDS_TRANSPORT_LOOPBACK, &id2, nullptr, nullptr, 0, 0));
            int c[2] = {-1, -1};
            OK(apds_shard_counts(sh, 5, c, nullptr));
            CHECK(c[0] == 3 && c[1] == 5, "double-attach case: peer counts %d %d", c[0], c[1]);
            OK(apds_shard_destroy(sh));
            OK(apds_thread_release());
        });
        OK(apds_shard_create(&first, 0, 2, APDS_TRANSPORT_LOOPBACK, &id2, nullptr, nullptr, 0, 0));
        const int rc = apds_shard_create(&second, 0, 2, APDS_TRANSPORT_LOOPBACK, &id2, nullptr, nullptr, 0, 0);
        CHECK(rc == -5 && !second, "a second attachment of rank 0 returned %d", rc);
        int c[2] = {-1, -1};
        OK(apds_shard_counts(first, 3, c, nullptr));   // the first attachment's post and events are intact
        CHECK(c[0] == 3 && c[1] == 5, "double-attach case: counts %d %d", c[0], c[1]);
        peer.join();
        OK(apds_shard_destroy(first));
    }
    printf("lag + double attach ... %s\n%d failed\n", failures.load() ? "FAILED" : "ok", failures.load());
    return failures.load() ? 1 : 0;
}

}  // namespace

// `shard_loopback_test fuzz <cases> <seed>`: random worlds (1 .. 6 ranks), row counts from one row up (shards without rows included), query
// counts around the message rounding (0, 1, 1023, 1024, 1025, ...) and k = 1 / 2, every form of rank_main against the single-device keys
int fuzz(int cases, uint64_t seed) {
    SplitMix g{seed};
    const int edge[] = {0, 1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 2049};
    for (int c = 0; c < cases && !failures.load(); c++) {
        const int world = 1 + (int)(g.next() % 6);
        const int nt = (g.next() % 4 == 0) ? 1 + (int)(g.next() % 40) : 1 + (int)(g.next() % 30000);
        std::vector<int> nqs((size_t)world);
        for (int& n : nqs) n = (g.next() % 3 == 0) ? edge[g.next() % (sizeof(edge) / sizeof(edge[0]))] : (int)(g.next() % 3000);
        const int k = 1 + (int)(g.next() % 2);
        char label[160];
        int off = snprintf(label, sizeof(label), "fuzz %d: queries", c);
        for (int n : nqs) off += snprintf(label + off, sizeof(label) - (size_t)off, " %d", n);
        run_world(world, APDS_TRANSPORT_LOOPBACK, nt, nqs, k, label);
    }
    printf("%d failed\n", failures.load());
    return failures.load() ? 1 : 0;
}

int main(int argc, char** argv) {
    if (apds_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    if (argc >= 2 && std::string(argv[1]) == "lag") return lag_case();
    if (argc >= 3 && std::string(argv[1]) == "fuzz") return fuzz(atoi(argv[2]), argc >= 4 ? strtoull(argv[3], nullptr, 10) : 1);
    run_world(2, APDS_TRANSPORT_LOOPBACK, 6001, {700, 1300}, 2, "loopback");
    run_world(4, APDS_TRANSPORT_LOOPBACK, 40003, {1500, 0, 2300, 37}, 2, "loopback");
    run_world(3, APDS_TRANSPORT_LOOPBACK, 5000, {64, 65, 1}, 1, "loopback");
    run_world(3, APDS_TRANSPORT_LOOPBACK, 20011, {300, 0, 77}, 24, "loopback, k above 16 (paged scan, any-k merge)");
    run_world(2, APDS_TRANSPORT_LOOPBACK, 9001, {130, 200}, 3, "loopback, k = 3");
    run_world(1, APDS_TRANSPORT_RCCL, 9000, {1100}, 2, "rccl (world 1: communicator, all-gather, send/recv group)");
    table_shard_case();
    int ver = 0;
    {   // the RCCL build this ran on
        apds_comm_id id;
        OK(apds_comm_id_create(APDS_TRANSPORT_LOOPBACK, &id));
        void* shard = nullptr;
        OK(apds_set_device(0));
        OK(apds_shard_create(&shard, 0, 1, APDS_TRANSPORT_LOOPBACK, &id, nullptr, nullptr, 0, 0));
        OK(apds_shard_info(shard, nullptr, nullptr, nullptr, nullptr, nullptr, &ver));
        OK(apds_shard_destroy(shard));
    }
    printf("rccl version code %d\n", ver);
    printf("%d failed\n", failures.load());
    return failures.load() ? 1 : 0;
}
