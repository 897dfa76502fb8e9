// handoff_tsan.cpp -- the product's cross-thread hand-off primitives (libzl_amd/csrc/zl_handoff.h) under ThreadSanitizer:
// four caller threads post requests while one "cycle" thread drains them; two setter threads (serialised by a mutex the reader
// never takes) publish parameter snapshots while the cycle thread takes them.  Checks: every request arrives exactly once, per
// producer in order; every snapshot taken is internally consistent (all words of one publication); the last publication is seen.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I libzl_amd/csrc -I include tests/cpu_harness/handoff_tsan.cpp -lpthread
#include <atomic>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "zl_handoff.h"

struct Req { int32_t producer, seq; uint64_t payload[6]; };
struct Params { uint32_t w[66]; };          // the size class of zlhip_clip_params' head

int main()
{
    static ZlRequestQueue<Req, 256> q;
    static ZlSnapshot<Params> snap;
    constexpr int NP = 4, PER = 20000, PUBS = 20000;
    std::atomic<bool> go{false}, producersDone{false}, settersDone{false};
    std::atomic<long> dropped{0};
    std::vector<std::thread> th;
    for (int p = 0; p < NP; ++p)
        th.emplace_back([&, p] {
            while (!go.load()) { }
            for (int i = 0; i < PER; ++i) {
                Req r; r.producer = p; r.seq = i;
                for (auto &x : r.payload) x = (uint64_t)p * 1000003u + (uint64_t)i;
                while (!q.push(r)) std::this_thread::yield();          // full: the cycle has not drained yet
            }
        });
    std::mutex setMu;
    std::atomic<uint32_t> lastPublished{0};
    for (int s = 0; s < 2; ++s)
        th.emplace_back([&, s] {
            while (!go.load()) { }
            for (int i = 0; i < PUBS; ++i) {
                std::lock_guard<std::mutex> lk(setMu);
                const uint32_t v = lastPublished.load(std::memory_order_relaxed) + 1;
                Params p; for (auto &w : p.w) w = v;
                snap.publish(p);
                lastPublished.store(v, std::memory_order_relaxed);
            }
            (void)s;
        });
    long got = 0, torn = 0, order = 0, taken = 0;
    uint32_t lastSeen = 0;
    std::thread cycle([&] {
        int next[NP] = {0, 0, 0, 0};
        while (!go.load()) { }
        for (;;) {
            const bool pd = producersDone.load(), sd = settersDone.load();
            Req r;
            while (q.pop(r)) {
                ++got;
                if (r.seq != next[r.producer]) ++order;
                next[r.producer] = r.seq + 1;
                for (auto x : r.payload) if (x != (uint64_t)r.producer * 1000003u + (uint64_t)r.seq) ++torn;
            }
            Params p;
            if (snap.take(p)) {
                ++taken;
                for (auto w : p.w) if (w != p.w[0]) ++torn;
                if (p.w[0] < lastSeen) ++order;
                lastSeen = p.w[0];
            }
            if (pd && sd && got == (long)NP * PER && !snap.dirty.load()) break;
        }
    });
    go.store(true);
    for (int i = 0; i < NP; ++i) th[(size_t)i].join();
    producersDone.store(true);
    for (size_t i = NP; i < th.size(); ++i) th[i].join();
    settersDone.store(true);
    cycle.join();
    std::printf("requests %ld of %d, out of order %ld, torn %ld, snapshots taken %ld, last seen %u of %u, dropped %ld\n", got, NP * PER, order, torn, taken, lastSeen,
                lastPublished.load(), dropped.load());
    return (got == (long)NP * PER && order == 0 && torn == 0 && lastSeen == lastPublished.load()) ? 0 : 1;
}
