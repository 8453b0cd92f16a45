// CPU build of the call merger behind vdb_search (vectordb_amd/csrc/host_merge.h) under
// ThreadSanitizer: 64 threads, three keys, a stand-in search that answers every request of a batch.
// Every seventh batch of one key throws instead of answering: its callers must all come back with the
// failure code (none may hang, none may be answered by a later batch).
// Exit code 0 = every caller got its own answer (or the failure) and the sanitizer saw no race.
#include "../../vectordb_amd/csrc/host_merge.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <new>
#include <thread>

struct Req {
    vdb::MergeKey key;
    int64_t weight = 1;
    bool done = false;
    long long query = 0;
    long long answer = -1;     // written by the leader of the batch, read by the owner after submit()
    int rc = -1;
};

int main() {
    vdb::CallMerger<Req> merger;
    std::atomic<long long> batches{0}, merged{0}, wrong{0}, in_flight_same_key[3];
    for (auto& a : in_flight_same_key) a = 0;
    std::atomic<long long> thrown{0}, failed_seen{0};
    const int nthreads = 64, rounds = 200;
    auto worker = [&](int t) {
        for (int r = 0; r < rounds; ++r) {
            Req req;
            const int which = (t + r) % 3;
            req.key = vdb::MergeKey((uint64_t)(which + 1), 10);
            req.weight = 1 + (t % 3);
            req.query = (long long)t * 100000 + r;
            merger.submit(req, 16, [&](std::vector<Req*>& batch) {
                // one batch per key at a time
                if (in_flight_same_key[which].fetch_add(1) != 0) ++wrong;
                int64_t w = 0;
                for (Req* b : batch) w += b->weight;
                if (w > 16 && batch.size() > 1) ++wrong;
                std::this_thread::sleep_for(std::chrono::microseconds(30));
                ++batches;
                merged += (long long)batch.size();
                in_flight_same_key[which].fetch_sub(1);
                if (which == 2 && batches.load() % 7 == 0) { ++thrown; throw std::bad_alloc(); }
                for (Req* b : batch) { b->answer = b->query * 7 + 1; b->rc = 0; }
            }, [&](std::vector<Req*>& batch) {
                for (Req* b : batch) { b->rc = -6; b->answer = -1; }
            });
            if (req.rc == -6 && req.answer == -1) ++failed_seen;
            else if (req.rc != 0 || req.answer != req.query * 7 + 1) ++wrong;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    std::printf("requests %d batches %lld (%.2f requests per batch) wrong %lld, batches that threw %lld, callers that saw the failure %lld\n",
                nthreads * rounds, batches.load(), (double)merged.load() / (double)batches.load(), wrong.load(),
                thrown.load(), failed_seen.load());
    if (thrown.load() > 0 && failed_seen.load() < thrown.load()) return 1;
    return wrong.load() == 0 && merged.load() == (long long)nthreads * rounds ? 0 : 1;
}
