// CPU build of the call merger behind vdb_search (vectordb_amd/csrc/host_merge.h) under
// ThreadSanitizer: 64 threads, three keys, a stand-in search that answers every request of a batch.
// Exit code 0 = every caller got its own answer and the sanitizer saw no race.
#include "../../vectordb_amd/csrc/host_merge.h"
#include <atomic>
#include <chrono>
#include <cstdio>
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
    int tables[3] = {0, 1, 2};
    const int nthreads = 64, rounds = 200;
    auto worker = [&](int t) {
        for (int r = 0; r < rounds; ++r) {
            Req req;
            const int which = (t + r) % 3;
            req.key = vdb::MergeKey(&tables[which], 10);
            req.weight = 1 + (t % 3);
            req.query = (long long)t * 100000 + r;
            merger.submit(req, 16, [&](std::vector<Req*>& batch) {
                // one batch per key at a time
                if (in_flight_same_key[which].fetch_add(1) != 0) ++wrong;
                int64_t w = 0;
                for (Req* b : batch) w += b->weight;
                if (w > 16 && batch.size() > 1) ++wrong;
                std::this_thread::sleep_for(std::chrono::microseconds(30));
                for (Req* b : batch) { b->answer = b->query * 7 + 1; b->rc = 0; }
                ++batches;
                merged += (long long)batch.size();
                in_flight_same_key[which].fetch_sub(1);
            });
            if (req.rc != 0 || req.answer != req.query * 7 + 1) ++wrong;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    std::printf("requests %d batches %lld (%.2f requests per batch) wrong %lld\n", nthreads * rounds, batches.load(),
                (double)merged.load() / (double)batches.load(), wrong.load());
    return wrong.load() == 0 && merged.load() == (long long)nthreads * rounds ? 0 : 1;
}
