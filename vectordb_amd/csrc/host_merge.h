// Merging of concurrent blocking calls (host side, no GPU code): the turn-taking behind vdb_search.
//
// The reference serves every HTTP request on its own thread against one engine handle
// (/root/reference/app.py:11-12, 186-190), so blocking searches arrive side by side.  Calls that can
// share one batched search (same table, same k) are merged: the first caller of a key runs at once;
// callers of the same key that arrive while it runs wait, and the next one to get the turn takes
// everything waiting for that key along with it (up to a weight limit).  Calls with different keys
// never wait for each other.  There is no waiting window: a lone caller is never delayed.
//
// Kept free of HIP so that a CPU build of exactly this code runs under ThreadSanitizer
// (tests/native/merge_tsan.cpp).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace vdb {

// (table generation id, k).  The id is handed out once per created table and never reused, so that a
// table dropped and created again under the same name (or at the same address) while callers wait is
// another key: its callers are not merged with those of the table that is gone.
typedef std::pair<uint64_t, int> MergeKey;

// What a request must offer: `MergeKey key; int64_t weight; bool done;` (done is guarded by the
// merger's mutex; results are handed over by the leader before it sets done).
template <class Req>
class CallMerger {
public:
    // Blocks until the request has been answered, by this thread (as leader of a batch) or by another
    // caller's batch.  `run(batch)` answers every request of the batch; it runs without the lock held.
    // If `run` throws (std::bad_alloc of a staging vector, say), `on_throw(batch)` marks every request
    // of the batch as failed, the key is released and the followers are woken as after a normal run;
    // the exception does not leave submit() (the callers sit behind a C ABI).
    template <class Run>
    void submit(Req& req, int64_t max_weight, Run run) {
        submit(req, max_weight, run, [](std::vector<Req*>&) {});
    }
    template <class Run, class OnThrow>
    void submit(Req& req, int64_t max_weight, Run run, OnThrow on_throw) {
        std::vector<Req*> batch;
        {
            std::unique_lock<std::mutex> lk(mu_);
            pending_.push_back(&req);
            cv_.wait(lk, [&] { return req.done || busy_.count(req.key) == 0; });
            if (req.done) return;
            busy_.insert(req.key);                    // this caller's turn for its key
            int64_t total = req.weight;               // the leader always goes; others while they fit
            std::vector<Req*> rest;
            for (Req* r : pending_) {
                if (r == &req) batch.push_back(r);
                else if (r->key == req.key && total + r->weight <= max_weight) { batch.push_back(r); total += r->weight; }
                else rest.push_back(r);
            }
            pending_.swap(rest);
        }
        try {
            run(batch);
        } catch (...) {
            on_throw(batch);
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (Req* r : batch) if (r != &req) r->done = true;
            busy_.erase(req.key);
        }
        cv_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Req*> pending_;
    std::set<MergeKey> busy_;
};

}  // namespace vdb
