// C-ABI layer: table registry, HBM-resident store with device-side append, search orchestration.
// See include/vdb.h for the contract of every entry point.
#include "vdb_internal.h"
#include "host_merge.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <string>
#include <string_view>
#include <vector>
#include <sys/types.h>
#include <unistd.h>

namespace vdb {

// ------------------------------------------------------------------------------ error state
static thread_local std::string tl_error;

static int fail(int code, const std::string& msg) {
    tl_error = msg;
    return code;
}

#define VDB_HIP(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(VDB_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

// ------------------------------------------------------------------------------ global state
struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
};

struct Workspace {
    hipStream_t stream = nullptr;        // own stream for the host entry points
    hipStream_t last_stream = nullptr;   // stream the last call was enqueued on
    hipEvent_t  done = nullptr;          // recorded after the last call's work
    bool        used = false;
    bool        host_idle = false;       // released by a blocking call on the workspace's own stream: no event was recorded
    bool        fixed = false;           // buffers are carved from a fixed region: never (re)allocated
    DevBuf qprep, qinfo, cand_s, cand_t, need_exact;      // search scratch
    DevBuf small;                                         // batched path: cnt | lost | thr0 | thr1
    DevBuf cand_h;                                        // scan path: list heads
    DevBuf q16;                                           // batched path, 16-bit tables: query hi | lo parts
    DevBuf qimg;                                          // batched path: tiled 16-bit image of the queries (ping-pong kernel)
    DevBuf q_raw, o_ids, o_scores, o_counts;              // staging of the host entry point (o_ids holds ids | scores | counts)
    void*  h_pin = nullptr;                               // pinned host mirror: queries | ids | scores | counts
    char*  d_pin = nullptr;                               // the same memory as the device sees it
    size_t h_cap = 0;
    // Calls of a few queries skip both copies: the kernels read the queries from the pinned host block and write
    // the answers into it (3 KB and 124 bytes over the bus instead of two copy operations in the stream), and the
    // finalize step leaves one word per query there that says whether the query still needs the exact pass.
    int*   host_flag = nullptr;
    bool   host_flag_used = false;       // the search wrote the words (a search over the mirror, with its second tier, does not)
    // Blocking host calls: the lists of flagged queries live inside the block the caller copies back anyway
    // (ne_ext), and the exact pass is launched by the caller only when the copied-back list holds a query
    // (defer_exact): two early-exit launches less per call on the path whose cost is all fixed cost.
    int*   ne_ext = nullptr;
    bool   defer_exact = false;
    struct Deferred {
        bool pending = false;
        TableView tv; const float* qprep = nullptr; int64_t nq = 0; int dpad = 0; SelGeom g{}; int Gxe = 0; int k = 0;
        float* cs = nullptr; long long* ct = nullptr; long long* out_ids = nullptr; float* out_scores = nullptr;
        int* out_counts = nullptr; int* flagged = nullptr;
    } deferred;
};

// A device array that grows in place: a virtual address range is reserved once for the largest size
// the table can reach and backed by physical memory chunk by chunk (hipMemCreate / hipMemMap), so
// that growth never copies or moves a row and never waits for searches in flight.  Where the
// virtual-memory API is not available the array is a plain allocation that grows by allocate + copy.
struct VArray {
    char*  base = nullptr;
    size_t reserved = 0;                 // bytes of address space (0 = plain allocation)
    size_t mapped = 0;                   // bytes backed by memory
    size_t chunk = 0;                    // every chunk of a range has this size
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks;
};

struct Table {
    std::string name;
    uint64_t gen = 0;                    // never reused: what the call mergers key on (host_merge.h)
    uint64_t uuid = 0;                   // identity across snapshots: drawn at create, restored by load
    std::mutex snap_mu;                  // one save of this table at a time
    int64_t dim = 0;
    int metric = 0, dtype = 0;
    int m = 0;          // 16-byte units per row
    int dpad = 0;       // elements per padded row
    int64_t ld16 = 0;   // elements per row of the 16-bit image the batched pass multiplies (0 = none)
    std::atomic<int64_t> rows{0};        // published row count: searches see rows [0, rows)
    std::atomic<int64_t> cap{0};
    VArray a_rows, a_ids, a_n2, a_rows16;
    char*      d_rows = nullptr;
    long long* d_ids = nullptr;
    float*     d_n2 = nullptr;
    unsigned short* d_rows16 = nullptr;  // fp32 tables: bf16 mirror; 16-bit tables: alias of d_rows
    float*     d_maxn2 = nullptr;
    int*       h_bad = nullptr;          // pinned, device-visible: set by an append that met a non-finite row
    int*       d_bad = nullptr;          // device address of h_bad
    bool       vmm = false;
    // IP tables: norm statistics and the list of rows far longer than the rest (table_stats_update).
    // Updated under ins_mu before new rows are published; searches copy `pub` under stat_mu.
    char*         d_stat = nullptr;      // hist[256] u32 | n_out int | max_rest float | pad | two lists of OUTLIER_CAP rows
    int64_t       stat_rows = 0;         // rows the histogram and the list cover
    float         cand_limit = __builtin_huge_valf();     // 64 x the upper edge of the median bucket of n2
    float         limit_used = __builtin_huge_valf();     // limit the current list was built with (inf: no list)
    int           out_cur = 0;           // which of the two lists is current
    std::mutex    stat_mu;
    struct OutPub { float limit = __builtin_huge_valf(); const long long* list = nullptr; int n = 0; float max_rest = 0.f; } pub;
    std::shared_mutex mu;                // shared: searches and appends; exclusive: drop (and growth without VMM)
    std::mutex ins_mu;                   // appends to this table run one at a time
    bool dropped = false;

    bool mirrored() const { return dtype == VDB_F32 && ld16 > 0; }
    size_t bytes_per_row() const {
        return (size_t)m * 16 + sizeof(long long) + sizeof(float) + (mirrored() ? (size_t)ld16 * 2 : 0);
    }
};

struct HostReq;
struct Global {
    std::mutex mu;                       // registry, workspaces, options
    bool inited = false;
    int device = 0;
    int n_cu = 256;
    std::map<std::string, std::shared_ptr<Table>, std::less<>> tables;      // std::less<>: looked up by string_view, no temporary
    std::atomic<uint64_t> next_gen{1};
    std::vector<std::unique_ptr<Workspace>> free_ws;
    long long* d_counters = nullptr;
    // options
    int use_blas_threshold = 2;          // a batched pass over up to 128 queries costs about 1.2 scans: see search_core
    double cache_capacity_gb = 0.0;
    double insert_buffer_gb = 1.0;
    size_t table_bytes = 0;
    int bf16_mirror = 1;                 // fp32 tables keep a bf16 image of their rows for the batched pass
    int use_vmm = 1;                     // tables grow in reserved virtual ranges (0: allocate + copy)
    size_t vmm_gran = 0;                 // allocation granularity of the virtual-memory API (0 = unavailable)
    int tune_pp = 1;                     // batches over 128 queries run on the ping-pong kernel
    int force_exact = 0;                 // measurement: every query takes the exact pass
    int tune_pp_parts = 1;               // the last batched pass tightens its threshold after an eighth of the rows
    int mirror_tier2 = 1;                // queries the mirror scan cannot certify are re-scanned over the fp32 rows
    int scan_mirror = 0;                 // single queries over fp32 tables scan the bf16 mirror
    int row_bounds = 1;                  // L2 keys carry their own row's error term (0: table-wide norm bound)
    int tune_nt = 1, tune_waves = 8, tune_bpc = 4, tune_contig = 1, tune_dyn = 1, tune_gemm = 1, tune_split_f16 = 0, tune_split_bf16 = 0, tune_sample = 1, tune_gemm32s = 2;
    // host-insert staging (guarded by insert_mu)
    std::mutex insert_mu;
    DevBuf stage_rows, stage_ids;
    hipStream_t insert_stream = nullptr;
    // profiling
    std::atomic<long long> batched_queries{0};
    std::atomic<int> prof_on{0};         // 0 off, n: every n-th launch is bracketed by events
    std::atomic<unsigned> prof_seq{0};
    std::mutex prof_mu;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    // order of table scans issued on different streams (guarded by chain_mu)
    std::mutex chain_mu;
    int scan_serialize = 1;
    hipEvent_t chain_ev[16] = {};
    int chain_head = -1;
    hipStream_t chain_last = nullptr;
    // merging of concurrent blocking host searches (see vdb_search and host_merge.h)
    int coalesce_host = 1;
    int host_direct = 1;                 // blocking calls of up to four queries read and write the pinned host block directly
    std::atomic<int> debug_fail_map{0};  // test hook, see varray_map
    long long mirrors_dropped = 0;
    int ip_outliers = 1;                 // IP tables keep rows far longer than the rest on a side list (table_stats_update)       // tables whose bf16 mirror made room for rows (table_drop_mirror)
};
static Global G;

// Scratch memory of vdb_search_dev_noalloc: part of the code object, so that the entry point itself
// never allocates.  The array's name is a host-side stand-in; its device address is looked up once
// per device at vdb_init (hipGetSymbolAddress) and only that pointer is handed to kernels.
static const size_t NOALLOC_BYTES = (size_t)24 << 20;
static __device__ __align__(256) unsigned char vdb_noalloc_scratch[NOALLOC_BYTES];
static const int NOALLOC_MAX_DEV = 64;
static void* g_noalloc_ptr[NOALLOC_MAX_DEV] = {};
static std::atomic<int> g_noalloc_busy{0};      // one host call at a time works on the array

static size_t round_up(size_t v, size_t g) { return (v + g - 1) / g * g; }

static int enter() {
    if (!G.inited) return fail(VDB_ESTATE, "vdb_init has not been called");
    hipError_t e = hipSetDevice(G.device);
    if (e != hipSuccess) return fail(VDB_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return VDB_OK;
}

static int ensure(DevBuf& b, size_t bytes, hipStream_t busy_on, bool fixed = false) {
    if (bytes <= b.cap) return VDB_OK;
    if (fixed) return fail(VDB_EINVAL, "the search needs more scratch memory than the no-allocation entry point holds");
    if (b.p) {
        if (busy_on) VDB_HIP(hipStreamSynchronize(busy_on));
        VDB_HIP(hipFree(b.p));
        b.p = nullptr; b.cap = 0;
    }
    size_t want = bytes + bytes / 4;
    if (want < 4096) want = 4096;
    VDB_HIP(hipMalloc(&b.p, want));
    b.cap = want;
    return VDB_OK;
}

static std::shared_ptr<Table> find_table(const char* name) {
    std::lock_guard<std::mutex> lk(G.mu);
    auto it = G.tables.find(std::string_view(name));
    return it == G.tables.end() ? nullptr : it->second;
}

// A workspace may be reused at once by work on the stream it last served (stream order keeps the
// two uses apart); otherwise only after its last work has completed.
static int acquire_ws(hipStream_t st, std::unique_ptr<Workspace>& out, bool own_stream = false) {
    {
        std::lock_guard<std::mutex> lk(G.mu);
        int pick = -1;
        for (size_t i = 0; i < G.free_ws.size(); ++i) {
            Workspace* w = G.free_ws[i].get();
            // unused, or last used on the stream the new work goes to (the blocking entry points work on the
            // workspace's own stream): stream order is all the new call needs
            if (!w->used || (st != nullptr && w->last_stream == st) || (own_stream && w->last_stream == w->stream)) { pick = (int)i; break; }
        }
        if (pick < 0) {
            for (size_t i = 0; i < G.free_ws.size(); ++i) {
                Workspace* w = G.free_ws[i].get();
                const hipError_t q = w->host_idle ? hipStreamQuery(w->stream) : hipEventQuery(w->done);
                if (q == hipSuccess) { pick = (int)i; break; }
            }
        }
        if (pick >= 0) {
            out = std::move(G.free_ws[pick]);
            G.free_ws.erase(G.free_ws.begin() + pick);
            return VDB_OK;
        }
    }
    std::unique_ptr<Workspace> w(new Workspace());
    VDB_HIP(hipStreamCreate(&w->stream));
    VDB_HIP(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
    out = std::move(w);
    return VDB_OK;
}

static void release_ws(std::unique_ptr<Workspace>& w, hipStream_t st) {
    w->used = true;
    w->last_stream = st;
    // a blocking call has waited for its own stream (or seen its last store): no event, the stream itself says
    // whether anything is left on it
    w->host_idle = st == w->stream;
    if (!w->host_idle) (void)hipEventRecord(w->done, st);
    std::lock_guard<std::mutex> lk(G.mu);
    G.free_ws.push_back(std::move(w));
}

SelGeom sel_geom(int k) {
    SelGeom g;
    int margin = k / 8;
    if (margin < 22) margin = 22;
    g.K = ((k + margin + 31) / 32) * 32;
    int need = 2 * g.K > g.K + 64 ? 2 * g.K : g.K + 64;
    g.C = 64;
    while (g.C < need) g.C <<= 1;
    int w = 4096 / g.C;          // W*C*12 bytes of LDS per block stays within 48 KiB
    if (w > 16) w = 16;
    if (w < 1) w = 1;
    g.W = w;
    return g;
}

// Lists of the scan itself: a scanning wave never merges two lists (it hands its own over), so its list
// needs room for K entries and one wave step of appends, not for two lists as the merges of the exact
// pass do.  Same K; from K = 160 on a smaller list and more waves per workgroup (k = 256: 8 waves instead of 4,
// k = 1000: 2 instead of 1).  The finalize step copes with the larger number of lists per query (fin_pth).
static SelGeom scan_geom(const SelGeom& g, int max_waves) {
    SelGeom s = g;
    s.C = 64;
    while (s.C < g.K + 64) s.C <<= 1;
    if (s.C > g.C) s.C = g.C;
    int w = 4096 / s.C;
    if (w > 16) w = 16;
    if (w < 1) w = 1;
    if (w < 2) return g;           // one wave per workgroup either way: nothing to gain
    if (max_waves > 0 && max_waves < w) w = max_waves;
    s.W = w;
    return s;
}

// ------------------------------------------------------------------------------ tiny kernels
__global__ void fill_empty_kernel(long long* ids, float* scores, int* counts, long long nq, int k, int l2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq * k) { ids[i] = -1; scores[i] = l2 ? __builtin_huge_valf() : -__builtin_huge_valf(); }
    if (i < nq) counts[i] = 0;
}

// ------------------------------------------------------------------------------ search core
struct ProfScope {
    // brackets one kernel launch with HIP events on its stream when profiling is enabled
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    bool on = false;
    int begin(hipStream_t st) {
        const int every = G.prof_on.load();
        on = every > 0 && (every == 1 || G.prof_seq.fetch_add(1) % (unsigned)every == 0);
        if (!on) return VDB_OK;
        {
            std::lock_guard<std::mutex> lk(G.prof_mu);
            if (!G.prof_pool.empty()) { ev = G.prof_pool.back(); G.prof_pool.pop_back(); }
        }
        if (!ev.first) {
            VDB_HIP(hipEventCreate(&ev.first));
            VDB_HIP(hipEventCreate(&ev.second));
        }
        VDB_HIP(hipEventRecord(ev.first, st));
        return VDB_OK;
    }
    int end(hipStream_t st) {
        if (!on) return VDB_OK;
        VDB_HIP(hipEventRecord(ev.second, st));
        std::lock_guard<std::mutex> lk(G.prof_mu);
        G.prof_events.push_back(ev);
        return VDB_OK;
    }
};

static const int GEMM_CAP = 8192;          // candidate slots per query on the batched path
static const double GEMM_TARGET = 2048.0;  // candidates per query each pass aims for

// Per-query path: one bandwidth-bound scan of the table per query.
static int search_scan_path(const TableView& tv_in, const SelGeom& g, int dpad, int64_t nq, const float* q_dev,
                            int k, long long* out_ids, float* out_scores, int* out_counts,
                            hipStream_t st, Workspace& ws) {
    TableView tv = tv_in;
    const SelGeom gs = scan_geom(g, G.tune_waves);        // the scan's lists; g stays the geometry of the exact pass
    // What the scan reads.  fp32 tables can be scanned on their bf16 mirror (option scan_mirror): half
    // the bytes; the rounding of the rows (2^-9 of ||x|| ||q||) goes into the certificate's bound, and
    // the candidates are re-scored from the fp32 rows as always, so the answer is the same or the
    // query is flagged: for a second scan over the fp32 rows below, and for the exact pass if that
    // cannot certify it either.  Needs a mirror row as long as the prepared query (dim % 32 == 0).
    TableView tsc = tv;
    // The list must reach well past the k-th hit for the wider bound to certify (K = 32 for k = 10 does; K = 128 for
    // k = 100 sent two queries in three to the exact pass).
    const bool mirror = G.scan_mirror != 0 && !ws.fixed && tv.dt == VDB_F32 && tv.rows16 != nullptr && tv.ld16 == dpad &&
                        g.K >= 3 * k;
    if (mirror) { tsc.rows = (const char*)tv.rows16; tsc.m = (int)(tv.ld16 / 8); tsc.dt = VDB_BF16; }
    const int Gx1 = scan_grid(tsc, gs, G.n_cu);
    const int Gx2 = mirror ? scan_grid(tv, gs, G.n_cu) : Gx1;     // second tier over the fp32 rows
    const int Gxe = scan_grid(tv, g, G.n_cu);                     // grid of the exact pass (one K-list per workgroup)
    // lists per query the buffers hold: one per scanning wave, or one per workgroup of the exact pass
    const size_t lists = (size_t)(Gx1 > Gx2 ? Gx1 : Gx2) * gs.W > (size_t)Gxe ? (size_t)(Gx1 > Gx2 ? Gx1 : Gx2) * gs.W : (size_t)Gxe;
    const int64_t chunk = 256;
    const int64_t nqc = nq < chunk ? nq : chunk;
    int rc;
    if ((rc = ensure(ws.qprep, (size_t)nqc * dpad * sizeof(float), ws.last_stream, ws.fixed))) return rc;
    if ((rc = ensure(ws.qinfo, (size_t)nqc * sizeof(QInfo), ws.last_stream, ws.fixed))) return rc;
    if ((rc = ensure(ws.cand_s, (size_t)nqc * lists * g.K * sizeof(float), ws.last_stream, ws.fixed))) return rc;
    if ((rc = ensure(ws.cand_t, (size_t)nqc * lists * g.K * sizeof(long long), ws.last_stream, ws.fixed))) return rc;
    if ((rc = ensure(ws.cand_h, (size_t)nqc * lists * sizeof(float), ws.last_stream, ws.fixed))) return rc;
    if (!ws.ne_ext && (rc = ensure(ws.need_exact, (size_t)(2 * nqc + 2) * sizeof(int), ws.last_stream, ws.fixed))) return rc;
    // Many short rows and a long candidate list: every wave would spend its first K*ln(rows/K)
    // rows filling and sorting its own list.  A scan of a small sample of the rows (1/256) first
    // gives the K-th best key of the sample, a lower bound of the K-th best key of the table; the
    // full scan starts with it as a floor, so that only about 256*K rows in total pass the compare.
    // bf16 rows are scanned against the query as packed bf16 hi + lo parts (two products per element,
    // the split leaves 2^-18 of |q| unrepresented): both are part of the certificate's bound
    FlatBound fb;
    if (tsc.dt == VDB_BF16) { fb.dd_mult = 2.0; fb.extra_rel = 4.0e-6; }
    if (mirror) fb.extra_rel += 1.96e-3;
    fb.force_exact = G.force_exact;
    const size_t rowbytes = (size_t)tsc.m * 16;
    const bool sampled = g.K >= 64 && rowbytes <= (size_t)(G.tune_sample > 1 ? G.tune_sample : 4096) && tv.n >= 262144 && G.tune_sample != 0;
    TableView tvs = tsc;
    int Gs = 0;
    if (sampled) {
        tvs.n = tv.n / 256 > 4LL * g.K ? tv.n / 256 : 4LL * g.K;
        Gs = scan_grid(tvs, gs, G.n_cu);
        if (Gs > G.n_cu) Gs = G.n_cu;                                 // one block per CU is plenty for 0.4 % of the rows
        if ((rc = ensure(ws.small, (size_t)nqc * sizeof(float), ws.last_stream, ws.fixed))) return rc;
    }
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t nqi = nq - q0 < chunk ? nq - q0 : chunk;
        float* qprep = (float*)ws.qprep.p;
        QInfo* qinfo = (QInfo*)ws.qinfo.p;
        float* cs = (float*)ws.cand_s.p;
        long long* ct = (long long*)ws.cand_t.p;
        int* ne = ws.ne_ext ? ws.ne_ext : (int*)ws.need_exact.p;
        // L2 keys carry the error term of their own row (see RowBound)
        if (tv.metric == VDB_METRIC_L2 && G.row_bounds) tv.rb = tsc.rb = tvs.rb = l2_row_bound(fb, tv.d, qinfo, (int)nqi);
        VDB_HIP(launch_prep_query(q_dev + (size_t)q0 * tv.d, nqi, tv.d, dpad, tv.metric, qprep, qinfo, ne, st));
        const float* floor = nullptr;
        if (sampled) {
            VDB_HIP(launch_scan(tvs, qprep, nqi, dpad, gs, Gs, cs, ct, (float*)ws.cand_h.p, nullptr, nullptr, st));
            VDB_HIP(launch_lists_floor(gs, Gs * gs.W, nqi, cs, ct, (const float*)ws.cand_h.p, (float*)ws.small.p, st));
            floor = (const float*)ws.small.p;
        }
        {
            // A scan saturates HBM on its own, so scans issued on different streams run one after
            // the other (two at once would only evict each other's cache lines), while the short
            // kernels before and after a scan overlap with the scans of the other streams.
            std::unique_lock<std::mutex> chain(G.chain_mu);
            // (the no-allocation entry point creates no events: its scans are neither chained nor timed)
            // (nor are scans of a few megabytes: they come from the caches, and an event per call is most of their cost)
            const bool ordered = G.scan_serialize != 0 && !ws.fixed && (double)tsc.n * tsc.m * 16.0 >= 256e6;
            if (!ordered) chain.unlock();
            if (ordered) {
                if (G.chain_head >= 0 && G.chain_last != st)
                    VDB_HIP(hipStreamWaitEvent(st, G.chain_ev[G.chain_head], 0));
            }
            ProfScope prof;
            if (!ws.fixed && (rc = prof.begin(st))) return rc;
            VDB_HIP(launch_scan(tsc, qprep, nqi, dpad, gs, Gx1, cs, ct, (float*)ws.cand_h.p, floor, nullptr, st));
            if (!ws.fixed && (rc = prof.end(st))) return rc;
            if (ordered) {
                const int slot = (G.chain_head + 1) % 16;
                if (!G.chain_ev[slot]) VDB_HIP(hipEventCreateWithFlags(&G.chain_ev[slot], hipEventDisableTiming));
                VDB_HIP(hipEventRecord(G.chain_ev[slot], st));
                G.chain_head = slot;
                G.chain_last = st;
            }
        }
        VDB_HIP(launch_finalize(tv, qprep, qinfo, nqi, dpad, gs, Gx1 * gs.W, k, cs, ct, (const float*)ws.cand_h.p,
                                out_ids + (size_t)q0 * k, out_scores + (size_t)q0 * k, out_counts + q0,
                                ne, G.d_counters, fb, nullptr, st, mirror ? nullptr : ws.host_flag));
        if (!mirror && ws.host_flag) ws.host_flag_used = true;
        int* flagged = ne;
        if (mirror && G.mirror_tier2) {
            // second tier: the queries the mirror could not certify are scanned again over the fp32 rows (a
            // launch whose other workgroups leave at once), with the narrow bound; what this cannot certify
            // either goes on to the exact pass
            int* ne2 = ne + 1 + nqi;
            FlatBound fb2;
            fb2.force_exact = G.force_exact;
            if (tv.metric == VDB_METRIC_L2 && G.row_bounds) tv.rb = l2_row_bound(fb2, tv.d, qinfo, (int)nqi);
            VDB_HIP(launch_scan(tv, qprep, nqi, dpad, gs, Gx2, cs, ct, (float*)ws.cand_h.p, nullptr, ne, st));
            VDB_HIP(launch_finalize(tv, qprep, qinfo, nqi, dpad, gs, Gx2 * gs.W, k, cs, ct, (const float*)ws.cand_h.p,
                                    out_ids + (size_t)q0 * k, out_scores + (size_t)q0 * k, out_counts + q0,
                                    ne2, G.d_counters, fb2, ne, st));
            flagged = ne2;
        }
        if (ws.defer_exact && nq <= chunk) {
            Workspace::Deferred& df = ws.deferred;
            df.pending = true; df.tv = tv; df.qprep = qprep; df.nq = nqi; df.dpad = dpad; df.g = g; df.Gxe = Gxe; df.k = k;
            df.cs = cs; df.ct = ct; df.out_ids = out_ids; df.out_scores = out_scores; df.out_counts = out_counts; df.flagged = flagged;
            continue;
        }
        VDB_HIP(launch_exact(tv, qprep, nqi, dpad, g, Gxe, k, cs, ct,
                             out_ids + (size_t)q0 * k, out_scores + (size_t)q0 * k, out_counts + q0,
                             flagged, G.d_counters, st));
    }
    return VDB_OK;
}

// Batched path: MFMA GEMM with a thresholded epilogue.  Thresholds are bootstrapped by smaller
// passes (all scores of the first rows, then strided samples of the row tiles, each larger than
// the last); the full pass runs on the 16-bit MFMA for every table type: bf16 / fp16 rows with the
// query split into hi + lo parts, fp32 rows split into bf16 hi + lo as well (from 64 queries; the
// fp32 MFMA kernel serves smaller batches and the sampled passes).
static int search_gemm_path(const TableView& tv_in, const SelGeom& g, int dpad, int64_t nq, const float* q_dev,
                            int k, long long* out_ids, float* out_scores, int* out_counts,
                            hipStream_t st, Workspace& ws) {
    TableView tv = tv_in;
    const int K = g.K;
    const bool l2 = tv.metric == VDB_METRIC_L2;
    const long long ntiles = (tv.n + 127) / 128;
    // Sample sizes of the threshold passes, in row tiles, smallest first.  Each pass is about
    // GEMM_TARGET / K times larger than the one before, so that the K-th best key of one pass,
    // used as the threshold of the next, lets about GEMM_TARGET candidates per query through.
    // The first pass keeps every score of the first n0 <= GEMM_CAP rows.
    const double ratio = GEMM_TARGET / (double)K;
    long long n0 = tv.n < GEMM_CAP ? tv.n : GEMM_CAP;
    std::vector<long long> level_tiles;              // intermediate passes (strided samples), then all tiles
    {
        std::vector<long long> rev;
        long long tcur = ntiles;
        while (true) {
            long long tprev = (long long)std::ceil((double)tcur / ratio);
            if (tprev < 8) tprev = 8;
            if (tprev * 128 <= n0 || tprev >= tcur) break;
            rev.push_back(tprev);
            tcur = tprev;
        }
        for (size_t i = rev.size(); i > 0; --i) level_tiles.push_back(rev[i - 1]);
    }
    TableView tv0 = tv;
    tv0.n = n0;
    const int Gx = scan_grid(tv, g, G.n_cu);
    const int64_t chunk = 1024;
    const int64_t nqc = nq < chunk ? nq : chunk;
    size_t slots = (size_t)nqc * GEMM_CAP;
    if ((size_t)nqc * Gx * K > slots) slots = (size_t)nqc * Gx * K;
    const bool half = tv.dt != VDB_F32;
    // 16-bit tables: the query rounded to the table's type leaves a relative error of 2^-9 (bf16) or
    // 2^-11 (fp16) of ||x|| ||q|| in the fast keys; the finalize step re-scores every candidate within
    // twice the error bound of the k-th best fast key, so a coarse pass only widens that band.  With
    // the split options the query goes in as hi + lo parts instead (two products, narrow band).
    const int split = tv.dt == VDB_BF16 ? (G.tune_split_bf16 ? 1 : 0) : (G.tune_split_f16 ? 1 : 0);
    // fp32 tables run on the 16-bit matrix unit as well: rows and queries rounded to bf16 (one
    // product per term, relative error 2^-8), or presented as bf16 hi + lo with three products per
    // term (relative error 3 * 2^-16) when gemm32_split is 1
    const bool s32 = !half && G.tune_gemm32s != 0;
    const int s32split = G.tune_gemm32s == 1 ? 1 : 0;
    const int qstride = (dpad + 63) / 64 * 64;       // 16-bit query rows: zero padded to whole k-steps of the kernels
    // Batches of more than 128 queries run every pass but the first on the ping-pong kernel: 256 x 256
    // tiles, operands straight from global memory into LDS; fp32 tables multiply their bf16 mirror.
    const bool use_pp = G.tune_pp != 0 && nq > 128 && pp_eligible(tv, dpad) && ((half && !split) || (s32 && !s32split));
    // Smaller batches over fp32 tables read the mirror as well (as a bf16 table: half the bytes of a pass
    // that is bound by reading the rows once); it holds the very values the converting kernel would
    // round the fp32 rows to, so keys and bounds are the same.
    const bool use_m = s32 && !s32split && tv.rows16 != nullptr && G.bf16_mirror != 0;
    TableView tvm = tv;
    if (use_m) { tvm.rows = (const char*)tv.rows16; tvm.m = (int)(tv.ld16 / 8); tvm.dt = VDB_BF16; }
    FlatBound fb;
    fb.l2_bias = use_pp ? 1.0 : 0.0;
    fb.force_exact = G.force_exact;
    // accumulation chain in units of d: one per product, and a factor of two of margin on the
    // matrix unit's internal summation order and rounding
    fb.dd_mult = 2.0 * (s32 ? (s32split ? 3.0 : 1.0) : ((half && split) ? 2.0 : 1.0));
    fb.extra_rel = s32 ? (s32split ? 5.0e-5 : 3.92e-3)
                       : (!half ? 0.0 : (tv.dt == VDB_BF16 ? (split ? 1.6e-5 : 1.96e-3) : (split ? 2.5e-7 : 4.9e-4)));
    fb.extra_abs = tv.dt == VDB_F16 ? 3.0e-8 * std::sqrt((double)tv.d) : 0.0;      // fp16 subnormal steps
    int rc;
    if ((rc = ensure(ws.qprep, (size_t)nqc * dpad * sizeof(float), ws.last_stream))) return rc;
    if ((rc = ensure(ws.qinfo, (size_t)nqc * sizeof(QInfo), ws.last_stream))) return rc;
    if ((rc = ensure(ws.cand_s, slots * sizeof(float), ws.last_stream))) return rc;
    if ((rc = ensure(ws.cand_t, slots * sizeof(long long), ws.last_stream))) return rc;
    if ((rc = ensure(ws.need_exact, (size_t)(2 * nqc + 2) * sizeof(int), ws.last_stream))) return rc;
    if ((rc = ensure(ws.small, (size_t)nqc * 20, ws.last_stream))) return rc;
    if ((half || s32) && (rc = ensure(ws.q16, (size_t)nqc * qstride * 4, ws.last_stream))) return rc;
    if (use_pp && (rc = ensure(ws.qimg, pp_query_image_bytes((int)nqc, dpad), ws.last_stream))) return rc;
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int nqi = (int)(nq - q0 < chunk ? nq - q0 : chunk);
        float* qprep = (float*)ws.qprep.p;
        QInfo* qinfo = (QInfo*)ws.qinfo.p;
        float* cs = (float*)ws.cand_s.p;
        long long* ct = (long long*)ws.cand_t.p;
        int* ne = (int*)ws.need_exact.p;
        int* cnt = (int*)ws.small.p;
        int* lost = cnt + nqc;
        float* thr0 = (float*)(lost + nqc);
        float* thr1 = thr0 + nqc;
        int* bad = (int*)(thr1 + nqc);
        unsigned short* qhi = (unsigned short*)ws.q16.p;
        unsigned short* qlo = (half || s32) ? qhi + (size_t)nqc * qstride : nullptr;
        auto gemm = [&](const TableView& v, const float* thr, long long stride, long long tiles, int dense) -> hipError_t {
            if (half) return launch_gemm16_thresh(v, qhi, qlo, qstride, split, nqi, dpad, thr, stride, tiles, cs, ct, cnt, lost, GEMM_CAP, dense, G.n_cu, st);
            return launch_gemm_thresh(v, qprep, nqi, dpad, thr, 0, stride, tiles, cs, ct, cnt, lost, GEMM_CAP, dense, G.n_cu, st);
        };
        if (l2 && G.row_bounds) tv.rb = tv0.rb = tvm.rb = l2_row_bound(fb, tv.d, qinfo, nqi);
        VDB_HIP(launch_prep_query(q_dev + (size_t)q0 * tv.d, nqi, tv.d, dpad, tv.metric, qprep, qinfo, ne, st));
        VDB_HIP(hipMemsetAsync(bad, 0, (size_t)nqc * 4, st));
        if (half) VDB_HIP(launch_split_query16(tv.dt, qprep, nqi, dpad, qstride, qhi, qlo, bad, st));
        if (s32) VDB_HIP(launch_split_query16(VDB_BF16, qprep, nqi, dpad, qstride, qhi, qlo, bad, st));
        const bool pp = use_pp && nqi > 128;
        const unsigned short* qimg = (const unsigned short*)ws.qimg.p;
        if (pp) VDB_HIP(launch_tile_query16(half ? tv.dt : VDB_BF16, l2 ? 1 : 0, qprep, nqi, dpad, (unsigned short*)ws.qimg.p, bad, st));
        // pass 0: every score of the first n0 (<= GEMM_CAP) rows is kept, row r in slot r (no
        // counter traffic: a million atomics on nq addresses cost more than the whole pass);
        // thr0 = K-th best of them
        VDB_HIP(hipMemsetD32Async((hipDeviceptr_t)cnt, (int)n0, (size_t)nqc, st));
        VDB_HIP(hipMemsetAsync(lost, 0, (size_t)nqc * 4, st));
        VDB_HIP(hipMemsetD32Async((hipDeviceptr_t)thr1, (int)0xff800000, (size_t)nqc, st));      // -inf
        VDB_HIP(gemm(tv0, thr1, 1, (n0 + 127) / 128, 1));
        VDB_HIP(launch_kth_threshold(cs, GEMM_CAP, cnt, 0, GEMM_CAP, K, nullptr, thr0, nqi, st, &tv, nullptr));   // slot = row
        float* tau = thr0;
        float* spare = thr1;
        for (long long tiles : level_tiles) {
            // strided sample of `tiles` row tiles under the current threshold -> next threshold
            const long long stride = ntiles / tiles;
            VDB_HIP(hipMemsetAsync(cnt, 0, (size_t)nqc * 8, st));
            if (pp) VDB_HIP(launch_gemm_pp(tv, qimg, nqi, dpad, tau, stride, 0, -1, cs, ct, cnt, lost, GEMM_CAP, G.n_cu, st));
            else if (s32) VDB_HIP(launch_gemm32s(use_m ? tvm : tv, qhi, qlo, qstride, s32split, nqi, use_m ? (int)tv.ld16 : dpad, tau, stride, cs, ct, cnt, lost, GEMM_CAP, G.n_cu, st));
            else VDB_HIP(gemm(tv, tau, stride, ntiles / stride, 0));
            VDB_HIP(launch_kth_threshold(cs, GEMM_CAP, cnt, 0, GEMM_CAP, K, tau, spare, nqi, st, &tv, ct));
            std::swap(tau, spare);
        }
        // last pass: all row tiles
        VDB_HIP(hipMemsetAsync(cnt, 0, (size_t)nqc * 8, st));
        ProfScope prof;
        if ((rc = prof.begin(st))) return rc;
        if (pp) {
            // The threshold from the last sample lets about K * (rows / sample) rows through, and every
            // candidate costs the kernel about 230 cycles of one CU.  The first eighth of the tiles
            // therefore runs alone; the K-th best key of its candidates is a much tighter threshold for
            // the other seven eighths (candidates per query at 10 M rows: 2000 -> 500).  Candidates of both
            // parts stay in the array; the finalize step is told the tighter threshold, which is valid
            // for every row that is not a candidate.
            const long long tpp = (tv.n + pp_tile_rows(nqi) - 1) / pp_tile_rows(nqi);
            const long long part = G.tune_pp_parts ? tpp / 8 : 0;
            if (part >= 4LL * G.n_cu) {
                VDB_HIP(launch_gemm_pp(tv, qimg, nqi, dpad, tau, 1, 0, part, cs, ct, cnt, lost, GEMM_CAP, G.n_cu, st));
                VDB_HIP(launch_kth_threshold(cs, GEMM_CAP, cnt, 0, GEMM_CAP, K, tau, spare, nqi, st, &tv, ct));
                std::swap(tau, spare);
                VDB_HIP(launch_gemm_pp(tv, qimg, nqi, dpad, tau, 1, part, -1, cs, ct, cnt, lost, GEMM_CAP, G.n_cu, st));
            } else {
                VDB_HIP(launch_gemm_pp(tv, qimg, nqi, dpad, tau, 1, 0, -1, cs, ct, cnt, lost, GEMM_CAP, G.n_cu, st));
            }
        }
        else if (s32) VDB_HIP(launch_gemm32s(use_m ? tvm : tv, qhi, qlo, qstride, s32split, nqi, use_m ? (int)tv.ld16 : dpad, tau, 1, cs, ct, cnt, lost, GEMM_CAP, G.n_cu, st));
        else VDB_HIP(gemm(tv, tau, 1, ntiles, 0));
        if ((rc = prof.end(st))) return rc;
        VDB_HIP(launch_finalize_flat(tv, qprep, qinfo, nqi, dpad, g, k, cs, ct, cnt, GEMM_CAP, tau, lost, bad, fb,
                                     out_ids + (size_t)q0 * k, out_scores + (size_t)q0 * k, out_counts + q0,
                                     ne, G.d_counters, st));
        VDB_HIP(launch_exact(tv, qprep, nqi, dpad, g, Gx, k, cs, ct,
                             out_ids + (size_t)q0 * k, out_scores + (size_t)q0 * k, out_counts + q0,
                             ne, G.d_counters, st));
    }
    return VDB_OK;
}

static int search_core(Table& t, int64_t nq, const float* q_dev, int k, long long* out_ids,
                       float* out_scores, int* out_counts, hipStream_t st, Workspace& ws) {
    TableView tv;
    tv.rows = t.d_rows; tv.ids = t.d_ids; tv.n2 = t.d_n2; tv.max_n2 = t.d_maxn2;
    tv.rows16 = t.d_rows16; tv.ld16 = t.ld16;
    tv.n = t.rows.load(std::memory_order_acquire); tv.d = (int)t.dim; tv.m = t.m; tv.dt = t.dtype; tv.metric = t.metric;
    {
        // the list covers every row this search sees (it is brought up to date before rows are published); entries
        // of rows past tv.n are skipped by the kernels
        std::lock_guard<std::mutex> lk(t.stat_mu);
        if (t.pub.n > 0) { tv.out_limit = t.pub.limit; tv.outliers = t.pub.list; tv.n_out = t.pub.n; tv.max_rest = t.pub.max_rest; }
    }
    const int l2 = t.metric == VDB_METRIC_L2;
    if (tv.n == 0) {
        const long long tot = nq * k > nq ? nq * k : nq;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                           out_ids, out_scores, out_counts, (long long)nq, k, l2);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    SelGeom g = sel_geom(k);
    if (G.tune_waves > 0 && G.tune_waves < g.W) g.W = G.tune_waves;
    // engine_config.use_blas_threshold: batches at least this large take the GEMM path
    bool gemm = nq >= G.use_blas_threshold && tv.n >= 65536 && g.K <= 256 && G.tune_gemm != 0 && !ws.fixed;
    if (gemm) {
        // Small tables: a batched pass has a fixed cost (threshold passes, flat finalize) that a
        // few scans undercut.  Both estimates are fits of measured times (seconds; row bytes B):
        // a batched pass reads the rows once at about 6.1 TB/s and multiplies at about 5e14
        // multiply-adds per second on the 16-bit matrix unit (times the products per term).
        const double B = (double)tv.n * tv.m * 16.0;
        // bytes a batched pass reads: fp32 tables are multiplied from their bf16 mirror
        const bool mirrored = tv.dt == VDB_F32 && tv.rows16 != nullptr && G.bf16_mirror != 0 && G.tune_gemm32s == 2;
        const double Bp = mirrored ? (double)tv.n * tv.ld16 * 2.0 : B;
        const bool scan_m = G.scan_mirror != 0 && tv.dt == VDB_F32 && tv.rows16 != nullptr && tv.ld16 == t.dpad && g.K >= 3 * k;
        const double est_scan = 60e-6 + (double)nq * (20e-6 + (scan_m ? Bp : B) / 6.5e12);
        double est_gemm;
        if (tv.dt == VDB_F32 && G.tune_gemm32s == 0) {
            const double pass_rate = nq <= 32 ? 5.0e12 : nq <= 64 ? 4.2e12 : 3.8e12;      // fp32 MFMA kernel
            est_gemm = 270e-6 + (double)((nq + 127) / 128) * B / pass_rate;
        } else {
            const double products = tv.dt == VDB_F32 ? (G.tune_gemm32s == 1 ? 3.0 : 1.0)
                                                     : ((tv.dt == VDB_BF16 ? G.tune_split_bf16 : G.tune_split_f16) ? 2.0 : 1.0);
            est_gemm = 270e-6 + Bp / 6.1e12 + products * (double)tv.n * t.dpad * (double)nq / 5.0e14;
        }
        if (est_scan < est_gemm) gemm = false;
    }
    if (gemm) G.batched_queries += nq;
    if (gemm) return search_gemm_path(tv, g, t.dpad, nq, q_dev, k, out_ids, out_scores, out_counts, st, ws);
    return search_scan_path(tv, g, t.dpad, nq, q_dev, k, out_ids, out_scores, out_counts, st, ws);
}

// ------------------------------------------------------------------------------ table storage

static void varray_free(VArray& a) {
    if (a.reserved) {
        size_t off = 0;
        for (auto& c : a.chunks) {
            (void)hipMemUnmap(a.base + off, c.second);
            (void)hipMemRelease(c.first);
            off += c.second;
        }
        if (a.base) (void)hipMemAddressFree(a.base, a.reserved);
    } else if (a.base) {
        (void)hipFree(a.base);
    }
    a = VArray();
}

// Backs [mapped, want) of a reserved range, chunk by chunk.  All chunks of a range have one size:
// chunks of mixed sizes were refused now and then on the MI355X box (hipMemSetAccess: invalid
// argument, also for sizes and offsets that were multiples of 2 MiB and of each other), equal ones never.
static hipError_t varray_map(VArray& a, size_t want) {
    if (want > a.reserved) return hipErrorOutOfMemory;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = G.device;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = G.device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    while (a.mapped < want) {
        // test hook: the n-th chunk asked for from now on is refused as if memory had run out
        if (G.debug_fail_map > 0 && --G.debug_fail_map == 0) return hipErrorOutOfMemory;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, a.chunk, &prop, 0);
        if (e != hipSuccess) return e;
        e = hipMemMap(a.base + a.mapped, a.chunk, 0, h, 0);
        if (e != hipSuccess) { (void)hipMemRelease(h); return e; }
        e = hipMemSetAccess(a.base + a.mapped, a.chunk, &acc, 1);
        if (e != hipSuccess) { (void)hipMemUnmap(a.base + a.mapped, a.chunk); (void)hipMemRelease(h); return e; }
        a.chunks.emplace_back(h, a.chunk);
        a.mapped += a.chunk;
    }
    return hipSuccess;
}

// Largest row count a table is given address space for: what 320 GiB of rows hold (more than one
// GPU's HBM), so that a table never runs out of range before it runs out of memory.
static int64_t table_max_rows(const Table& t) {
    const size_t per = (size_t)t.m * 16 > (size_t)t.ld16 * 2 ? (size_t)t.m * 16 : (size_t)t.ld16 * 2;
    return (int64_t)(((size_t)320 << 30) / per);
}

static int table_reserve_ranges(Table& t) {
    // called once, before the first row arrives; failure leaves the table on plain allocations
    if (!G.use_vmm || G.vmm_gran == 0) return VDB_OK;
    const int64_t mr = table_max_rows(t);
    struct Want { VArray* a; size_t bytes; };
    Want w[4] = {{&t.a_rows, (size_t)mr * t.m * 16}, {&t.a_ids, (size_t)mr * 8}, {&t.a_n2, (size_t)mr * 4},
                 {&t.a_rows16, t.mirrored() ? (size_t)mr * t.ld16 * 2 : 0}};
    for (auto& x : w) {
        if (!x.bytes) continue;
        void* p = nullptr;
        // chunk: what 8192 rows take, as a power of two between 2 MiB and 1 GiB
        size_t chunk = G.vmm_gran;
        while (chunk < x.bytes / (size_t)mr * 8192 && chunk < ((size_t)1 << 30)) chunk <<= 1;
        x.a->chunk = chunk;
        const size_t sz = round_up(x.bytes + 256, chunk);
        if (hipMemAddressReserve(&p, sz, 0, nullptr, 0) != hipSuccess || !p) {
            (void)hipGetLastError();
            for (auto& y : w) varray_free(*y.a);
            return VDB_OK;                          // plain allocations instead
        }
        x.a->base = (char*)p;
        x.a->reserved = sz;
    }
    t.vmm = true;
    t.d_rows = t.a_rows.base; t.d_ids = (long long*)t.a_ids.base; t.d_n2 = (float*)t.a_n2.base;
    t.d_rows16 = t.mirrored() ? (unsigned short*)t.a_rows16.base : (t.dtype != VDB_F32 ? (unsigned short*)t.d_rows : nullptr);
    return VDB_OK;
}

static int budget_check(size_t extra) {
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.cache_capacity_gb > 0.0 && (double)(G.table_bytes + extra) > G.cache_capacity_gb * 1073741824.0)
        return fail(VDB_ENOMEM, "table storage would exceed gpu_resource_config.cache_capacity");
    return VDB_OK;
}

// Capacity for at least new_cap rows.  With reserved ranges (the normal case) nothing moves and
// searches in flight are not disturbed: the caller holds the table's shared lock and its append
// mutex.  Without them the arrays are reallocated and copied: the caller holds the exclusive lock.
static int table_alloc(Table& t, int64_t new_cap) {
    if (new_cap <= t.cap.load()) return VDB_OK;
    if (new_cap > table_max_rows(t)) return fail(VDB_ENOMEM, "table would exceed the address range reserved for it");
    if (t.vmm) {
        struct Part { VArray* a; size_t per_row; };
        Part parts[4] = {{&t.a_rows, (size_t)t.m * 16}, {&t.a_ids, 8}, {&t.a_n2, 4}, {&t.a_rows16, t.mirrored() ? (size_t)t.ld16 * 2 : 0}};
        size_t want[4], extra = 0;
        for (int i = 0; i < 4; ++i) {
            want[i] = parts[i].per_row ? round_up((size_t)new_cap * parts[i].per_row + 256, parts[i].a->chunk) : 0;
            if (want[i] > parts[i].a->reserved) want[i] = parts[i].a->reserved;
            if (want[i] > parts[i].a->mapped) extra += want[i] - parts[i].a->mapped;
        }
        int rc = budget_check(extra);
        if (rc) return rc;
        for (int i = 0; i < 4; ++i) {
            if (!parts[i].per_row) continue;
            const size_t before = parts[i].a->mapped;
            const hipError_t e = varray_map(*parts[i].a, want[i]);
            {
                // chunks mapped before a failure stay mapped (and are freed with the table): they count
                std::lock_guard<std::mutex> lk(G.mu);
                G.table_bytes += parts[i].a->mapped - before;
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(VDB_ENOMEM, std::string("mapping table memory: ") + hipGetErrorString(e));
            }
        }
        // rows that every mapped array holds
        int64_t cap = INT64_MAX;
        for (int i = 0; i < 4; ++i)
            if (parts[i].per_row) { const int64_t c = (int64_t)((parts[i].a->mapped - 256) / parts[i].per_row); if (c < cap) cap = c; }
        t.cap.store(cap);
        return VDB_OK;
    }
    // plain allocations: new arrays, device-to-device copy of the rows present
    const int64_t rows = t.rows.load();
    const size_t sz[4] = {(size_t)new_cap * t.m * 16 + 256, (size_t)new_cap * 8, (size_t)new_cap * 4,
                          t.mirrored() ? (size_t)new_cap * t.ld16 * 2 + 256 : 0};
    size_t need = 0, have = 0;
    VArray* arr[4] = {&t.a_rows, &t.a_ids, &t.a_n2, &t.a_rows16};
    for (int i = 0; i < 4; ++i) { need += sz[i]; have += arr[i]->mapped; }
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (G.cache_capacity_gb > 0.0 && (double)(G.table_bytes - have + need) > G.cache_capacity_gb * 1073741824.0)
            return fail(VDB_ENOMEM, "table storage would exceed gpu_resource_config.cache_capacity");
    }
    char* np[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) if (sz[i]) e = hipMalloc((void**)&np[i], sz[i]);
    // searches enqueued earlier on caller streams may still read the old arrays
    if (e == hipSuccess) e = hipDeviceSynchronize();
    const size_t used[4] = {(size_t)rows * t.m * 16, (size_t)rows * 8, (size_t)rows * 4, (size_t)rows * t.ld16 * 2};
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        if (sz[i] && rows > 0) e = hipMemcpy(np[i], arr[i]->base, used[i], hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (int i = 0; i < 4; ++i) if (np[i]) (void)hipFree(np[i]);
        return fail(VDB_ENOMEM, std::string("allocating table memory: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < 4; ++i) {
        if (arr[i]->base) (void)hipFree(arr[i]->base);
        arr[i]->base = np[i]; arr[i]->mapped = sz[i]; arr[i]->reserved = 0;
    }
    t.d_rows = t.a_rows.base; t.d_ids = (long long*)t.a_ids.base; t.d_n2 = (float*)t.a_n2.base;
    t.d_rows16 = t.mirrored() ? (unsigned short*)t.a_rows16.base : (t.dtype != VDB_F32 ? (unsigned short*)t.d_rows : nullptr);
    {
        std::lock_guard<std::mutex> lk(G.mu);
        G.table_bytes = G.table_bytes - have + need;
    }
    t.cap.store(new_cap);
    return VDB_OK;
}

static int table_grow_for(Table& t, int64_t extra) {
    const int64_t rows = t.rows.load(), cap = t.cap.load();
    if (rows + extra <= cap) return VDB_OK;
    int64_t nc = cap + (cap < ((int64_t)1 << 22) ? cap : cap / 4);      // doubles while small, then grows by a quarter
    if (nc < rows + extra) nc = rows + extra;
    if (nc < 1024) nc = 1024;
    const int64_t mr = table_max_rows(t);
    if (nc > mr) nc = mr;
    int rc = table_alloc(t, nc);
    if (rc == VDB_ENOMEM && nc > rows + extra) rc = table_alloc(t, rows + extra);    // exactly what is needed
    return rc;
}

// The bf16 mirror of an fp32 table is derived data that only makes batched searches faster.  When
// rows no longer fit beside it (HBM or gpu_resource_config.cache_capacity), it goes: the rows win, and
// batched searches over this table convert the fp32 rows on the fly from then on (as with option
// bf16_mirror 0).  The caller holds the table's exclusive lock, so no search is being enqueued; work
// already enqueued that reads the mirror is waited for.
static void table_drop_mirror(Table& t) {
    if (!t.mirrored()) return;
    (void)hipDeviceSynchronize();
    const size_t have = t.a_rows16.mapped;
    varray_free(t.a_rows16);
    t.d_rows16 = nullptr;
    t.ld16 = 0;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        G.table_bytes = G.table_bytes >= have ? G.table_bytes - have : 0;
        ++G.mirrors_dropped;
    }
    std::fprintf(stderr, "vdb: table %s: rows no longer fit beside the bf16 mirror (%zu MiB), which was dropped; "
                         "batched searches over this table now convert its fp32 rows on the fly\n",
                 t.name.c_str(), have >> 20);
}

static void table_free(Table& t) {
    (void)hipDeviceSynchronize();
    size_t have = 0;
    for (VArray* a : {&t.a_rows, &t.a_ids, &t.a_n2, &t.a_rows16}) { have += a->mapped; varray_free(*a); }
    if (t.d_maxn2) (void)hipFree(t.d_maxn2);
    if (t.d_stat) (void)hipFree(t.d_stat);
    t.d_stat = nullptr;
    {
        std::lock_guard<std::mutex> lk(t.stat_mu);
        t.pub = Table::OutPub();
    }
    if (t.h_bad) (void)hipHostFree(t.h_bad);
    t.d_rows = nullptr; t.d_ids = nullptr; t.d_n2 = nullptr; t.d_rows16 = nullptr; t.d_maxn2 = nullptr;
    t.h_bad = nullptr; t.d_bad = nullptr;
    t.rows.store(0); t.cap.store(0); t.dropped = true;
    std::lock_guard<std::mutex> lk(G.mu);
    G.table_bytes = G.table_bytes >= have ? G.table_bytes - have : 0;
}

}  // namespace vdb

using namespace vdb;

// ================================================================================== C ABI
extern "C" {

const char* vdb_last_error(void) { return tl_error.c_str(); }

int vdb_init(int device) {
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.inited) {
        if (device == G.device) return VDB_OK;
        return fail(VDB_ESTATE, "already initialised on another device");
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(VDB_ENOGPU, std::string("no usable HIP device: ") +
                                    (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= count) return fail(VDB_EINVAL, "device ordinal out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(VDB_ENOGPU, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(VDB_ENOGPU, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    G.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = hipMalloc((void**)&G.d_counters, 32 * sizeof(long long));
    if (e == hipSuccess) e = hipMemset(G.d_counters, 0, 32 * sizeof(long long));
    if (e == hipSuccess) {
        // appends are tiny next to a table scan: a high-priority stream lets them run between the
        // workgroups of the scans that are queued, instead of behind all of them
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); hi = 0; }
        e = hipStreamCreateWithPriority(&G.insert_stream, hipStreamDefault, hi);
    }
    if (e != hipSuccess) return fail(VDB_ENOGPU, std::string("device set-up: ") + hipGetErrorString(e));
    G.device = device;
    if (device < NOALLOC_MAX_DEV && !g_noalloc_ptr[device]) {
        // this device's copy of the scratch array (each GPU holds its own, at its own address)
        void* sp = nullptr;
        e = hipGetSymbolAddress(&sp, HIP_SYMBOL(vdb_noalloc_scratch));
        if (e != hipSuccess || !sp) return fail(VDB_ENOGPU, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_noalloc_ptr[device] = sp;
    }
    {
        // allocation granularity of the virtual-memory API; 0 leaves tables on plain allocations
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        // chunks are sized and placed in multiples of 2 MiB whatever smaller granule the runtime reports:
        // access rights on a chunk at a 4 KiB granule offset were refused (hipMemSetAccess: invalid argument)
        size_t own = (size_t)2 << 20;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0) {
            while (own < gran) own <<= 1;
            G.vmm_gran = own;
        }
        else { (void)hipGetLastError(); G.vmm_gran = 0; }
    }
    G.inited = true;
    return VDB_OK;
}

int vdb_shutdown(void) {
    if (!G.inited) return VDB_OK;
    (void)hipSetDevice(G.device);
    (void)hipDeviceSynchronize();
    std::vector<std::shared_ptr<Table>> ts;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        for (auto& kv : G.tables) ts.push_back(kv.second);
        G.tables.clear();
    }
    for (auto& t : ts) {
        std::unique_lock<std::shared_mutex> tl(t->mu);
        table_free(*t);
    }
    std::lock_guard<std::mutex> lk(G.mu);
    for (auto& w : G.free_ws) {
        for (DevBuf* b : {&w->qprep, &w->qinfo, &w->cand_s, &w->cand_t, &w->need_exact, &w->q_raw,
                          &w->o_ids, &w->o_scores, &w->o_counts, &w->small, &w->cand_h, &w->q16, &w->qimg})
            if (b->p) (void)hipFree(b->p);
        if (w->h_pin) (void)hipHostFree(w->h_pin);
        if (w->stream) (void)hipStreamDestroy(w->stream);
        if (w->done) (void)hipEventDestroy(w->done);
    }
    G.free_ws.clear();
    if (G.stage_rows.p) (void)hipFree(G.stage_rows.p);
    if (G.stage_ids.p) (void)hipFree(G.stage_ids.p);
    G.stage_rows = DevBuf(); G.stage_ids = DevBuf();
    if (G.insert_stream) (void)hipStreamDestroy(G.insert_stream);
    G.insert_stream = nullptr;
    {
        std::lock_guard<std::mutex> cl(G.chain_mu);
        for (hipEvent_t& e : G.chain_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        G.chain_head = -1; G.chain_last = nullptr;
    }
    {
        std::lock_guard<std::mutex> pl(G.prof_mu);
        for (auto& pr : G.prof_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (auto& pr : G.prof_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        G.prof_pool.clear(); G.prof_events.clear();
    }
    if (G.d_counters) (void)hipFree(G.d_counters);
    G.d_counters = nullptr;
    G.inited = false;
    return VDB_OK;
}

int vdb_set_option(const char* key, double value) {
    if (!key) return fail(VDB_EINVAL, "null option key");
    std::lock_guard<std::mutex> lk(G.mu);
    const std::string k(key);
    if (k == "use_blas_threshold") { if (value < 1) return fail(VDB_EINVAL, "use_blas_threshold < 1"); G.use_blas_threshold = (int)value; }
    else if (k == "cache_capacity_gb") { if (value < 0) return fail(VDB_EINVAL, "cache_capacity_gb < 0"); G.cache_capacity_gb = value; }
    else if (k == "insert_buffer_size_gb") { if (value <= 0) return fail(VDB_EINVAL, "insert_buffer_size_gb <= 0"); G.insert_buffer_gb = value; }
    else if (k == "scan_nt") G.tune_nt = value != 0;
    else if (k == "scan_waves") { if (value < 0 || value > 16) return fail(VDB_EINVAL, "scan_waves not in [0,16]"); G.tune_waves = (int)value; }
    else if (k == "scan_blocks_per_cu") { if (value < 0 || value > 32) return fail(VDB_EINVAL, "scan_blocks_per_cu not in [0,32]"); G.tune_bpc = (int)value; }
    else if (k == "scan_contiguous") G.tune_contig = value != 0;
    else if (k == "scan_dynamic") G.tune_dyn = value != 0;
    else if (k == "scan_short_gb") { if (value < 0) return fail(VDB_EINVAL, "scan_short_gb < 0"); set_scan_short_bytes(value * 1e9); }
    else if (k == "scan_serialize") { std::lock_guard<std::mutex> cl(G.chain_mu); G.scan_serialize = value != 0; }
    else if (k == "scan_sample") G.tune_sample = (int)value;     // 0 off, 1 rows up to 1024 bytes, larger: that many bytes
    else if (k == "coalesce_host") G.coalesce_host = value != 0;
    else if (k == "host_direct") G.host_direct = value != 0;
    else if (k == "fin_threads") set_fin_threads((int)value);
    else if (k == "gemm_path") G.tune_gemm = value != 0;
    else if (k == "gemm_split_f16") G.tune_split_f16 = value != 0;
    else if (k == "gemm32_split") G.tune_gemm32s = value < 0 ? 0 : value > 2 ? 2 : (int)value;
    else if (k == "gemm_split_bf16") G.tune_split_bf16 = value != 0;
    else if (k == "gemm32s_threads") set_gemm32s_threads((int)value);
    else if (k == "gemm_pp") G.tune_pp = value != 0;
    else if (k == "gemm_pp_tile") set_pp_tile_rows((int)value);
    else if (k == "force_exact") G.force_exact = value != 0;
    else if (k == "gemm_pp_parts") G.tune_pp_parts = value != 0;
    else if (k == "scan_mirror") G.scan_mirror = value != 0;
    else if (k == "scan_mirror_tier2") G.mirror_tier2 = value != 0;
    else if (k == "l2_row_bounds") G.row_bounds = value != 0;
    else if (k == "bf16_mirror") G.bf16_mirror = value != 0;
    else if (k == "virtual_ranges") G.use_vmm = value != 0;
    else if (k == "ip_outliers") G.ip_outliers = value != 0;
    else if (k == "debug_fail_map") G.debug_fail_map.store(value > 0 ? (int)value : 0);
    else return fail(VDB_EINVAL, "unknown option: " + k);
    set_scan_tune(G.tune_nt, G.tune_waves, G.tune_bpc, G.tune_contig, G.tune_dyn);
    return VDB_OK;
}

int vdb_create_table(const char* name, int64_t dim, int metric, int dtype) {
    int rc = enter();
    if (rc) return rc;
    if (!name || !*name) return fail(VDB_EINVAL, "empty table name");
    if (dim < 1 || dim > 16384) return fail(VDB_EINVAL, "dimension must be in [1, 16384]");
    if (metric != VDB_METRIC_L2 && metric != VDB_METRIC_IP && metric != VDB_METRIC_COSINE)
        return fail(VDB_EINVAL, "unknown metric");
    if (dtype != VDB_F32 && dtype != VDB_BF16 && dtype != VDB_F16) return fail(VDB_EINVAL, "unknown dtype");
    auto t = std::make_shared<Table>();
    t->name = name; t->dim = dim; t->metric = metric; t->dtype = dtype;
    t->gen = G.next_gen.fetch_add(1);
    {
        std::random_device rd;
        t->uuid = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull) ^ t->gen;
        if (!t->uuid) t->uuid = 1;
    }
    const int epl = elems_per_unit(dtype);
    t->m = (int)((dim + epl - 1) / epl);
    t->dpad = t->m * epl;
    if (dtype == VDB_F32) t->ld16 = G.bf16_mirror ? (dim + 31) / 32 * 32 : 0;
    else t->ld16 = (int64_t)t->m * 8;
    VDB_HIP(hipMalloc((void**)&t->d_maxn2, 16));
    VDB_HIP(hipMemset(t->d_maxn2, 0, 16));
    hipError_t e = hipHostMalloc((void**)&t->h_bad, 4 * sizeof(int), hipHostMallocMapped);       // [0] refused row seen, [1] bits of the norm bound
    if (e == hipSuccess) { t->h_bad[0] = 0; t->h_bad[1] = 0; e = hipHostGetDevicePointer((void**)&t->d_bad, t->h_bad, 0); }
    if (e != hipSuccess) {
        (void)hipFree(t->d_maxn2);
        if (t->h_bad) (void)hipHostFree(t->h_bad);
        return fail(VDB_EHIP, std::string("pinned flag for appends: ") + hipGetErrorString(e));
    }
    (void)table_reserve_ranges(*t);
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.tables.count(name)) {
        for (VArray* a : {&t->a_rows, &t->a_ids, &t->a_n2, &t->a_rows16}) varray_free(*a);
        (void)hipFree(t->d_maxn2);
        (void)hipHostFree(t->h_bad);
        return fail(VDB_EEXISTS, std::string("table exists: ") + name);
    }
    G.tables[name] = t;
    return VDB_OK;
}

int vdb_has_table(const char* name, int* out_exists) {
    if (!G.inited) return fail(VDB_ESTATE, "vdb_init has not been called");
    if (!name || !out_exists) return fail(VDB_EINVAL, "null argument");
    *out_exists = find_table(name) ? 1 : 0;
    return VDB_OK;
}

int vdb_drop_table(const char* name) {
    int rc = enter();
    if (rc) return rc;
    if (!name) return fail(VDB_EINVAL, "null table name");
    std::shared_ptr<Table> t;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        auto it = G.tables.find(name);
        if (it == G.tables.end()) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
        t = it->second;
        G.tables.erase(it);
    }
    std::unique_lock<std::shared_mutex> tl(t->mu);
    table_free(*t);
    return VDB_OK;
}

int vdb_create_index(const char* name, int index_type, int nlist) {
    if (!G.inited) return fail(VDB_ESTATE, "vdb_init has not been called");
    if (!name) return fail(VDB_EINVAL, "null table name");
    if (!find_table(name)) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    if (index_type != VDB_INDEX_FLAT && index_type != VDB_INDEX_IVFLAT) return fail(VDB_EINVAL, "unknown index type");
    if (nlist < 0) return fail(VDB_EINVAL, "nlist < 0");
    return VDB_OK;     // every search scans all rows: there is nothing to build
}

int vdb_count(const char* name, int64_t* out_rows) {
    if (!G.inited) return fail(VDB_ESTATE, "vdb_init has not been called");
    if (!name || !out_rows) return fail(VDB_EINVAL, "null argument");
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    std::shared_lock<std::shared_mutex> tl(t->mu);
    *out_rows = t->rows.load();
    return VDB_OK;
}

int vdb_table_info(const char* name, int64_t* dim, int* metric, int* dtype, int64_t* rows,
                   int64_t* capacity, int64_t* row_stride_bytes) {
    if (!G.inited) return fail(VDB_ESTATE, "vdb_init has not been called");
    if (!name) return fail(VDB_EINVAL, "null table name");
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    std::shared_lock<std::shared_mutex> tl(t->mu);
    if (dim) *dim = t->dim;
    if (metric) *metric = t->metric;
    if (dtype) *dtype = t->dtype;
    if (rows) *rows = t->rows.load();
    if (capacity) *capacity = t->cap.load();
    if (row_stride_bytes) *row_stride_bytes = (int64_t)t->m * 16;
    return VDB_OK;
}

int64_t vdb_tables_created(void) { return (int64_t)G.next_gen.load() - 1; }

int vdb_table_mirror_bytes(const char* name, int64_t* out_bytes) {
    if (!G.inited) return fail(VDB_ESTATE, "vdb_init has not been called");
    if (!name || !out_bytes) return fail(VDB_EINVAL, "null argument");
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    std::shared_lock<std::shared_mutex> tl(t->mu);
    *out_bytes = t->mirrored() ? (int64_t)t->a_rows16.mapped : 0;
    return VDB_OK;
}

int vdb_reserve(const char* name, int64_t rows) {
    int rc = enter();
    if (rc) return rc;
    if (!name || rows < 0) return fail(VDB_EINVAL, "bad argument");
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    if (t->vmm) {
        std::shared_lock<std::shared_mutex> tl(t->mu);
        std::lock_guard<std::mutex> il(t->ins_mu);
        if (t->dropped) return fail(VDB_ENOTFOUND, "table was dropped");
        const int rc2 = table_alloc(*t, rows);
        if (rc2 != VDB_ENOMEM || !t->mirrored()) return rc2;
    }
    std::unique_lock<std::shared_mutex> tl(t->mu);
    std::lock_guard<std::mutex> il(t->ins_mu);
    if (t->dropped) return fail(VDB_ENOTFOUND, "table was dropped");
    rc = table_alloc(*t, rows);
    if (rc == VDB_ENOMEM && t->mirrored()) {
        table_drop_mirror(*t);                       // rows before the mirror: see table_drop_mirror
        rc = table_alloc(*t, rows);
    }
    return rc;
}

// IP on raw vectors is the reference's metric (/root/reference/app.py:22,141), and the bounds of the
// batched path scale with the longest row of the table: a few rows a hundred times longer than the rest
// (and unrelated to the queries) would send every batched query to the exact pass.  Such rows are kept on a
// list instead: rows whose squared norm exceeds 64 x the upper edge of the median bucket of a histogram
// of the norms' exponents (8 to 11 times the typical norm).  The batched path takes its bound over the
// other rows and re-scores the listed ones canonically for every query (finalize_band_kernel).  More
// than OUTLIER_CAP such rows: no list, the table-wide bound as before.  Called with ins_mu held, after
// rows [stat_rows, end) have landed and before they are published.
static int table_stats_update(Table& t, int64_t end, hipStream_t st) {
    if (t.metric != VDB_METRIC_IP || !G.ip_outliers || end <= t.stat_rows) { t.stat_rows = end > t.stat_rows ? end : t.stat_rows; return VDB_OK; }
    const float inf = __builtin_huge_valf();
    if (!t.d_stat) {
        VDB_HIP(hipMalloc((void**)&t.d_stat, 2048 + 2 * OUTLIER_CAP * sizeof(long long)));
        VDB_HIP(hipMemsetAsync(t.d_stat, 0, 2048, st));
    }
    unsigned int* d_hist = (unsigned int*)t.d_stat;
    int* d_nout = (int*)(t.d_stat + 1024);
    float* d_maxrest = (float*)(t.d_stat + 1028);
    long long* lists[2] = {(long long*)(t.d_stat + 2048), (long long*)(t.d_stat + 2048) + OUTLIER_CAP};
    const int64_t row0 = t.stat_rows, n = end - row0;
    const bool sampled = row0 < 4096 || ((row0 + 63) & ~(int64_t)63) < end;
    if (sampled) {
        unsigned int h[256];
        VDB_HIP(launch_norm_hist(t.d_n2, row0, n, d_hist, st));
        VDB_HIP(hipMemcpyAsync(h, d_hist, sizeof h, hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        unsigned long long total = 0, acc = 0;
        for (unsigned v : h) total += v;
        t.cand_limit = inf;
        if (total >= 16) {
            int b = 0;
            for (; b < 256; ++b) { acc += h[b]; if (2 * acc >= total) break; }
            if (b >= 1 && b <= 240) t.cand_limit = std::ldexp(64.0f, b - 126);        // bucket b holds [2^(b-127), 2^(b-126))
        }
    }
    const float maxn2 = __builtin_bit_cast(float, t.h_bad[1]);
    const bool want = maxn2 > t.cand_limit;
    Table::OutPub pub;
    if (want) {
        const bool extend = t.limit_used == t.cand_limit;
        if (!extend) {
            // another limit: the list is built anew in the buffer no search reads (searches enqueued long
            // ago may still read the one before: wait for them)
            VDB_HIP(hipDeviceSynchronize());
            t.out_cur ^= 1;
            VDB_HIP(hipMemsetAsync(d_nout, 0, 8, st));
            VDB_HIP(launch_collect_outliers(t.d_n2, 0, end, t.cand_limit, lists[t.out_cur], d_nout, d_maxrest, st));
            t.limit_used = t.cand_limit;
        } else {
            VDB_HIP(launch_collect_outliers(t.d_n2, row0, n, t.limit_used, lists[t.out_cur], d_nout, d_maxrest, st));
        }
        struct { int n; float mr; } res;
        VDB_HIP(hipMemcpyAsync(&res, d_nout, sizeof res, hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        if (res.n > 0 && res.n <= OUTLIER_CAP) { pub.limit = t.limit_used; pub.list = lists[t.out_cur]; pub.n = res.n; pub.max_rest = res.mr; }
    } else {
        t.limit_used = inf;
    }
    {
        std::lock_guard<std::mutex> lk(t.stat_mu);
        t.pub = pub;
    }
    t.stat_rows = end;
    return VDB_OK;
}

// Appends run one at a time per table and beside searches: the new rows are written behind the
// published row count, the stream is drained, and only then is the count advanced (one atomic
// store), so a search sees either none or all rows of a call.  A call that meets a row that is not
// finite in the table's storage type publishes nothing and fails with VDB_EINVAL.
static int insert_locked(Table& t, int64_t n, const float* rows, const long long* ids,
                         bool host, bool synth, uint64_t seed, int64_t first_row, hipStream_t user_st) {
    int rc;
    if ((rc = table_grow_for(t, n))) return rc;
    const int d = (int)t.dim;
    const int64_t base = t.rows.load();
    unsigned short* mirror = t.mirrored() ? t.d_rows16 : nullptr;
    *t.h_bad = 0;
    // the call's largest row norm is collected beside the table's (d_maxn2[1]) and joins it only when no
    // row of the call was refused
    float* const pending = t.d_maxn2 + 1;
    if (synth || !host) {
        hipStream_t st = synth ? G.insert_stream : user_st;
        VDB_HIP(hipMemsetAsync(pending, 0, sizeof(float), st));
        VDB_HIP(launch_append(t.d_rows, mirror, t.ld16, t.d_ids, t.d_n2, pending, t.d_bad, base, n, d, t.m, t.dtype,
                              t.metric, synth ? nullptr : rows, synth ? nullptr : ids, seed, first_row, st));
        VDB_HIP(launch_fold_max(t.d_maxn2, pending, t.d_bad, st));
        VDB_HIP(hipStreamSynchronize(st));
    } else {
        std::lock_guard<std::mutex> il(G.insert_mu);
        double cap_rows_d = G.insert_buffer_gb * 1073741824.0 / ((double)d * sizeof(float) + sizeof(long long));
        int64_t chunk = cap_rows_d > 1e15 ? n : (int64_t)cap_rows_d;
        if (chunk < 1) chunk = 1;
        if (chunk > n) chunk = n;
        if ((rc = ensure(G.stage_rows, (size_t)chunk * d * sizeof(float), G.insert_stream))) return rc;
        if ((rc = ensure(G.stage_ids, (size_t)chunk * sizeof(long long), G.insert_stream))) return rc;
        VDB_HIP(hipMemsetAsync(pending, 0, sizeof(float), G.insert_stream));
        for (int64_t i0 = 0; i0 < n; i0 += chunk) {
            const int64_t ni = n - i0 < chunk ? n - i0 : chunk;
            VDB_HIP(hipMemcpyAsync(G.stage_rows.p, rows + (size_t)i0 * d, (size_t)ni * d * sizeof(float),
                                   hipMemcpyHostToDevice, G.insert_stream));
            VDB_HIP(hipMemcpyAsync(G.stage_ids.p, ids + i0, (size_t)ni * sizeof(long long),
                                   hipMemcpyHostToDevice, G.insert_stream));
            VDB_HIP(launch_append(t.d_rows, mirror, t.ld16, t.d_ids, t.d_n2, pending, t.d_bad, base + i0, ni, d, t.m,
                                  t.dtype, t.metric, (const float*)G.stage_rows.p, (const long long*)G.stage_ids.p,
                                  0, 0, G.insert_stream));
            if (i0 + ni >= n) VDB_HIP(launch_fold_max(t.d_maxn2, pending, t.d_bad, G.insert_stream));
            VDB_HIP(hipStreamSynchronize(G.insert_stream));      // the staging buffers are reused by the next chunk
        }
    }
    if (*t.h_bad) return fail(VDB_EINVAL, "a row holds a value that is not finite in the table's storage type; nothing was inserted");
    if ((rc = table_stats_update(t, base + n, (synth || host) ? G.insert_stream : user_st))) return rc;
    t.rows.store(base + n, std::memory_order_release);
    return VDB_OK;
}

static int insert_into(Table& t, int64_t n, const float* rows, const long long* ids,
                       bool host, bool synth, uint64_t seed, int64_t first_row, hipStream_t user_st) {
    if (t.vmm) {
        {
            // searches keep running: nothing they read moves or changes
            std::shared_lock<std::shared_mutex> tl(t.mu);
            std::lock_guard<std::mutex> il(t.ins_mu);
            if (t.dropped) return fail(VDB_ENOTFOUND, "table was dropped");
            const int rc = insert_locked(t, n, rows, ids, host, synth, seed, first_row, user_st);
            if (rc != VDB_ENOMEM || !t.mirrored()) return rc;
        }
        // out of memory with a mirror in place: the mirror makes room (exclusive lock: no search may be
        // reading it), and the insert is tried once more
        std::unique_lock<std::shared_mutex> tl(t.mu);
        std::lock_guard<std::mutex> il(t.ins_mu);
        if (t.dropped) return fail(VDB_ENOTFOUND, "table was dropped");
        table_drop_mirror(t);
        return insert_locked(t, n, rows, ids, host, synth, seed, first_row, user_st);
    }
    std::unique_lock<std::shared_mutex> tl(t.mu);
    std::lock_guard<std::mutex> il(t.ins_mu);
    if (t.dropped) return fail(VDB_ENOTFOUND, "table was dropped");
    int rc = insert_locked(t, n, rows, ids, host, synth, seed, first_row, user_st);
    if (rc == VDB_ENOMEM && t.mirrored()) {
        table_drop_mirror(t);
        rc = insert_locked(t, n, rows, ids, host, synth, seed, first_row, user_st);
    }
    return rc;
}

static int insert_common(const char* name, int64_t n, const float* rows, const long long* ids,
                         bool host, bool synth, uint64_t seed, int64_t first_row, hipStream_t user_st) {
    int rc = enter();
    if (rc) return rc;
    if (!name || n < 0) return fail(VDB_EINVAL, "bad argument");
    if (!synth && n > 0 && (!rows || !ids)) return fail(VDB_EINVAL, "null rows or ids");
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    if (n == 0) return VDB_OK;
    return insert_into(*t, n, rows, ids, host, synth, seed, first_row, user_st);
}

// Small host inserts that arrive from several threads at once are merged like blocking searches are:
// the first caller appends at once, callers that arrive meanwhile wait, and the next one to get the
// turn appends everything waiting for the same table in one copy and one launch (each caller's rows
// stay together).  A merged append that is refused (a row that is not finite) is repeated caller by
// caller, so that only the caller with the bad row sees the error.
struct InsReq {
    MergeKey       key;                  // (table, -1)
    int64_t        weight = 0;           // = n
    bool           done = false;         // guarded by the merger
    std::shared_ptr<Table> t;
    int64_t        n = 0;
    const float*   rows = nullptr;
    const long long* ids = nullptr;
    int            rc = VDB_OK;
    std::string    err;
};

static void insert_host_batch(std::vector<InsReq*>& batch) {
    auto one = [](InsReq* r) {
        r->rc = insert_into(*r->t, r->n, r->rows, r->ids, true, false, 0, 0, nullptr);
        if (r->rc != VDB_OK) r->err = tl_error;
    };
    if (batch.size() == 1) { one(batch[0]); return; }
    Table& t = *batch[0]->t;
    int64_t total = 0;
    for (InsReq* r : batch) total += r->n;
    std::vector<float> rows((size_t)total * t.dim);
    std::vector<long long> ids((size_t)total);
    int64_t at = 0;
    for (InsReq* r : batch) {
        std::memcpy(rows.data() + (size_t)at * t.dim, r->rows, (size_t)r->n * t.dim * sizeof(float));
        std::memcpy(ids.data() + at, r->ids, (size_t)r->n * sizeof(long long));
        at += r->n;
    }
    const int rc = insert_into(t, total, rows.data(), ids.data(), true, false, 0, 0, nullptr);
    if (rc == VDB_OK) { for (InsReq* r : batch) r->rc = VDB_OK; return; }
    for (InsReq* r : batch) one(r);          // nothing was published: every caller gets its own verdict
}

int vdb_insert(const char* name, int64_t n, const float* rows, const int64_t* ids) {
    if (!G.coalesce_host || n < 1 || n > 256 || !name || !rows || !ids)
        return insert_common(name, n, rows, (const long long*)ids, true, false, 0, 0, nullptr);
    int rc = enter();
    if (rc) return rc;
    InsReq req;
    req.t = find_table(name);
    if (!req.t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    req.n = n; req.rows = rows; req.ids = (const long long*)ids;
    req.key = MergeKey(req.t->gen, -1);
    req.weight = n;
    static CallMerger<InsReq> merger;
    merger.submit(req, 4096, [](std::vector<InsReq*>& b) { insert_host_batch(b); },
                  [](std::vector<InsReq*>& b) { for (InsReq* r : b) { r->rc = VDB_ENOMEM; r->err = "out of host memory while merging inserts"; } });
    return req.rc == VDB_OK ? VDB_OK : fail(req.rc, req.err);
}

int vdb_insert_dev(const char* name, int64_t n, const float* rows_dev, const int64_t* ids_dev, void* stream) {
    return insert_common(name, n, rows_dev, (const long long*)ids_dev, false, false, 0, 0, (hipStream_t)stream);
}

int vdb_insert_synthetic(const char* name, int64_t n, uint64_t seed, int64_t first_row) {
    return insert_common(name, n, nullptr, nullptr, false, true, seed, first_row, nullptr);
}

int vdb_synth_fill_dev(float* out_dev, int64_t n, int64_t dim, uint64_t seed, int64_t first_row, void* stream) {
    int rc = enter();
    if (rc) return rc;
    if (!out_dev || n < 0 || dim < 1 || dim > 16384) return fail(VDB_EINVAL, "bad argument");
    VDB_HIP(launch_synth_fill(out_dev, n, (int)dim, seed, first_row, (hipStream_t)stream));
    return VDB_OK;
}

int vdb_fetch_rows(const char* name, int64_t first, int64_t n, float* out_rows, int64_t* out_ids) {
    int rc = enter();
    if (rc) return rc;
    if (!name || first < 0 || n < 0) return fail(VDB_EINVAL, "bad argument");
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    std::shared_lock<std::shared_mutex> tl(t->mu);
    if (first + n > t->rows.load()) return fail(VDB_EINVAL, "row range beyond the table");
    if (n == 0) return VDB_OK;
    if (out_rows) {
        float* tmp = nullptr;
        VDB_HIP(hipMalloc((void**)&tmp, (size_t)n * t->dim * sizeof(float)));
        hipError_t e = launch_fetch_rows(t->d_rows, first, n, (int)t->dim, t->m, t->dtype, tmp, nullptr);
        if (e == hipSuccess) e = hipMemcpy(out_rows, tmp, (size_t)n * t->dim * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(VDB_EHIP, std::string("fetch rows: ") + hipGetErrorString(e));
    }
    if (out_ids) VDB_HIP(hipMemcpy(out_ids, t->d_ids + first, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    return VDB_OK;
}

// ---------------------------------------------------------------------------------- snapshots
// File layout (little endian), version 2: SnapHeader2, then segments back to back, each a SegHeader
// followed by rows[nrows * m * 16], ids[nrows * 8], n2[nrows * 4].  The segments cover rows
// [0, header.rows) in order; bytes behind the last segment the header accounts for (an append that
// did not reach its header update) are ignored.  The stored rows are already quantised (and normalised
// for COSINE), so a loaded table is bit-identical to the saved one and searches on it return the same
// ids and scores.  A snapshot carries the identity of the table it was taken from (a random 64-bit
// value drawn at create time, kept across load): a later save of the same table to the same path
// appends a segment with the rows that arrived since and rewrites the header last, instead of
// writing the whole table again.  Version 1 files (SnapHeader, then the three arrays) are still read.
struct SnapHeader {
    char     magic[8];      // "VDBSNAP1"
    int64_t  dim;
    int32_t  metric, dtype;
    int64_t  rows;
    int32_t  m;
    float    max_n2;
};
struct SnapHeader2 {
    char     magic[8];      // "VDBSNAP2"
    int64_t  dim;
    int32_t  metric, dtype;
    int64_t  rows;          // rows the segments that count cover
    int32_t  m;
    float    max_n2;
    uint64_t uuid;          // identity of the table
    int64_t  nsegs;
    int64_t  reserved[2];
};
struct SegHeader {
    char     magic[8];      // "VDBSEG01"
    int64_t  first_row, nrows;
};

static const size_t SNAP_CHUNK = (size_t)64 << 20;

struct HostPinned {
    void* p = nullptr;
    ~HostPinned() { if (p) (void)hipHostFree(p); }
};
struct FileCloser {
    FILE* f = nullptr;
    ~FileCloser() { if (f) fclose(f); }
};

static int snap_write(FILE* f, const char* dev, size_t bytes, void* bounce) {
    for (size_t o = 0; o < bytes; o += SNAP_CHUNK) {
        const size_t nb = bytes - o < SNAP_CHUNK ? bytes - o : SNAP_CHUNK;
        VDB_HIP(hipMemcpy(bounce, dev + o, nb, hipMemcpyDeviceToHost));
        if (fwrite(bounce, 1, nb, f) != nb) return fail(VDB_EINVAL, "short write to snapshot file");
    }
    return VDB_OK;
}

static int snap_read(FILE* f, char* dev, size_t bytes, void* bounce) {
    for (size_t o = 0; o < bytes; o += SNAP_CHUNK) {
        const size_t nb = bytes - o < SNAP_CHUNK ? bytes - o : SNAP_CHUNK;
        if (fread(bounce, 1, nb, f) != nb) return fail(VDB_EINVAL, "snapshot file is truncated");
        VDB_HIP(hipMemcpy(dev + o, bounce, nb, hipMemcpyHostToDevice));
    }
    return VDB_OK;
}

static int snap_sync(FILE* f) {
    if (fflush(f) != 0 || fsync(fileno(f)) != 0) return fail(VDB_EINVAL, "flush of snapshot file failed");
    return VDB_OK;
}

// one segment with rows [first, first + n) of the table at the file's current position
static int snap_write_segment(FILE* f, const Table& t, int64_t first, int64_t n, void* bounce) {
    SegHeader sh;
    std::memset(&sh, 0, sizeof sh);
    std::memcpy(sh.magic, "VDBSEG01", 8);
    sh.first_row = first; sh.nrows = n;
    if (fwrite(&sh, sizeof sh, 1, f) != 1) return fail(VDB_EINVAL, "short write to snapshot file");
    int rc;
    const size_t rb = (size_t)t.m * 16;
    if ((rc = snap_write(f, t.d_rows + (size_t)first * rb, (size_t)n * rb, bounce))) return rc;
    if ((rc = snap_write(f, (const char*)(t.d_ids + first), (size_t)n * sizeof(long long), bounce))) return rc;
    if ((rc = snap_write(f, (const char*)(t.d_n2 + first), (size_t)n * sizeof(float), bounce))) return rc;
    return VDB_OK;
}

static void snap_fill_header(SnapHeader2& h, const Table& t, int64_t rows, float max_n2, int64_t nsegs) {
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, "VDBSNAP2", 8);
    h.dim = t.dim; h.metric = t.metric; h.dtype = t.dtype; h.rows = rows; h.m = t.m;
    h.max_n2 = max_n2; h.uuid = t.uuid; h.nsegs = nsegs;
}

// File offset behind the last segment that header `h` accounts for, or -1 if the segments do not
// cover [0, h.rows) in order.
static long long snap_walk(FILE* f, const SnapHeader2& h) {
    long long off = (long long)sizeof(SnapHeader2);
    int64_t covered = 0;
    const long long per_row = (long long)h.m * 16 + 12;
    while (covered < h.rows) {
        SegHeader sh;
        if (fseeko(f, (off_t)off, SEEK_SET) != 0 || fread(&sh, sizeof sh, 1, f) != 1) return -1;
        if (std::memcmp(sh.magic, "VDBSEG01", 8) != 0 || sh.first_row != covered || sh.nrows < 1 || sh.nrows > h.rows - covered) return -1;
        off += (long long)sizeof sh + sh.nrows * per_row;
        covered += sh.nrows;
    }
    return off;
}

static int save_table_impl(const char* name, const char* path, int64_t* rows_written) {
    int rc = enter();
    if (rc) return rc;
    if (!name || !path || !*path) return fail(VDB_EINVAL, "null table name or path");
    if (rows_written) *rows_written = 0;
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    std::shared_lock<std::shared_mutex> tl(t->mu);
    if (t->dropped) return fail(VDB_ENOTFOUND, "table was dropped");
    std::lock_guard<std::mutex> sl(t->snap_mu);           // one save of a table at a time
    // the norm bound is read after the row count: it then covers at least the rows that are saved
    const int64_t rows = t->rows.load(std::memory_order_acquire);
    float max_n2 = 0.f;
    VDB_HIP(hipMemcpy(&max_n2, t->d_maxn2, sizeof(float), hipMemcpyDeviceToHost));
    HostPinned hp;
    VDB_HIP(hipHostMalloc(&hp.p, SNAP_CHUNK, hipHostMallocDefault));
    // a snapshot of this very table under the same name: append what arrived since
    {
        FileCloser fc;
        fc.f = fopen(path, "r+b");
        SnapHeader2 h;
        if (fc.f && fread(&h, sizeof h, 1, fc.f) == 1 && std::memcmp(h.magic, "VDBSNAP2", 8) == 0 &&
            h.uuid == t->uuid && h.dim == t->dim && h.metric == t->metric && h.dtype == t->dtype && h.m == t->m &&
            h.rows >= 0 && h.rows <= rows) {
            const long long end = snap_walk(fc.f, h);
            if (end > 0) {
                if (h.rows == rows) return VDB_OK;        // nothing new
                if (fseeko(fc.f, (off_t)end, SEEK_SET) != 0) return fail(VDB_EINVAL, "cannot seek in snapshot file");
                if ((rc = snap_write_segment(fc.f, *t, h.rows, rows - h.rows, hp.p))) return rc;
                if (fflush(fc.f) != 0 || ftruncate(fileno(fc.f), ftello(fc.f)) != 0) return fail(VDB_EINVAL, "cannot trim snapshot file");
                if ((rc = snap_sync(fc.f))) return rc;
                // the header goes last: until it is rewritten the file reads as the older snapshot
                SnapHeader2 h2;
                snap_fill_header(h2, *t, rows, max_n2 > h.max_n2 ? max_n2 : h.max_n2, h.nsegs + 1);
                if (fseeko(fc.f, 0, SEEK_SET) != 0 || fwrite(&h2, sizeof h2, 1, fc.f) != 1) return fail(VDB_EINVAL, "cannot rewrite snapshot header");
                if ((rc = snap_sync(fc.f))) return rc;
                if (rows_written) *rows_written = rows - h.rows;
                return VDB_OK;
            }
        }
    }
    // whole table: written under a temporary name, then renamed (a reader never sees half a snapshot)
    const std::string tmp = std::string(path) + ".tmp";
    {
        FileCloser fc;
        fc.f = fopen(tmp.c_str(), "wb");
        if (!fc.f) return fail(VDB_EINVAL, "cannot open snapshot file for writing: " + tmp);
        SnapHeader2 h;
        snap_fill_header(h, *t, rows, max_n2, rows > 0 ? 1 : 0);
        if (fwrite(&h, sizeof h, 1, fc.f) != 1) return fail(VDB_EINVAL, "short write to snapshot file");
        if (rows > 0 && (rc = snap_write_segment(fc.f, *t, 0, rows, hp.p))) return rc;
        if ((rc = snap_sync(fc.f))) return rc;
    }
    if (std::rename(tmp.c_str(), path) != 0) return fail(VDB_EINVAL, std::string("cannot rename snapshot to ") + path);
    if (rows_written) *rows_written = rows;
    return VDB_OK;
}

int vdb_save_table(const char* name, const char* path) { return save_table_impl(name, path, nullptr); }
int vdb_save_table_rows(const char* name, const char* path, int64_t* out_rows_written) {
    return save_table_impl(name, path, out_rows_written);
}

int vdb_load_table(const char* name, const char* path) {
    int rc = enter();
    if (rc) return rc;
    if (!name || !*name || !path || !*path) return fail(VDB_EINVAL, "null table name or path");
    FileCloser fc;
    fc.f = fopen(path, "rb");
    if (!fc.f) return fail(VDB_ENOTFOUND, std::string("cannot open snapshot file: ") + path);
    SnapHeader2 h;
    std::memset(&h, 0, sizeof h);
    char magic[8];
    if (fread(magic, 8, 1, fc.f) != 1) return fail(VDB_EINVAL, std::string("not a table snapshot: ") + path);
    const bool v1 = std::memcmp(magic, "VDBSNAP1", 8) == 0;
    if (!v1 && std::memcmp(magic, "VDBSNAP2", 8) != 0) return fail(VDB_EINVAL, std::string("not a table snapshot: ") + path);
    if (fseeko(fc.f, 0, SEEK_SET) != 0) return fail(VDB_EINVAL, "cannot seek in snapshot file");
    if (v1) {
        SnapHeader h1;
        if (fread(&h1, sizeof h1, 1, fc.f) != 1) return fail(VDB_EINVAL, std::string("not a table snapshot: ") + path);
        h.dim = h1.dim; h.metric = h1.metric; h.dtype = h1.dtype; h.rows = h1.rows; h.m = h1.m; h.max_n2 = h1.max_n2;
    } else if (fread(&h, sizeof h, 1, fc.f) != 1) {
        return fail(VDB_EINVAL, std::string("not a table snapshot: ") + path);
    }
    if (h.dim < 1 || h.dim > 16384 || h.rows < 0 ||
        (h.metric != VDB_METRIC_L2 && h.metric != VDB_METRIC_IP && h.metric != VDB_METRIC_COSINE) ||
        (h.dtype != VDB_F32 && h.dtype != VDB_BF16 && h.dtype != VDB_F16))
        return fail(VDB_EINVAL, "snapshot header holds values out of range");
    const int epl = elems_per_unit(h.dtype);
    if (h.m != (int)((h.dim + epl - 1) / epl)) return fail(VDB_EINVAL, "snapshot row stride does not match its dimension");
    if (!(h.max_n2 >= 0.0f)) return fail(VDB_EINVAL, "snapshot header holds values out of range");
    // the payload must be as long as the header promises before any device memory is taken
    if (fseeko(fc.f, 0, SEEK_END) != 0) return fail(VDB_EINVAL, "cannot seek in snapshot file");
    const long long fsize = (long long)ftello(fc.f);
    if (v1) {
        const long long payload = (long long)h.rows * ((long long)h.m * 16 + 12);
        if (fsize != (long long)sizeof(SnapHeader) + payload) return fail(VDB_EINVAL, "snapshot file length does not match its header");
    } else {
        const long long end = snap_walk(fc.f, h);
        if (end < 0 || end > fsize) return fail(VDB_EINVAL, "snapshot file length does not match its header");
    }

    if ((rc = vdb_create_table(name, h.dim, h.metric, h.dtype))) return rc;
    auto t = find_table(name);
    if (!t) return fail(VDB_ESTATE, "table vanished while loading");
    {
        std::unique_lock<std::shared_mutex> tl(t->mu);
        std::lock_guard<std::mutex> il(t->ins_mu);
        if (!v1) t->uuid = h.uuid;                    // later saves of the loaded table extend the same file
        if (h.rows > 0) {
            HostPinned hp;
            rc = table_alloc(*t, h.rows);
            if (rc == VDB_ENOMEM && t->mirrored()) { table_drop_mirror(*t); rc = table_alloc(*t, h.rows); }
            if (!rc && hipHostMalloc(&hp.p, SNAP_CHUNK, hipHostMallocDefault) != hipSuccess)
                rc = fail(VDB_ENOMEM, "cannot allocate the host staging buffer");
            const size_t rb = (size_t)h.m * 16;
            if (!rc && v1) {
                if (fseeko(fc.f, (off_t)sizeof(SnapHeader), SEEK_SET) != 0) rc = fail(VDB_EINVAL, "cannot seek in snapshot file");
                if (!rc) rc = snap_read(fc.f, t->d_rows, (size_t)h.rows * rb, hp.p);
                if (!rc) rc = snap_read(fc.f, (char*)t->d_ids, (size_t)h.rows * sizeof(long long), hp.p);
                if (!rc) rc = snap_read(fc.f, (char*)t->d_n2, (size_t)h.rows * sizeof(float), hp.p);
            } else if (!rc) {
                if (fseeko(fc.f, (off_t)sizeof(SnapHeader2), SEEK_SET) != 0) rc = fail(VDB_EINVAL, "cannot seek in snapshot file");
                int64_t covered = 0;
                while (!rc && covered < h.rows) {
                    SegHeader sh;
                    if (fread(&sh, sizeof sh, 1, fc.f) != 1 || std::memcmp(sh.magic, "VDBSEG01", 8) != 0 ||
                        sh.first_row != covered || sh.nrows < 1 || sh.nrows > h.rows - covered) { rc = fail(VDB_EINVAL, "snapshot segment is damaged"); break; }
                    rc = snap_read(fc.f, t->d_rows + (size_t)covered * rb, (size_t)sh.nrows * rb, hp.p);
                    if (!rc) rc = snap_read(fc.f, (char*)(t->d_ids + covered), (size_t)sh.nrows * sizeof(long long), hp.p);
                    if (!rc) rc = snap_read(fc.f, (char*)(t->d_n2 + covered), (size_t)sh.nrows * sizeof(float), hp.p);
                    covered += sh.nrows;
                }
            }
            if (!rc && t->mirrored()) {
                // the bf16 image of the rows is derived data: rebuilt, not stored
                hipError_t e = launch_build_mirror(t->d_rows, t->d_rows16, t->ld16, 0, h.rows, t->m, nullptr);
                if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
                if (e != hipSuccess) rc = fail(VDB_EHIP, std::string("rebuilding the bf16 image: ") + hipGetErrorString(e));
            }
        }
        if (!rc && hipMemcpy(t->d_maxn2, &h.max_n2, sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(VDB_EHIP, "cannot store the row-norm bound");
        if (!rc) {
            t->h_bad[1] = (int)__builtin_bit_cast(unsigned int, h.max_n2);
            rc = table_stats_update(*t, h.rows, nullptr);
        }
        if (!rc) t->rows.store(h.rows);
    }
    if (rc) {
        const std::string keep = tl_error;
        (void)vdb_drop_table(name);
        return fail(rc, keep);
    }
    return VDB_OK;
}

static int check_search_args(const char* name, int64_t nq, const void* q, int k, const void* a,
                             const void* b, const void* c) {
    if (!name || nq < 0) return fail(VDB_EINVAL, "bad argument");
    if (k < 1 || k > VDB_MAX_TOPK) return fail(VDB_EINVAL, "k must be in [1, VDB_MAX_TOPK]");
    if (nq > 0 && (!q || !a || !b || !c)) return fail(VDB_EINVAL, "null query or output pointer");
    return VDB_OK;
}

int vdb_search_dev(const char* name, int64_t nq, const float* queries_dev, int k,
                   int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev, void* stream) {
    int rc = enter();
    if (rc) return rc;
    if ((rc = check_search_args(name, nq, queries_dev, k, out_ids_dev, out_scores_dev, out_counts_dev))) return rc;
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    if (nq == 0) return VDB_OK;
    hipStream_t st = (hipStream_t)stream;
    std::unique_ptr<Workspace> ws;
    if ((rc = acquire_ws(st, ws))) return rc;
    {
        std::shared_lock<std::shared_mutex> tl(t->mu);
        if (t->dropped) rc = fail(VDB_ENOTFOUND, "table was dropped");
        else rc = search_core(*t, nq, queries_dev, k, (long long*)out_ids_dev, out_scores_dev, out_counts_dev, st, *ws);
    }
    release_ws(ws, st);
    return rc;
}

// Same search as vdb_search_dev for callers that must not allocate device memory (tight serving loops,
// harnesses that forbid it): every intermediate buffer is carved from the 24 MiB scratch array above,
// always by the per-query scan (no batched path), nothing is created, allocated or freed.
// All calls on one GPU share the array.  A call that arrives while another thread is inside the entry
// point is refused (VDB_ESTATE); calls enqueued one after the other on one stream are kept apart by
// stream order; calls on different streams must be ordered by the caller (the work of a call that has
// returned may still be running).
int vdb_search_dev_noalloc(const char* name, int64_t nq, const float* queries_dev, int k,
                           int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev, void* stream) {
    int rc = enter();
    if (rc) return rc;
    if ((rc = check_search_args(name, nq, queries_dev, k, out_ids_dev, out_scores_dev, out_counts_dev))) return rc;
    auto t = find_table(name);
    if (!t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    if (nq == 0) return VDB_OK;
    char* base = G.device < NOALLOC_MAX_DEV ? (char*)g_noalloc_ptr[G.device] : nullptr;
    if (!base) return fail(VDB_ESTATE, "no scratch array on this device");
    struct Busy {
        bool mine = false;
        Busy() { int expect = 0; mine = g_noalloc_busy.compare_exchange_strong(expect, 1); }
        ~Busy() { if (mine) g_noalloc_busy.store(0); }
    } busy;
    if (!busy.mine) return fail(VDB_ESTATE, "vdb_search_dev_noalloc: another call is using the scratch array");
    std::shared_lock<std::shared_mutex> tl(t->mu);
    if (t->dropped) return fail(VDB_ENOTFOUND, "table was dropped");
    // sizes as search_scan_path asks for them, for the table as it is now
    TableView tv{};
    tv.rows = t->d_rows; tv.n = t->rows.load(std::memory_order_acquire); tv.d = (int)t->dim; tv.m = t->m;
    tv.dt = t->dtype; tv.metric = t->metric;
    SelGeom g = sel_geom(k);
    if (G.tune_waves > 0 && G.tune_waves < g.W) g.W = G.tune_waves;
    const SelGeom gs = scan_geom(g, G.tune_waves);
    const int Gx = tv.n > 0 ? scan_grid(tv, gs, G.n_cu) : 1;
    const int Gxe = tv.n > 0 ? scan_grid(tv, g, G.n_cu) : 1;
    const size_t lists = (size_t)nq * ((size_t)Gx * gs.W > (size_t)Gxe ? (size_t)Gx * gs.W : (size_t)Gxe);
    const size_t want[7] = {(size_t)nq * t->dpad * sizeof(float), (size_t)nq * sizeof(QInfo), lists * g.K * sizeof(float),
                            lists * g.K * sizeof(long long), lists * sizeof(float), (size_t)(2 * nq + 2) * sizeof(int),
                            (size_t)nq * sizeof(float)};
    if (nq > 256) return fail(VDB_EINVAL, "vdb_search_dev_noalloc takes at most 256 queries per call");
    Workspace ws;
    ws.fixed = true;
    DevBuf* bufs[7] = {&ws.qprep, &ws.qinfo, &ws.cand_s, &ws.cand_t, &ws.cand_h, &ws.need_exact, &ws.small};
    size_t off = 0;
    for (int i = 0; i < 7; ++i) {
        const size_t sz = round_up(want[i], 256);
        if (off + sz > NOALLOC_BYTES)
            return fail(VDB_EINVAL, "this search needs " + std::to_string((off + sz + ((size_t)1 << 20) - 1) >> 20) +
                                    " MiB or more of scratch memory; vdb_search_dev_noalloc holds 24 MiB (fewer queries per call or a smaller k fit)");
        bufs[i]->p = base + off;
        bufs[i]->cap = sz;
        off += sz;
    }
    rc = search_core(*t, nq, queries_dev, k, (long long*)out_ids_dev, out_scores_dev, out_counts_dev, (hipStream_t)stream, ws);
    for (DevBuf* b : bufs) *b = DevBuf();          // nothing here is owned
    return rc;
}

namespace vdb {
// One blocking host call, possibly carrying the queries of several callers (see vdb_search).
struct HostReq {
    MergeKey       key;                  // (table, k): what may share one batched search
    int64_t        weight = 0;           // = nq
    bool           done = false;         // guarded by the merger
    std::shared_ptr<Table> t;
    int            k = 0;
    int64_t        nq = 0;
    const float*   queries = nullptr;
    int64_t*       out_ids = nullptr;
    float*         out_scores = nullptr;
    int32_t*       out_counts = nullptr;
    int            rc = VDB_OK;
    std::string    err;
};

// Runs the requests of `batch` (same table, same k) as one search.  The result code and message
// go into every request; the message is also this thread's last error.
static void search_host_batch(std::vector<HostReq*>& batch) {
    Table& tab = *batch[0]->t;
    const int k = batch[0]->k;
    int64_t nq = 0;
    for (HostReq* r : batch) nq += r->nq;
    int rc = VDB_OK;
    std::unique_ptr<Workspace> ws;
    if ((rc = acquire_ws(nullptr, ws, true))) {
        for (HostReq* r : batch) { r->rc = rc; r->err = tl_error; }
        return;
    }
    hipStream_t st = ws->stream;
    const size_t d = (size_t)tab.dim;
    // Results leave the device as one block (ids | scores | counts) and, like the queries, pass
    // through a pinned host buffer: copies from and to pageable memory are staged by the runtime
    // one by one and cost several times the 3 KB they move.
    const size_t qbytes = (size_t)nq * d * sizeof(float);
    const size_t ib = (size_t)nq * k * sizeof(long long), sb = (size_t)nq * k * sizeof(float), cb = (size_t)nq * sizeof(int);
    const size_t rbytes = ib + sb + cb;                      // the results
    // up to 256 queries (one chunk of the scan path): the lists of flagged queries ride behind the results
    const size_t nbytes = nq <= 256 ? round_up((size_t)(2 * nq + 2) * sizeof(int), 16) : 0;
    const size_t obytes = round_up(rbytes, 16) + nbytes;
    const size_t qspan = round_up(qbytes, 16);               // the answers start on a 16-byte boundary of the pinned block
    const bool pinned = qspan + obytes <= ((size_t)64 << 20);
    do {
        if ((rc = ensure(ws->q_raw, qbytes, ws->last_stream))) break;
        if ((rc = ensure(ws->o_ids, obytes, ws->last_stream))) break;
        char* o_dev = (char*)ws->o_ids.p;
        if (pinned && ws->h_cap < qspan + obytes) {
            if (ws->h_pin) { (void)hipStreamSynchronize(st); (void)hipHostFree(ws->h_pin); ws->h_pin = nullptr; ws->d_pin = nullptr; ws->h_cap = 0; }
            size_t want = (qspan + obytes) * 2;
            if (want < 65536) want = 65536;
            if (hipHostMalloc(&ws->h_pin, want, hipHostMallocMapped) != hipSuccess) { ws->h_pin = nullptr; rc = fail(VDB_ENOMEM, "pinned staging buffer"); break; }
            if (hipHostGetDevicePointer((void**)&ws->d_pin, ws->h_pin, 0) != hipSuccess) ws->d_pin = nullptr;
            ws->h_cap = want;
        }
        // a few queries: no copies, the kernels work on the pinned block itself
        const bool direct = pinned && nbytes > 0 && nq <= 4 && qbytes <= 16384 && ws->d_pin != nullptr && G.host_direct != 0;
        char* hq = pinned ? (char*)ws->h_pin : nullptr;
        char* ho = pinned ? hq + qspan : nullptr;
        hipError_t e = hipSuccess;
        if (pinned) {
            size_t off = 0;
            for (HostReq* r : batch) { std::memcpy(hq + off, r->queries, (size_t)r->nq * d * sizeof(float)); off += (size_t)r->nq * d * sizeof(float); }
            if (!direct) e = hipMemcpyAsync(ws->q_raw.p, hq, qbytes, hipMemcpyHostToDevice, st);
        } else {
            size_t off = 0;
            for (HostReq* r : batch) {
                if (e == hipSuccess) e = hipMemcpyAsync((char*)ws->q_raw.p + off, r->queries, (size_t)r->nq * d * sizeof(float), hipMemcpyHostToDevice, st);
                off += (size_t)r->nq * d * sizeof(float);
            }
        }
        if (e != hipSuccess) { rc = fail(VDB_EHIP, std::string("query upload: ") + hipGetErrorString(e)); break; }
        {
            std::shared_lock<std::shared_mutex> tl(tab.mu);
            const bool defer = pinned && nbytes > 0;
            ws->ne_ext = defer ? (int*)(o_dev + round_up(rbytes, 16)) : nullptr;
            ws->defer_exact = defer;
            ws->deferred.pending = false;
            int* hflag = direct ? (int*)(ho + round_up(rbytes, 16)) : nullptr;          // direct: one word per query, set by the finalize step
            if (direct) {
                for (int64_t i = 0; i < nq; ++i) hflag[i] = 0;
                ws->host_flag = (int*)(ws->d_pin + qspan + round_up(rbytes, 16));
            }
            char* const o_use = direct ? ws->d_pin + qspan : o_dev;
            if (tab.dropped) rc = fail(VDB_ENOTFOUND, "table was dropped");
            else rc = search_core(tab, nq, direct ? (const float*)ws->d_pin : (const float*)ws->q_raw.p, k, (long long*)o_use,
                                  (float*)(o_use + ib), (int*)(o_use + ib + sb), st, *ws);
            ws->defer_exact = false;
            ws->host_flag = nullptr;
            const bool flags = ws->host_flag_used;
            ws->host_flag_used = false;
            if (rc == VDB_OK && direct) {
                // The answers are in the pinned block when every query's word is set (1: final, 2: the exact pass is
                // still to come).  Over a table of a few megabytes the caller watches the words instead of waiting for
                // the stream: the kernel's last store reaches the host about 5 us before hipStreamSynchronize returns,
                // a tenth of such a call.  Not for long: after 200 us it waits for the stream like any other call.
                bool seen = false;
                if (flags && (double)tab.rows.load(std::memory_order_relaxed) * tab.m * 16.0 < 256e6) {
                    const auto t0 = std::chrono::steady_clock::now();
                    for (unsigned it = 1; !seen; ++it) {
                        bool all = true;
                        for (int64_t i = 0; i < nq; ++i) all = all && *(volatile int*)&hflag[i] != 0;
                        if (all) { seen = true; break; }
                        if ((it & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                }
                if (!seen) e = hipStreamSynchronize(st);
                if (e == hipSuccess && ws->deferred.pending) {
                    bool any = false;
                    if (flags) {
                        for (int64_t i = 0; i < nq; ++i) any = any || hflag[i] == 2;
                    } else {
                        int cnt = 0;
                        e = hipMemcpy(&cnt, ws->deferred.flagged, sizeof(int), hipMemcpyDeviceToHost);
                        any = cnt > 0;
                    }
                    if (e == hipSuccess && any) {
                        const Workspace::Deferred& df = ws->deferred;
                        e = launch_exact(df.tv, df.qprep, df.nq, df.dpad, df.g, df.Gxe, df.k, df.cs, df.ct, df.out_ids, df.out_scores,
                                         df.out_counts, df.flagged, G.d_counters, st);
                        if (e == hipSuccess) e = hipStreamSynchronize(st);
                    }
                }
                if (e != hipSuccess) rc = fail(VDB_EHIP, std::string("search: ") + hipGetErrorString(e));
            } else if (rc == VDB_OK && defer && ws->deferred.pending) {
                // results and the list of flagged queries in one copy; the exact pass only if the list holds a query
                e = hipMemcpyAsync(ho, o_dev, obytes, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                const Workspace::Deferred& df = ws->deferred;
                const int* hne = (const int*)(ho + round_up(rbytes, 16)) + (df.flagged - ws->ne_ext);
                if (e == hipSuccess && hne[0] > 0) {
                    e = launch_exact(df.tv, df.qprep, df.nq, df.dpad, df.g, df.Gxe, df.k, df.cs, df.ct, df.out_ids, df.out_scores,
                                     df.out_counts, df.flagged, G.d_counters, st);
                    if (e == hipSuccess) e = hipMemcpyAsync(ho, o_dev, rbytes, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                }
                if (e != hipSuccess) rc = fail(VDB_EHIP, std::string("result download: ") + hipGetErrorString(e));
            } else if (rc == VDB_OK) {
                if (pinned) {
                    e = hipMemcpyAsync(ho, o_dev, obytes, hipMemcpyDeviceToHost, st);
                } else {
                    int64_t q0 = 0;
                    for (HostReq* r : batch) {
                        if (e == hipSuccess) e = hipMemcpyAsync(r->out_ids, o_dev + (size_t)q0 * k * sizeof(long long), (size_t)r->nq * k * sizeof(long long), hipMemcpyDeviceToHost, st);
                        if (e == hipSuccess) e = hipMemcpyAsync(r->out_scores, o_dev + ib + (size_t)q0 * k * sizeof(float), (size_t)r->nq * k * sizeof(float), hipMemcpyDeviceToHost, st);
                        if (e == hipSuccess) e = hipMemcpyAsync(r->out_counts, o_dev + ib + sb + (size_t)q0 * sizeof(int), (size_t)r->nq * sizeof(int), hipMemcpyDeviceToHost, st);
                        q0 += r->nq;
                    }
                }
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) rc = fail(VDB_EHIP, std::string("result download: ") + hipGetErrorString(e));
            } else {
                (void)hipStreamSynchronize(st);
            }
        }
        if (rc == VDB_OK && pinned) {
            int64_t q0 = 0;
            for (HostReq* r : batch) {
                std::memcpy(r->out_ids, ho + (size_t)q0 * k * sizeof(long long), (size_t)r->nq * k * sizeof(long long));
                std::memcpy(r->out_scores, ho + ib + (size_t)q0 * k * sizeof(float), (size_t)r->nq * k * sizeof(float));
                std::memcpy(r->out_counts, ho + ib + sb + (size_t)q0 * sizeof(int), (size_t)r->nq * sizeof(int));
                q0 += r->nq;
            }
        }
    } while (0);
    ws->ne_ext = nullptr; ws->defer_exact = false; ws->deferred.pending = false; ws->host_flag = nullptr; ws->host_flag_used = false;
    release_ws(ws, st);
    for (HostReq* r : batch) { r->rc = rc; if (rc != VDB_OK) r->err = tl_error; }
}

}  // namespace vdb

// Blocking calls from several host threads are merged: one scan of the table saturates HBM, so N
// concurrent single-query calls would take N scans one after the other, while one call with N
// queries takes about the time of 1.2 scans on the batched path and returns the same ids and
// scores.  There is no waiting window: the first caller runs at once; callers that arrive while
// it runs queue up, and the next one to get the turn takes everything queued for the same table
// and k along with it (up to 1024 queries).  Option "coalesce_host" 0 turns this off.
int vdb_search(const char* name, int64_t nq, const float* queries, int k,
               int64_t* out_ids, float* out_scores, int32_t* out_counts) {
    int rc = enter();
    if (rc) return rc;
    if ((rc = check_search_args(name, nq, queries, k, out_ids, out_scores, out_counts))) return rc;
    HostReq req;
    req.t = find_table(name);
    if (!req.t) return fail(VDB_ENOTFOUND, std::string("no such table: ") + name);
    if (nq == 0) return VDB_OK;
    req.k = k; req.nq = nq; req.queries = queries;
    req.out_ids = out_ids; req.out_scores = out_scores; req.out_counts = out_counts;
    std::vector<HostReq*> batch;
    if (!G.coalesce_host) {
        try {
            batch.push_back(&req);
            search_host_batch(batch);
        } catch (...) {
            return fail(VDB_ENOMEM, "out of host memory");
        }
        return req.rc;
    }
    req.key = MergeKey(req.t->gen, k);
    req.weight = nq;
    static CallMerger<HostReq> merger;       // calls with different (table, k) never wait for each other
    merger.submit(req, 1024, [](std::vector<HostReq*>& b) { search_host_batch(b); },
                  [](std::vector<HostReq*>& b) { for (HostReq* r : b) { r->rc = VDB_ENOMEM; r->err = "out of host memory while merging searches"; } });
    return req.rc == VDB_OK ? VDB_OK : fail(req.rc, req.err);
}

int vdb_merge_topk_dev(int nparts, int64_t nq, int k, int metric, const int64_t* in_ids_dev,
                       const float* in_scores_dev, const int32_t* in_counts_dev,
                       int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev, void* stream) {
    int rc = enter();
    if (rc) return rc;
    if (nparts < 1 || nq < 0 || k < 1 || k > VDB_MAX_TOPK) return fail(VDB_EINVAL, "bad argument");
    if ((long long)nparts * k > 8192) return fail(VDB_EINVAL, "nparts * k exceeds 8192");
    if (nq == 0) return VDB_OK;
    if (!in_ids_dev || !in_scores_dev || !in_counts_dev || !out_ids_dev || !out_scores_dev || !out_counts_dev)
        return fail(VDB_EINVAL, "null pointer");
    VDB_HIP(launch_merge_topk(nparts, nq, k, metric,
                              in_ids_dev, (size_t)nq * k * sizeof(long long),
                              in_scores_dev, (size_t)nq * k * sizeof(float),
                              in_counts_dev, (size_t)nq * sizeof(int),
                              (long long*)out_ids_dev, out_scores_dev, out_counts_dev, (hipStream_t)stream));
    return VDB_OK;
}

int64_t vdb_record_bytes(int64_t nq, int k) {
    const int64_t b = nq * k * 12 + nq * 4;
    return (b + 15) / 16 * 16;
}

int vdb_merge_records_dev(int nparts, int64_t nq, int k, int metric, const void* records_dev,
                          int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev, void* stream) {
    int rc = enter();
    if (rc) return rc;
    if (nparts < 1 || nq < 0 || k < 1 || k > VDB_MAX_TOPK) return fail(VDB_EINVAL, "bad argument");
    if ((long long)nparts * k > 8192) return fail(VDB_EINVAL, "nparts * k exceeds 8192");
    if (nq == 0) return VDB_OK;
    if (!records_dev || !out_ids_dev || !out_scores_dev || !out_counts_dev) return fail(VDB_EINVAL, "null pointer");
    const size_t stride = (size_t)vdb_record_bytes(nq, k);
    const char* base = (const char*)records_dev;
    VDB_HIP(launch_merge_topk(nparts, nq, k, metric, base, stride, base + (size_t)nq * k * 8, stride,
                              base + (size_t)nq * k * 12, stride,
                              (long long*)out_ids_dev, out_scores_dev, out_counts_dev, (hipStream_t)stream));
    return VDB_OK;
}

int vdb_bench_read_dev(const void* buf_dev, int64_t bytes, int nt, int blocks, int threads, void* stream) {
    int rc = enter();
    if (rc) return rc;
    if (!buf_dev || bytes < 16 || blocks < 1 || threads < 64 || threads > 1024 || (threads & 63))
        return fail(VDB_EINVAL, "bad argument");
    VDB_HIP(launch_read_bw(buf_dev, bytes, nt, blocks, threads, (unsigned int*)G.d_counters, (hipStream_t)stream));
    return VDB_OK;
}

#ifdef VDB_FIN_STAMPS
// diagnostic build only: raw counter block (finalize stage stamps live at [8..])
int vdb_debug_read(long long* out32) {
    (void)hipDeviceSynchronize();
    return hipMemcpy(out32, G.d_counters, 32 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -5;
}
#endif

int vdb_prof_enable(int on) {
    G.prof_on.store(on > 0 ? on : 0);
    G.prof_seq.store(0);
    return VDB_OK;
}

int vdb_prof_read(double* scan_ms, int64_t* scan_launches) {
    int rc = enter();
    if (rc) return rc;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    {
        std::lock_guard<std::mutex> lk(G.prof_mu);
        evs.swap(G.prof_events);
    }
    double ms = 0.0;
    for (auto& ev : evs) {
        VDB_HIP(hipEventSynchronize(ev.second));
        float t = 0.f;
        VDB_HIP(hipEventElapsedTime(&t, ev.first, ev.second));
        ms += t;
    }
    if (scan_ms) *scan_ms = ms;
    if (scan_launches) *scan_launches = (int64_t)evs.size();
    std::lock_guard<std::mutex> lk(G.prof_mu);
    for (auto& ev : evs) G.prof_pool.push_back(ev);
    return VDB_OK;
}

int vdb_stats(int64_t* queries, int64_t* exact_fallbacks, int64_t* batched_queries) {
    int rc = enter();
    if (rc) return rc;
    long long h[2] = {0, 0};
    VDB_HIP(hipDeviceSynchronize());
    VDB_HIP(hipMemcpy(h, G.d_counters, sizeof(h), hipMemcpyDeviceToHost));
    if (queries) *queries = h[0];
    if (exact_fallbacks) *exact_fallbacks = h[1];
    if (batched_queries) *batched_queries = G.batched_queries.load();
    return VDB_OK;
}

}  // extern "C"
