// Internal declarations shared by the kernel translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vdb.h"

namespace vdb {

// One prepared query: what the certificate of the fast pass needs besides the vector itself.
struct QInfo {
    double n2;     // canonical fp64 squared norm of the prepared query
    float  bq;     // upper bound of its Euclidean norm
    float  pad;
};

// Per-row error term of the L2 fast keys.  A fast key 2 x.q - ||x||^2 is off the canonical one by at
// most e(x) = a ||x||^2 + (brel bq + cabs) ||x|| (bq: the largest query norm of the launch).  A pass
// that is given this record adds e(x) to every key, so that a key bounds its canonical key from above
// and the finalize steps never need a table-wide norm bound: one huge row does not loosen the
// certificate of the others.
struct RowBound {
    const QInfo* qinfo = nullptr;    // the launch's prepared queries; null: keys carry no error term
    int   nq = 0;
    float a = 0.f, brel = 0.f, cabs = 0.f;     // rounded up, fp32 evaluation error included
};

// Geometry of a stored table as the kernels see it.
struct TableView {
    const char*      rows;      // [n][m*16 bytes], row-major, zero padded to a 16-byte multiple
    const long long* ids;       // [n]
    const float*     n2;        // [n] fp32 squared norm of the stored row
    const float*     max_n2;    // device scalar: max over rows of n2
    const unsigned short* rows16; // fp32 tables: bf16 mirror [n][ld16] of the stored rows (may be null);
                                //            16-bit tables: the rows themselves
    long long        ld16;      // elements per row of rows16 (fp32 tables: a multiple of 32, zero padded)
    long long        n;
    int              d;         // logical dimension
    int              m;         // 16-byte units per row
    int              dt;        // VDB_F32 / VDB_BF16 / VDB_F16
    int              metric;    // VDB_METRIC_*
    RowBound         rb;        // L2 searches: per-row error term of this call's passes
    // IP tables with a few rows far longer than the rest (see OutlierView): the batched path's bounds are
    // taken over the other rows, and every query re-scores the long ones canonically
    float            out_limit = __builtin_huge_valf();   // rows with n2 above this are on the list (inf: no list)
    const long long* outliers = nullptr;                  // their row indices, ascending by arrival
    int              n_out = 0;
    float            max_rest = 0.f;                       // largest n2 among the rows not on the list
};

// Norm statistics of an IP table (kernels_store.hip): a histogram of the exponents of the rows' squared
// norms over a sample of the rows (every row below 4096, every 64th beyond) gives the typical norm;
// rows whose squared norm exceeds `limit` are collected on a list, up to OUTLIER_CAP of them.
constexpr int OUTLIER_CAP = 1024;
hipError_t launch_norm_hist(const float* n2, long long row0, long long n, unsigned int* hist256, hipStream_t st);
// appends the rows of [row0, row0 + n) with n2 > limit to list (count in *n_out, entries past OUTLIER_CAP
// are dropped but counted) and raises *max_rest to the largest n2 <= limit among them
hipError_t launch_collect_outliers(const float* n2, long long row0, long long n, float limit, long long* list,
                                   int* n_out, float* max_rest, hipStream_t st);

#if defined(__HIPCC__)
// (brel bq + cabs) for the launch; every wave evaluates it by itself
__device__ __forceinline__ float rowbound_s(const RowBound& rb, int lane) {
    float m = 0.f;
    for (int i = lane; i < rb.nq; i += 64) m = fmaxf(m, rb.qinfo[i].bq);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return fmaf(rb.brel, m, rb.cabs);
}
// ||x||^2 - e(x): what an L2 pass subtracts from 2 x.q
__device__ __forceinline__ float rowbound_adj(float n2, float a, float s) {
    // the hardware square root (1 ulp, subnormal arguments read as zero) is inside the margin of s
    return fmaf(-a, n2, n2) - s * __builtin_amdgcn_sqrtf(n2);
}
#endif

inline int elems_per_unit(int dt) { return dt == VDB_F32 ? 4 : 8; }

// Candidate-list sizing for a given k (host side).
struct SelGeom {
    int K;   // candidates kept per list (k plus a safety margin for the fp32 scoring error)
    int C;   // list capacity in LDS (power of two, >= 2K and >= K+64)
    int W;   // waves per scan block
};
SelGeom sel_geom(int k);

// Scratch memory of one search call (device pointers).
struct SearchScratch {
    float*     qprep;       // [nq][dpad]
    QInfo*     qinfo;       // [nq]
    float*     cand_s;      // [nq][G][K]
    long long* cand_t;      // [nq][G][K]
    int*       need_exact;  // [nq]
    long long* counters;    // [2]: queries, exact fallbacks (library-wide, not per scratch)
};

void set_scan_tune(int nt, int waves, int blocks_per_cu, int contiguous, int dynamic);
void set_scan_short_bytes(double bytes);
void set_fin_threads(int v);           // threads of the lists-mode finalize launch (256, 512 or 1024)
hipError_t launch_read_bw(const void* buf, long long bytes, int nt, int blocks, int threads,
                          unsigned int* sink, hipStream_t st);

// ---- launchers (all asynchronous on `st`) -------------------------------------------------
// Also empties two lists of flagged queries, {count, query indices...} of nq + 1 ints each, one behind
// the other at need_exact: the one the finalize step fills for the exact pass, and a second one for a
// pass in between.  `only` arguments below take such a list and restrict a launch to its queries.
hipError_t launch_prep_query(const float* q_raw, int64_t nq, int d, int dpad, int metric,
                             float* qprep, QInfo* qinfo, int* need_exact, hipStream_t st);

// Fast pass: fp32 scoring of every row, top-K candidates per (query, block).
// Returns the grid width G through *out_G (cand arrays must hold nq*G*K entries; use scan_grid()).
int  scan_grid(const TableView& t, const SelGeom& g, int n_cu);
hipError_t launch_scan(const TableView& t, const float* qprep, int64_t nq, int dpad,
                       const SelGeom& g, int G, float* cand_s, long long* cand_t, float* cand_h,
                       const float* floor, const int* only, hipStream_t st);

// Terms a fast pass adds to the plain fp32 scoring-error bound of the certificate.
struct FlatBound {
    double l2_bias = 0.0;     // 1 when L2 keys are accumulated on top of the row norm (partial sums of that size)
    double dd_mult = 1.0;     // accumulation chain length in units of d (2 when hi and lo parts are summed)
    double extra_rel = 0.0;   // relative error of the query as the MFMA sees it (x ||x|| ||q||)
    double extra_abs = 0.0;   // absolute error per unit of ||x|| (fp16 subnormal spacing)
    int    force_exact = 0;   // measurement: every query is sent to the exact pass
};
// The RowBound coefficients that cover the same error terms for a table of dimension d.
RowBound l2_row_bound(const FlatBound& fb, int d, const QInfo* qinfo, int nq);

// Merge candidates, re-score them canonically (fp64), order by (score, id), certify.
hipError_t launch_finalize(const TableView& t, const float* qprep, const QInfo* qinfo, int64_t nq,
                           int dpad, const SelGeom& g, int G, int k,
                           const float* cand_s, const long long* cand_t, const float* cand_h,
                           long long* out_ids, float* out_scores, int* out_counts,
                           int* need_exact, long long* counters, const FlatBound& fb, const int* only, hipStream_t st,
                           int* host_flag = nullptr);

// Same for the batched path: one unsorted candidate array of `cap` slots per query, `cnt[q]`
// appended, every non-candidate row known to have a fast key <= tau[q].
hipError_t launch_lists_floor(const SelGeom& g, int G, int64_t nq, const float* cand_s, const long long* cand_t,
                              const float* cand_h, float* floor_out, hipStream_t st);
hipError_t launch_finalize_flat(const TableView& t, const float* qprep, const QInfo* qinfo, int64_t nq,
                                int dpad, const SelGeom& g, int k,
                                const float* cand_s, const long long* cand_t, const int* cnt, int cap,
                                const float* tau, const int* lost, const int* bad, const FlatBound& fb,
                                long long* out_ids, float* out_scores, int* out_counts,
                                int* need_exact, long long* counters, hipStream_t st);

// Batched path: MFMA GEMM over the row tiles tile0 + t*tile_stride (t < ntiles) of 128 rows each,
// appending every score above thr[q] to query q's candidate array.
hipError_t launch_gemm_thresh(const TableView& t, const float* qprep, int nq, int dpad, const float* thr,
                              long long tile0, long long tile_stride, long long ntiles,
                              float* cand_s, long long* cand_t, int* cnt, int* lost, int cap, int dense,
                              int n_cu, hipStream_t st);
hipError_t launch_gemm16_thresh(const TableView& t, const unsigned short* qhi, const unsigned short* qlo, int qstride,
                                int split, int nq, int dpad, const float* thr,
                                long long tile_stride, long long ntiles,
                                float* cand_s, long long* cand_t, int* cnt, int* lost, int cap, int dense,
                                int n_cu, hipStream_t st);
void set_gemm32s_threads(int v);
// Ping-pong kernels (gemm_pp384.h: 384 x 256 tiles, for launches of up to 256 queries; gemm_pp.h: 256 x 256)
// over t.rows16 with the queries as a tiled 16-bit image (launch_tile_query16); visits every tile_stride-th
// tile of pp_tile_rows(nq) rows, tile0 and tile_count in units of visited tiles.  Needs nks * 32 <= t.ld16
// or, for 16-bit tables, dpad a multiple of 32 (pp_eligible).
void set_pp_tile_rows(int v);
int pp_tile_rows(int nq);
bool pp_eligible(const TableView& t, int dpad);
size_t pp_query_image_bytes(int nq, int dpad);
hipError_t launch_tile_query16(int dt, int l2, const float* qprep, int nq, int dpad, unsigned short* image,
                               int* bad, hipStream_t st);
hipError_t launch_gemm_pp(const TableView& t, const unsigned short* qimage, int nq, int dpad, const float* thr,
                          long long tile_stride, long long tile0, long long tile_count,
                          float* cand_s, long long* cand_t, int* cnt, int* lost, int cap,
                          int n_cu, hipStream_t st);
hipError_t launch_gemm32s(const TableView& t, const unsigned short* qhi, const unsigned short* qlo, int qstride,
                          int split, int nq, int dpad, const float* thr, long long tile_stride,
                          float* cand_s, long long* cand_t,
                          int* cnt, int* lost, int cap, int n_cu, hipStream_t st);
hipError_t launch_split_query16(int dt, const float* qprep, int nq, int dpad, int stride_out, unsigned short* hi,
                                unsigned short* lo, int* bad, hipStream_t st);
// thr_out[q] = max(thr_in[q], K-th best key among query q's candidates); candidates of query q start
// at cand_s + q*stride and number cnt[q] (capped at cap), or count_all when cnt is null.
// With `t` given, candidates whose row is on the table's list of long rows (n2 above t->out_limit) do not
// count: their fast keys say little.  cand_t null: slot i holds row i.
hipError_t launch_kth_threshold(const float* cand_s, long long stride, const int* cnt, long long count_all,
                                int cap, int K, const float* thr_in, float* thr_out, int nq, hipStream_t st,
                                const TableView* t = nullptr, const long long* cand_t = nullptr);

// Exact second pass for the queries whose fast pass could not be certified.
hipError_t launch_exact(const TableView& t, const float* qprep, int64_t nq, int dpad,
                        const SelGeom& g, int G, int k, float* cand_s, long long* cand_t,
                        long long* out_ids, float* out_scores, int* out_counts,
                        int* need_exact, long long* counters, hipStream_t st);

// Append n rows (raw fp32 from `raw`, or synthetic when raw == nullptr) at row index `row0`.
// *bad (host-visible) is set when a row is not finite in the storage type
hipError_t launch_append(char* rows, unsigned short* rows16, long long ld16, long long* ids, float* n2, float* max_n2,
                         int* bad, long long row0, long long n, int d, int m, int dt, int metric,
                         const float* raw, const long long* raw_ids,
                         uint64_t seed, long long synth_first_row, hipStream_t st);

// table_max = max(table_max, *pending) unless *bad is set; bad[1] = bits of table_max afterwards (see fold_max_kernel)
hipError_t launch_fold_max(float* table_max, const float* pending, int* bad, hipStream_t st);

// (re)builds the bf16 mirror of rows [row0, row0 + n) of an fp32 table from the stored rows
hipError_t launch_build_mirror(const char* rows, unsigned short* rows16, long long ld16, long long row0,
                               long long n, int m, hipStream_t st);

hipError_t launch_synth_fill(float* out, long long n, int d, uint64_t seed, long long first_row,
                             hipStream_t st);

hipError_t launch_fetch_rows(const char* rows, long long first, long long n, int d, int m, int dt,
                             float* out, hipStream_t st);

hipError_t launch_merge_topk(int nparts, int64_t nq, int k, int metric,
                             const void* in_ids, size_t ids_stride,
                             const void* in_scores, size_t scores_stride,
                             const void* in_counts, size_t counts_stride,
                             long long* out_ids, float* out_scores, int* out_counts,
                             hipStream_t st);

}  // namespace vdb
