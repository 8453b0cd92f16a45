// Single-query / small-batch path: a bandwidth-bound scan of the stored matrix fused with
// partial top-K selection, followed by canonical (fp64) re-scoring of the surviving candidates.
//
//   prep_query   normalise (COSINE) and zero-pad the query, record its norm
//   scan         every wave streams rows with 16-byte loads straight to registers, reduces the
//                fp32 dot product across the lanes of a row and keeps its K best rows in LDS;
//                the block merges its waves' lists and writes K candidates.  Scores never go
//                to HBM.
//   finalize     one block per query merges the G*K candidates, re-scores the K survivors in
//                the canonical fp64 order, sorts by (score, id) and proves that no row outside
//                the candidate set can belong to the answer (fp32 error bound); otherwise it
//                flags the query for the exact pass.
//   exact_*      the same scan with canonical fp64 scores and the (score, id) order throughout:
//                always right, used only for flagged queries (massive ties, adversarial data).
#include "vdb_internal.h"
#include "select.h"

namespace vdb {

// ------------------------------------------------------------------------------ helpers
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float f16_bits(uint32_t h) {
    const unsigned short b = (unsigned short)h;
    _Float16 v;
    __builtin_memcpy(&v, &b, 2);
    return (float)v;
}

// One stored element as fp32 (exact for every storage dtype).
__device__ __forceinline__ float load_elem(const char* row, int j, int dt) {
    if (dt == VDB_F32) return ((const float*)row)[j];
    const unsigned short b = ((const unsigned short*)row)[j];
    if (dt == VDB_BF16) return __uint_as_float((uint32_t)b << 16);
    return f16_bits(b);
}

template <int DT>
__device__ __forceinline__ float dot_unit(const uint4& x, const float* q, float acc) {
    if constexpr (DT == VDB_F32) {
        acc = fmaf(__uint_as_float(x.x), q[0], acc);
        acc = fmaf(__uint_as_float(x.y), q[1], acc);
        acc = fmaf(__uint_as_float(x.z), q[2], acc);
        acc = fmaf(__uint_as_float(x.w), q[3], acc);
    } else if constexpr (DT == VDB_BF16) {
        acc = fmaf(bf16_lo(x.x), q[0], acc); acc = fmaf(bf16_hi(x.x), q[1], acc);
        acc = fmaf(bf16_lo(x.y), q[2], acc); acc = fmaf(bf16_hi(x.y), q[3], acc);
        acc = fmaf(bf16_lo(x.z), q[4], acc); acc = fmaf(bf16_hi(x.z), q[5], acc);
        acc = fmaf(bf16_lo(x.w), q[6], acc); acc = fmaf(bf16_hi(x.w), q[7], acc);
    } else {
        acc = fmaf(f16_bits(x.x), q[0], acc); acc = fmaf(f16_bits(x.x >> 16), q[1], acc);
        acc = fmaf(f16_bits(x.y), q[2], acc); acc = fmaf(f16_bits(x.y >> 16), q[3], acc);
        acc = fmaf(f16_bits(x.z), q[4], acc); acc = fmaf(f16_bits(x.z >> 16), q[5], acc);
        acc = fmaf(f16_bits(x.w), q[6], acc); acc = fmaf(f16_bits(x.w >> 16), q[7], acc);
    }
    return acc;
}

__device__ __forceinline__ double wave_tree_sum(double acc) {
    // fixed binary tree of the canonical score: lane 0 ends with the tree value
    for (int s = 32; s >= 1; s >>= 1) acc = __dadd_rn(acc, __shfl_xor(acc, s));
    return acc;
}

// Canonical fp64 score of one stored row against a prepared query, computed by one wave.
// Every lane returns the same value.
__device__ __forceinline__ double canon_score(const char* row, const float* q, int d, int dt,
                                              bool l2, int lane) {
#pragma clang fp contract(off)
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) {
        const double x = (double)load_elem(row, j, dt);
        const double y = (double)q[j];
        if (l2) {
            const double df = __dsub_rn(x, y);
            acc = __dadd_rn(acc, __dmul_rn(df, df));
        } else {
            acc = __dadd_rn(acc, __dmul_rn(x, y));   // product exact: same as an fma
        }
    }
    return wave_tree_sum(acc);
}

// Kernels that ask for more than the default 64 KiB of dynamic LDS must say so first.
template <typename F>
static hipError_t allow_lds(F kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ------------------------------------------------------------------------------ prep_query
__global__ void __launch_bounds__(64)
prep_query_kernel(const float* __restrict__ q_raw, int d, int dpad, int metric,
                  float* __restrict__ qprep, QInfo* __restrict__ qinfo) {
    const int lane = threadIdx.x;
    const float* q = q_raw + (size_t)blockIdx.x * d;
    float* out = qprep + (size_t)blockIdx.x * dpad;
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) { const double x = (double)q[j]; acc = __dadd_rn(acc, __dmul_rn(x, x)); }
    double n2 = wave_tree_sum(acc);
    n2 = __shfl(n2, 0);
    if (metric == VDB_METRIC_COSINE) {
        const double inv = (n2 == 0.0) ? 0.0 : 1.0 / sqrt(n2);
        double acc2 = 0.0;
        for (int j = lane; j < dpad; j += 64) {
            float y = 0.f;
            if (j < d) y = (float)__dmul_rn((double)q[j], inv);
            out[j] = y;
            acc2 = __dadd_rn(acc2, __dmul_rn((double)y, (double)y));
        }
        n2 = __shfl(wave_tree_sum(acc2), 0);
    } else {
        for (int j = lane; j < dpad; j += 64) out[j] = (j < d) ? q[j] : 0.f;
    }
    if (lane == 0) {
        QInfo qi;
        qi.n2 = n2;
        qi.bq = (float)(sqrt(n2) * (1.0 + 1e-6)) * 1.000001f;
        qi.pad = 0.f;
        qinfo[blockIdx.x] = qi;
    }
}

hipError_t launch_prep_query(const float* q_raw, int64_t nq, int d, int dpad, int metric,
                             float* qprep, QInfo* qinfo, hipStream_t st) {
    hipLaunchKernelGGL(prep_query_kernel, dim3((unsigned)nq), dim3(64), 0, st,
                       q_raw, d, dpad, metric, qprep, qinfo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ scan
struct ScanParams {
    const char*  rows;
    const float* n2;
    const float* q;         // [nq][dpad]
    long long    n;
    int          m;
    int          dpad;
    int          K, C;
    int          l2;
    float*       cand_s;    // [nq][G][K]
    long long*   cand_t;
};

template <int NCH> struct ScanUnroll { static constexpr int U = (NCH <= 1) ? 8 : (NCH <= 3) ? 4 : (NCH <= 6) ? 2 : 1; };

// LPR lanes share one row per wave-instruction (64/LPR rows per instruction); a row takes NCH
// such instructions (NCH > 1 only with LPR == 64).  Every load is unconditional: lanes past the
// end of a row re-read its first 16 bytes against a zero query, rows past the end of the table
// re-read the last row and are discarded.
template <int DT, int LPR, int NCH>
__global__ void __launch_bounds__(1024)
scan_kernel(ScanParams p) {
    constexpr int EPL = (DT == VDB_F32) ? 4 : 8;
    constexpr int RPI = 64 / LPR;
    constexpr int U = ScanUnroll<NCH>::U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lir = lane & (LPR - 1);
    const int grp = lane / LPR;
    const float* q = p.q + (size_t)blockIdx.y * p.dpad;

    int   off[NCH];
    float qv[NCH][EPL];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int unit = c * LPR + lir;
        const bool act = unit < p.m;
        off[c] = act ? unit * 16 : 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) qv[c][e] = act ? q[unit * EPL + e] : 0.f;
    }

    WaveSel sel;
    sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane);

    const long long TW = (long long)gridDim.x * W;
    const long long gw = (long long)blockIdx.x * W + wave;
    constexpr long long RPS = (long long)RPI * U;            // rows per wave step
    const long long nsteps = (p.n + TW * RPS - 1) / (TW * RPS);
    const size_t rowbytes = (size_t)p.m * 16;
    const long long last = p.n - 1;

    for (long long st = 0; st < nsteps; ++st) {
        const long long base = (st * TW + gw) * RPS;
        uint4 x[U][NCH];
        float nn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = base + u * RPI + grp;
            const long long rc = r < last ? r : last;
            const char* ptr = p.rows + (size_t)rc * rowbytes;
#pragma unroll
            for (int c = 0; c < NCH; ++c) x[u][c] = *(const uint4*)(ptr + off[c]);
            nn[u] = p.l2 ? p.n2[rc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc = dot_unit<DT>(x[u][c], qv[c], acc);
#pragma unroll
            for (int s = LPR >> 1; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
            const long long r = base + u * RPI + grp;
            const float key = p.l2 ? (2.f * acc - nn[u]) : acc;
            const bool pass = (r <= last) && (lir == 0) && better(key, r, sel.thr_s, sel.thr_t);
            sel.push(pass, key, r, lane);
        }
    }

    block_merge(sel, S_all, T_all, W, wave, lane);
    if (wave == 0) {
        const size_t o = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * p.K;
        for (int i = lane; i < p.K; i += 64) { p.cand_s[o + i] = S_all[i]; p.cand_t[o + i] = T_all[i]; }
    }
}

// Generic row length: NCH decided at run time, query chunks read from LDS.
template <int DT>
__global__ void __launch_bounds__(1024)
scan_kernel_generic(ScanParams p) {
    constexpr int EPL = (DT == VDB_F32) ? 4 : 8;
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));
    float*     qs    = (float*)(smem + (size_t)W * p.C * (sizeof(float) + sizeof(long long)));

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float* q = p.q + (size_t)blockIdx.y * p.dpad;
    const int nch = (p.m + 63) >> 6;
    // query staged as nch chunks of 64 units, zero beyond the row
    for (int i = threadIdx.x; i < nch * 64 * EPL; i += blockDim.x) qs[i] = (i < p.dpad) ? q[i] : 0.f;
    __syncthreads();

    WaveSel sel;
    sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane);

    const long long TW = (long long)gridDim.x * W;
    const long long gw = (long long)blockIdx.x * W + wave;
    const long long nsteps = (p.n + TW * U - 1) / (TW * U);
    const size_t rowbytes = (size_t)p.m * 16;
    const long long last = p.n - 1;

    for (long long st = 0; st < nsteps; ++st) {
        const long long base = (st * TW + gw) * U;
        float acc[U];
        const char* ptr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = base + u;
            const long long rc = r < last ? r : last;
            ptr[u] = p.rows + (size_t)rc * rowbytes;
            acc[u] = 0.f;
        }
        for (int c = 0; c < nch; ++c) {
            const int unit = c * 64 + lane;
            const int o = (unit < p.m) ? unit * 16 : 0;
            float qv[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) qv[e] = qs[unit * EPL + e];
            uint4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = *(const uint4*)(ptr[u] + o);
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = dot_unit<DT>(x[u], qv, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = acc[u];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) a += __shfl_xor(a, s);
            const long long r = base + u;
            const long long rc = r < last ? r : last;
            const float key = p.l2 ? (2.f * a - p.n2[rc]) : a;
            const bool pass = (r <= last) && (lane == 0) && better(key, r, sel.thr_s, sel.thr_t);
            sel.push(pass, key, r, lane);
        }
    }

    block_merge(sel, S_all, T_all, W, wave, lane);
    if (wave == 0) {
        const size_t o = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * p.K;
        for (int i = lane; i < p.K; i += 64) { p.cand_s[o + i] = S_all[i]; p.cand_t[o + i] = T_all[i]; }
    }
}

// Shape of the specialised kernel chosen for a row of m 16-byte units.
struct ScanShape { int lpr, nch; bool generic; };
static ScanShape scan_shape(int m, int dt) {
    ScanShape s{64, 1, false};
    if (m <= 64) {
        int l = 1;
        while (l < m) l <<= 1;
        s.lpr = l; s.nch = 1;
        return s;
    }
    const int nch = (m + 63) / 64;
    s.lpr = 64; s.nch = nch;
    if (!(nch == 2 || nch == 3 || nch == 4 || nch == 6 || nch == 8)) s.generic = true;
    if (dt != VDB_F32 && nch > 4) s.generic = true;    // 8 query registers per chunk: would spill
    return s;
}

static int scan_rows_per_block_step(const ScanShape& s, int W) {
    if (s.generic) return W * 4;
    const int u = (s.nch <= 1) ? 8 : (s.nch <= 3) ? 4 : (s.nch <= 6) ? 2 : 1;
    return W * (64 / s.lpr) * u;
}

int scan_grid(const TableView& t, const SelGeom& g, int n_cu) {
    const ScanShape s = scan_shape(t.m, t.dt);
    const long long per = scan_rows_per_block_step(s, g.W);
    long long need = (t.n + per - 1) / per;
    if (need < 1) need = 1;
    // LDS and the 1024-thread bound admit one 16-wave block per CU; smaller blocks pack more.
    const long long cap = (long long)n_cu * (g.W >= 16 ? 1 : (16 / g.W));
    return (int)(need < cap ? need : cap);
}

template <int DT>
static hipError_t launch_scan_dt(const ScanShape& s, dim3 grid, dim3 block, size_t lds,
                                 const ScanParams& p, hipStream_t st) {
#define VDB_SCAN_CASE(L, N)                                                                       \
    if (s.lpr == L && s.nch == N) {                                                               \
        hipLaunchKernelGGL((scan_kernel<DT, L, N>), grid, block, lds, st, p);                     \
        return hipGetLastError();                                                                 \
    }
    if (!s.generic) {
        VDB_SCAN_CASE(1, 1) VDB_SCAN_CASE(2, 1) VDB_SCAN_CASE(4, 1) VDB_SCAN_CASE(8, 1)
        VDB_SCAN_CASE(16, 1) VDB_SCAN_CASE(32, 1) VDB_SCAN_CASE(64, 1)
        VDB_SCAN_CASE(64, 2) VDB_SCAN_CASE(64, 3) VDB_SCAN_CASE(64, 4)
        if constexpr (DT == VDB_F32) { VDB_SCAN_CASE(64, 6) VDB_SCAN_CASE(64, 8) }
    }
#undef VDB_SCAN_CASE
    const int nch = (p.m + 63) / 64;
    const size_t ldsg = lds + (size_t)nch * 64 * ((DT == VDB_F32) ? 4 : 8) * sizeof(float);
    hipError_t e = allow_lds(scan_kernel_generic<DT>, ldsg);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((scan_kernel_generic<DT>), grid, block, ldsg, st, p);
    return hipGetLastError();
}

hipError_t launch_scan(const TableView& t, const float* qprep, int64_t nq, int dpad,
                       const SelGeom& g, int G, float* cand_s, long long* cand_t, hipStream_t st) {
    ScanParams p;
    p.rows = t.rows; p.n2 = t.n2; p.q = qprep; p.n = t.n; p.m = t.m; p.dpad = dpad;
    p.K = g.K; p.C = g.C; p.l2 = (t.metric == VDB_METRIC_L2) ? 1 : 0;
    p.cand_s = cand_s; p.cand_t = cand_t;
    const ScanShape s = scan_shape(t.m, t.dt);
    const dim3 grid((unsigned)G, (unsigned)nq), block((unsigned)g.W * 64);
    const size_t lds = (size_t)g.W * g.C * (sizeof(float) + sizeof(long long));
    switch (t.dt) {
        case VDB_F32:  return launch_scan_dt<VDB_F32>(s, grid, block, lds, p, st);
        case VDB_BF16: return launch_scan_dt<VDB_BF16>(s, grid, block, lds, p, st);
        default:       return launch_scan_dt<VDB_F16>(s, grid, block, lds, p, st);
    }
}

// ------------------------------------------------------------------------------ finalize
struct FinParams {
    TableView        t;
    const float*     q;          // [nq][dpad]
    const QInfo*     qinfo;
    int              dpad;
    int              K, C, G, k;
    const float*     cand_s;
    const long long* cand_t;
    long long*       out_ids;
    float*           out_scores;
    int*             out_counts;
    int*             need_exact;
    long long*       counters;
};

__device__ __forceinline__ int pow2_ceil_dev(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__global__ void __launch_bounds__(1024)
finalize_kernel(FinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));
    const int KR = pow2_ceil_dev(p.K);
    long long* RT = (long long*)(smem + (size_t)W * p.C * (sizeof(float) + sizeof(long long)));
    float*     RS = (float*)(RT + KR);
    int*       s_nvalid = (int*)(RS + KR);      // all LDS lives in the one dynamic region

    const int qi = blockIdx.x;
    const long long ncand = (long long)p.G * p.K;
    block_topk_global(p.cand_s + (size_t)qi * ncand, p.cand_t + (size_t)qi * ncand, ncand,
                      S_all, T_all, p.C, p.K, W, wave, lane);
    // S_all/T_all[0..K): best fast keys with row indices (TAG_NONE = empty), sorted.
    if (threadIdx.x == 0) {
        int nv = 0;
        while (nv < p.K && T_all[nv] != TAG_NONE) ++nv;
        *s_nvalid = nv;
    }
    __syncthreads();
    const int nvalid = *s_nvalid;
    const bool l2 = p.t.metric == VDB_METRIC_L2;
    const float* q = p.q + (size_t)qi * p.dpad;
    const size_t rowbytes = (size_t)p.t.m * 16;
    for (int i = wave; i < KR; i += W) {
        float key = neg_inf();
        long long id = TAG_NONE;
        if (i < nvalid) {
            const long long r = T_all[i];
            const double sc = canon_score(p.t.rows + (size_t)r * rowbytes, q, p.t.d, p.t.dt, l2, lane);
            const float f = (float)sc;
            key = l2 ? -f : f;
            id = p.t.ids[r];
        }
        if (lane == 0) { RS[i] = key; RT[i] = id; }
    }
    __syncthreads();
    if (wave == 0) {
        // ids may legitimately equal TAG_NONE; empties are told apart by position (>= nvalid)
        // because a real entry is never worse than an empty one with the same key and id.
        wave_bitonic_sort(RS, RT, KR, lane);
        const int cnt = nvalid < p.k ? nvalid : p.k;
        for (int i = lane; i < p.k; i += 64) {
            const bool ok = i < cnt;
            p.out_ids[(size_t)qi * p.k + i] = ok ? RT[i] : -1ll;
            const float key = RS[i < KR ? i : 0];
            p.out_scores[(size_t)qi * p.k + i] = ok ? (l2 ? -key : key) : (l2 ? __builtin_huge_valf() : neg_inf());
        }
        if (lane == 0) {
            p.out_counts[qi] = cnt;
            bool cert = nvalid < p.K;         // every stored row was a candidate
            if (!cert && cnt > 0) {
                const double theta = (double)S_all[p.K - 1];
                const double bx = sqrt((double)(*p.t.max_n2)) * (1.0 + 1e-6);
                const double bq = (double)p.qinfo[qi].bq;
                const double u = 5.9604644775390625e-08;   // 2^-24
                const double dd = (double)(p.t.d + 16);
                if (!l2) {
                    const double eps = 1.05 * dd * u * bx * bq + 4.0 * u * fabs(theta) + 1e-37;
                    cert = (double)RS[cnt - 1] > theta + eps;
                } else {
                    const double eps = 1.05 * u * (2.0 * dd * bx * bq + 4.0 * (bx * bx + 2.0 * bx * bq))
                                       + 4.0 * u * fabs(theta) + 1e-37;
                    const double dist_k = -(double)RS[cnt - 1];
                    const double bound = p.qinfo[qi].n2 - theta - eps;     // outside rows are farther than this
                    cert = dist_k * (1.0 + 8.0 * u) + 1e-37 < bound;
                }
            }
            p.need_exact[qi] = cert ? 0 : 1;
            atomicAdd((unsigned long long*)&p.counters[0], 1ull);
        }
    }
}

hipError_t launch_finalize(const TableView& t, const float* qprep, const QInfo* qinfo, int64_t nq,
                           int dpad, const SelGeom& g, int G, int k,
                           const float* cand_s, const long long* cand_t,
                           long long* out_ids, float* out_scores, int* out_counts,
                           int* need_exact, long long* counters, hipStream_t st) {
    FinParams p;
    p.t = t; p.q = qprep; p.qinfo = qinfo; p.dpad = dpad;
    p.K = g.K; p.C = g.C; p.G = G; p.k = k;
    p.cand_s = cand_s; p.cand_t = cand_t;
    p.out_ids = out_ids; p.out_scores = out_scores; p.out_counts = out_counts;
    p.need_exact = need_exact; p.counters = counters;
    int kr = 1; while (kr < g.K) kr <<= 1;
    const size_t lds = (size_t)g.W * g.C * 12 + (size_t)kr * 12 + 16;
    hipError_t e = allow_lds(finalize_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nq), dim3((unsigned)g.W * 64), lds, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ exact pass
struct ExactParams {
    TableView    t;
    const float* q;
    int          dpad;
    int          K, C, k;
    float*       cand_s;
    long long*   cand_t;
    long long*   out_ids;
    float*       out_scores;
    int*         out_counts;
    int*         need_exact;
    long long*   counters;
    int          G;
};

__global__ void __launch_bounds__(1024)
exact_scan_kernel(ExactParams p) {
    const int qi = blockIdx.y;
    if (p.need_exact[qi] == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));
    const bool l2 = p.t.metric == VDB_METRIC_L2;
    const float* q = p.q + (size_t)qi * p.dpad;
    const size_t rowbytes = (size_t)p.t.m * 16;

    WaveSel sel;
    sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane);
    const long long TW = (long long)gridDim.x * W;
    for (long long r = (long long)blockIdx.x * W + wave; r < p.t.n; r += TW) {
        const double sc = canon_score(p.t.rows + (size_t)r * rowbytes, q, p.t.d, p.t.dt, l2, lane);
        const float f = (float)sc;
        const float key = l2 ? -f : f;
        const long long id = p.t.ids[r];
        const bool pass = (lane == 0) && better(key, id, sel.thr_s, sel.thr_t);
        sel.push(pass, key, id, lane);
    }
    block_merge(sel, S_all, T_all, W, wave, lane);
    if (wave == 0) {
        const size_t o = ((size_t)qi * gridDim.x + blockIdx.x) * p.K;
        for (int i = lane; i < p.K; i += 64) { p.cand_s[o + i] = S_all[i]; p.cand_t[o + i] = T_all[i]; }
    }
}

// A stored id may equal TAG_NONE, which block_topk_global treats as "empty"; the exact pass
// therefore merges with its own loop that keeps every entry of the first min(K, rows) per block.
__global__ void __launch_bounds__(1024)
exact_finalize_kernel(ExactParams p) {
    const int qi = blockIdx.x;
    if (p.need_exact[qi] == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));
    const bool l2 = p.t.metric == VDB_METRIC_L2;

    const long long ncand = (long long)p.G * p.K;
    const float* gs = p.cand_s + (size_t)qi * ncand;
    const long long* gt = p.cand_t + (size_t)qi * ncand;
    WaveSel sel;
    sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane);
    for (long long base = (long long)wave * 64; base < ncand; base += (long long)W * 64) {
        const long long i = base + lane;
        float key = neg_inf();
        long long tag = TAG_NONE;
        if (i < ncand) { key = gs[i]; tag = gt[i]; }
        const bool pass = (i < ncand) && better(key, tag, sel.thr_s, sel.thr_t);
        sel.push(pass, key, tag, lane);
    }
    block_merge(sel, S_all, T_all, W, wave, lane);
    if (wave == 0) {
        const long long cnt = p.t.n < p.k ? p.t.n : p.k;
        for (int i = lane; i < p.k; i += 64) {
            const bool ok = i < cnt;
            p.out_ids[(size_t)qi * p.k + i] = ok ? T_all[i] : -1ll;
            const float key = S_all[i];
            p.out_scores[(size_t)qi * p.k + i] = ok ? (l2 ? -key : key) : (l2 ? __builtin_huge_valf() : neg_inf());
        }
        if (lane == 0) {
            p.out_counts[qi] = (int)cnt;
            p.need_exact[qi] = 0;
            atomicAdd((unsigned long long*)&p.counters[1], 1ull);
        }
    }
}

hipError_t launch_exact(const TableView& t, const float* qprep, int64_t nq, int dpad,
                        const SelGeom& g, int G, int k, float* cand_s, long long* cand_t,
                        long long* out_ids, float* out_scores, int* out_counts,
                        int* need_exact, long long* counters, hipStream_t st) {
    ExactParams p;
    p.t = t; p.q = qprep; p.dpad = dpad; p.K = g.K; p.C = g.C; p.k = k;
    p.cand_s = cand_s; p.cand_t = cand_t;
    p.out_ids = out_ids; p.out_scores = out_scores; p.out_counts = out_counts;
    p.need_exact = need_exact; p.counters = counters; p.G = G;
    const size_t lds = (size_t)g.W * g.C * 12;
    hipLaunchKernelGGL(exact_scan_kernel, dim3((unsigned)G, (unsigned)nq), dim3((unsigned)g.W * 64),
                       lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exact_finalize_kernel, dim3((unsigned)nq), dim3((unsigned)g.W * 64), lds, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ shard merge
// One wave per query: nparts*k entries -> top k by (score, id).  Part p's arrays start
// `stride` bytes after part p-1's (dense [nparts][nq][k] arrays or gathered per-rank records).
__global__ void __launch_bounds__(64)
merge_topk_kernel(int nparts, int k, int l2, int cap,
                  const char* __restrict__ in_ids, size_t ids_stride,
                  const char* __restrict__ in_scores, size_t scores_stride,
                  const char* __restrict__ in_counts, size_t counts_stride, long long nq,
                  long long* __restrict__ out_ids, float* __restrict__ out_scores,
                  int* __restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* T = (long long*)smem;
    float*     S = (float*)(T + cap);
    const int lane = threadIdx.x;
    const long long qi = blockIdx.x;
    int total = 0;
    for (int i = lane; i < cap; i += 64) {
        float key = neg_inf();
        long long tag = TAG_NONE;
        if (i < nparts * k) {
            const int part = i / k, j = i - part * k;
            const int cnt = ((const int*)(in_counts + (size_t)part * counts_stride))[qi];
            if (j < cnt) {
                const float sc = ((const float*)(in_scores + (size_t)part * scores_stride))[qi * k + j];
                key = l2 ? -sc : sc;
                tag = ((const long long*)(in_ids + (size_t)part * ids_stride))[qi * k + j];
            }
        }
        S[i] = key; T[i] = tag;
    }
    for (int part = 0; part < nparts; ++part) total += ((const int*)(in_counts + (size_t)part * counts_stride))[qi];
    lds_wave_sync();
    wave_bitonic_sort(S, T, cap, lane);
    const int cnt = total < k ? total : k;
    for (int i = lane; i < k; i += 64) {
        const bool ok = i < cnt;
        out_ids[qi * k + i] = ok ? T[i] : -1ll;
        out_scores[qi * k + i] = ok ? (l2 ? -S[i] : S[i]) : (l2 ? __builtin_huge_valf() : neg_inf());
    }
    if (lane == 0) out_counts[qi] = cnt;
}

hipError_t launch_merge_topk(int nparts, int64_t nq, int k, int metric,
                             const void* in_ids, size_t ids_stride,
                             const void* in_scores, size_t scores_stride,
                             const void* in_counts, size_t counts_stride,
                             long long* out_ids, float* out_scores, int* out_counts,
                             hipStream_t st) {
    int cap = 64;
    while (cap < nparts * k) cap <<= 1;
    const size_t lds = (size_t)cap * 12;
    hipError_t e = allow_lds(merge_topk_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)nq), dim3(64), lds, st,
                       nparts, k, metric == VDB_METRIC_L2 ? 1 : 0, cap,
                       (const char*)in_ids, ids_stride, (const char*)in_scores, scores_stride,
                       (const char*)in_counts, counts_stride, (long long)nq,
                       out_ids, out_scores, out_counts);
    return hipGetLastError();
}

}  // namespace vdb
