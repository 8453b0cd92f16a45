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
#include <cmath>
#include <cstring>

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

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

// 16-byte row load; NT = non-temporal cache policy (rows are read once per scan)
template <bool NT>
__device__ __forceinline__ u32x4 load16(const char* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const u32x4*)p);
    else return *(const u32x4*)p;
}

template <int DT>
__device__ __forceinline__ float dot_unit(const u32x4& x, const float* q, float acc) {
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

// bf16 rows against a query held as packed bf16 hi + lo parts (q ~ qh + ql, |q - qh - ql| <= 2^-18 |q|):
// v_dot2c_f32_bf16 multiplies two element pairs per instruction, so 8 elements cost 8 instructions
// instead of 8 unpacks and 8 FMAs.  The products are exact in fp32; the accumulation error and the
// 2^-18 of the query split are part of the bound the finalize step certifies with.
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t x, uint32_t q, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2s, x), __builtin_bit_cast(bf16x2s, q), acc, false);
}
__device__ __forceinline__ float dot_unit_pk(const u32x4& x, const uint32_t* qh, const uint32_t* ql, float acc) {
    acc = dot2_bf16(x.x, qh[0], acc); acc = dot2_bf16(x.y, qh[1], acc);
    acc = dot2_bf16(x.z, qh[2], acc); acc = dot2_bf16(x.w, qh[3], acc);
    acc = dot2_bf16(x.x, ql[0], acc); acc = dot2_bf16(x.y, ql[1], acc);
    acc = dot2_bf16(x.z, ql[2], acc); acc = dot2_bf16(x.w, ql[3], acc);
    return acc;
}
__device__ __forceinline__ uint32_t bf16_bits(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f); }

// fp32 all-reduce over groups of LPR consecutive lanes without the LDS crossbar: DPP quad
// permutes and row mirrors inside 16-lane rows, permlane swaps across rows.  Every lane of a
// group ends with the group's sum (the pairing differs from an xor butterfly, which is
// irrelevant for the fast score).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPR >= 2)  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    if constexpr (LPR >= 4)  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    if constexpr (LPR >= 8)  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    if constexpr (LPR >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    if constexpr (LPR >= 32) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    if constexpr (LPR >= 64) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return v;
}

__device__ __forceinline__ double wave_tree_sum(double acc) {
    // fixed binary tree of the canonical score: lane 0 ends with the tree value
    for (int s = 32; s >= 1; s >>= 1) acc = __dadd_rn(acc, __shfl_xor(acc, s));
    return acc;
}

// Canonical fp64 score of one stored row against a prepared query, computed by one wave.
// Every lane returns the same value.  Lane l owns elements l, l+64, ... in increasing order;
// sixteen of them are fetched per round trip with unconditional loads (indices past d are
// clamped and the value replaced by zero, which leaves the sum unchanged bit for bit: the
// accumulator is never -0.0).
template <int DT>
__device__ __forceinline__ double canon_score_dt(const char* row, const float* q, int d, bool l2, int lane) {
#pragma clang fp contract(off)
    double acc = 0.0;
    for (int j0 = lane; j0 < d; j0 += 64 * 16) {
        float xv[16], yv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = j0 + 64 * u;
            const int jc = j < d ? j : d - 1;
            float x;
            if constexpr (DT == VDB_F32) x = ((const float*)row)[jc];
            else if constexpr (DT == VDB_BF16) x = __uint_as_float((uint32_t)((const unsigned short*)row)[jc] << 16);
            else x = f16_bits(((const unsigned short*)row)[jc]);
            const float y = q[jc];
            xv[u] = j < d ? x : 0.f;
            yv[u] = j < d ? y : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double x = (double)xv[u], y = (double)yv[u];
            if (l2) {
                const double df = __dsub_rn(x, y);
                acc = __dadd_rn(acc, __dmul_rn(df, df));
            } else {
                acc = __dadd_rn(acc, __dmul_rn(x, y));   // product exact: same as an fma
            }
        }
    }
    return wave_tree_sum(acc);
}

// The same for R rows at once (the exact pass): every row's sum is formed exactly as above, but the loads of
// all R rows of a round are in flight together, which is what keeps that pass near the memory's rate.
template <int DT, int R>
__device__ __forceinline__ void canon_score_rows(const char* const* row, const float* q, int d, bool l2, int lane, double* out) {
#pragma clang fp contract(off)
    constexpr int UL = 8;                              // elements per lane, row and round
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    for (int j0 = lane; j0 < d; j0 += 64 * UL) {
        float xv[R][UL], yv[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int j = j0 + 64 * u;
            const int jc = j < d ? j : d - 1;
            const float y = q[jc];                         // unconditional, like the row loads: a guarded load becomes a branch and a wait
            yv[u] = j < d ? y : 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float x;
                if constexpr (DT == VDB_F32) x = ((const float*)row[r])[jc];
                else if constexpr (DT == VDB_BF16) x = __uint_as_float((uint32_t)((const unsigned short*)row[r])[jc] << 16);
                else x = f16_bits(((const unsigned short*)row[r])[jc]);
                xv[r][u] = j < d ? x : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const double y = (double)yv[u];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double x = (double)xv[r][u];
                if (l2) {
                    const double df = __dsub_rn(x, y);
                    acc[r] = __dadd_rn(acc[r], __dmul_rn(df, df));
                } else {
                    acc[r] = __dadd_rn(acc[r], __dmul_rn(x, y));
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = wave_tree_sum(acc[r]);
}

template <int R>
__device__ __forceinline__ void canon_score_rows_dt(const char* const* row, const float* q, int d, int dt, bool l2, int lane, double* out) {
    if (dt == VDB_F32) canon_score_rows<VDB_F32, R>(row, q, d, l2, lane, out);
    else if (dt == VDB_BF16) canon_score_rows<VDB_BF16, R>(row, q, d, l2, lane, out);
    else canon_score_rows<VDB_F16, R>(row, q, d, l2, lane, out);
}

__device__ __forceinline__ double canon_score(const char* row, const float* q, int d, int dt,
                                              bool l2, int lane) {
    if (dt == VDB_F32) return canon_score_dt<VDB_F32>(row, q, d, l2, lane);
    if (dt == VDB_BF16) return canon_score_dt<VDB_BF16>(row, q, d, l2, lane);
    return canon_score_dt<VDB_F16>(row, q, d, l2, lane);
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
                  float* __restrict__ qprep, QInfo* __restrict__ qinfo, int* __restrict__ need_exact) {
    const int lane = threadIdx.x;
    // empty lists of flagged queries: the one the finalize step fills, and behind it the one of a second tier
    if (blockIdx.x == 0 && lane == 0) { need_exact[0] = 0; need_exact[1 + gridDim.x] = 0; }
    const float* q = q_raw + (size_t)blockIdx.x * d;
    float* out = qprep + (size_t)blockIdx.x * dpad;
    // Sixteen elements per lane are fetched per round trip and, for queries of up to 1024 elements, read only once:
    // the raw query may lie in host memory, where every dependent load is a trip over the bus.  The sums are formed
    // in the order of the plain loop (lane l: elements l, l + 64, ... in increasing order).
    const bool once = d <= 1024;
    float xv[16];
    double acc = 0.0;
    for (int j0 = lane; j0 < d; j0 += 1024) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int j = j0 + 64 * u; xv[u] = q[j < d ? j : d - 1]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (j0 + 64 * u < d) { const double x = (double)xv[u]; acc = __dadd_rn(acc, __dmul_rn(x, x)); }
        }
    }
    double n2 = wave_tree_sum(acc);
    n2 = __shfl(n2, 0);
    const bool cosine = metric == VDB_METRIC_COSINE;
    const double inv = (n2 == 0.0) ? 0.0 : 1.0 / sqrt(n2);
    double acc2 = 0.0;
    for (int j0 = lane; j0 < dpad; j0 += 1024) {
        if (!once) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int j = j0 + 64 * u; xv[u] = q[j < d ? j : d - 1]; }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = j0 + 64 * u;
            if (j < dpad) {
                float y = 0.f;
                if (j < d) y = cosine ? (float)__dmul_rn((double)xv[u], inv) : xv[u];
                out[j] = y;
                acc2 = __dadd_rn(acc2, __dmul_rn((double)y, (double)y));
            }
        }
    }
    if (cosine) n2 = __shfl(wave_tree_sum(acc2), 0);
    if (lane == 0) {
        QInfo qi;
        qi.n2 = n2;
        qi.bq = (float)(sqrt(n2) * (1.0 + 1e-6)) * 1.000001f;
        qi.pad = 0.f;
        qinfo[blockIdx.x] = qi;
    }
}

hipError_t launch_prep_query(const float* q_raw, int64_t nq, int d, int dpad, int metric,
                             float* qprep, QInfo* qinfo, int* need_exact, hipStream_t st) {
    hipLaunchKernelGGL(prep_query_kernel, dim3((unsigned)nq), dim3(64), 0, st,
                       q_raw, d, dpad, metric, qprep, qinfo, need_exact);
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
    int          contiguous;
    int          dynamic;   // the waves of a workgroup draw their steps from a counter in LDS
    float*       cand_s;    // [nq][G*W][K]: one sorted list per wave
    long long*   cand_t;
    float*       cand_h;    // [nq][G*W]: best key of each list, contiguous for the finalize kernel
    const float* floor;     // [nq] lower bound of the K-th best key per query, or null
    RowBound     rb;        // L2: error term added to every key (see RowBound)
    const int*   only;      // null, or a list {count, query indices...}: row y of the grid scans for query only[1 + y]
};

// Launch tunables of the scan (set through vdb_set_option, defaults chosen by measurement).
struct ScanTune {
    int nt = 1;              // non-temporal row loads (+13 % over the default cache policy)
    int waves = 8;           // waves per block (upper bound; the candidate-list geometry may lower it)
    int blocks_per_cu = 4;   // grid cap in blocks per CU (two are resident, two queue behind them)
    int contiguous = 1;      // 1 = each wave scans one contiguous range instead of interleaving
    int dynamic = 1;         // 1 = the waves of a workgroup share its steps out among themselves as they go
    double short_bytes = 2.8e10;   // tables below this size: one resident round of blocks, waves side by side
};
static ScanTune g_tune;
void set_scan_tune(int nt, int waves, int blocks_per_cu, int contiguous, int dynamic) {
    g_tune.nt = nt; g_tune.waves = waves; g_tune.blocks_per_cu = blocks_per_cu; g_tune.contiguous = contiguous;
    g_tune.dynamic = dynamic;
}
void set_scan_short_bytes(double bytes) { g_tune.short_bytes = bytes; }

#ifdef VDB_SCAN_STAMPS
// diagnostic build only (tools/scan_waves.py): per wave of the last scan, the time from its start to the end of
// its rows (10 ns ticks) and where it ran
__device__ unsigned int g_scan_wave_end[16384];
__device__ unsigned int g_scan_wave_xcc[16384];
#endif

template <int NCH> struct ScanUnroll { static constexpr int U = (NCH <= 1) ? 8 : (NCH <= 3) ? 4 : (NCH <= 6) ? 2 : 1; };

// LPR lanes share one row per wave-instruction (64/LPR rows per instruction); a row takes NCH
// such instructions (three of 16 or 32 lanes for rows of 48 or 96 units).  Every load is unconditional: lanes past the
// end of a row re-read its first 16 bytes against a zero query, rows past the end of the table
// re-read the last row and are discarded.
template <int DT, int LPR, int NCH, bool NT>
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
    int qy = blockIdx.y;
    if (p.only) { if (qy >= p.only[0]) return; qy = p.only[1 + qy]; }
    const float* q = p.q + (size_t)qy * p.dpad;

#ifdef VDB_SCAN_STAMPS
    const unsigned long long stamp0 = wall_clock64();
#endif
    constexpr bool PK = DT == VDB_BF16;            // packed bf16 arithmetic (see dot_unit_pk)
    int   off[NCH];
    float qv[NCH][PK ? 1 : EPL];
    uint32_t qh[NCH][4], ql[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int unit = c * LPR + lir;
        const bool act = unit < p.m;
        off[c] = act ? unit * 16 : 0;
        if constexpr (PK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h[2], l[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float x = act ? q[unit * 8 + 2 * e + b] : 0.f;
                    h[b] = bf16_bits(x);
                    l[b] = bf16_bits(x - __uint_as_float(h[b] << 16));
                }
                qh[c][e] = h[0] | (h[1] << 16);
                ql[c][e] = l[0] | (l[1] << 16);
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) qv[c][e] = act ? q[unit * EPL + e] : 0.f;
        }
    }

    WaveSel sel;
    sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane,
             p.floor ? p.floor[qy] : neg_inf());
    const bool rbon = p.l2 && p.rb.qinfo != nullptr;
    const float ea = rbon ? p.rb.a : 0.f;
    const float es = rbon ? rowbound_s(p.rb, lane) : 0.f;

    const long long TW = (long long)gridDim.x * W;
    const long long gw = (long long)blockIdx.x * W + wave;
    constexpr long long RPS = (long long)RPI * U;            // rows per wave step
    const long long nsteps = (p.n + TW * RPS - 1) / (TW * RPS);
    const size_t rowbytes = (size_t)p.m * 16;
    const long long last = p.n - 1;

    // The waves of a workgroup get through equal shares at different speeds (they end up to 50 us apart on a
    // scan of half a millisecond), so the workgroup's steps are handed out as the waves come for them: a
    // counter in LDS, one ds_add per step of U x NCH loads.  Which wave scores a row does not matter to the
    // result (every row is scored once, by the same arithmetic); the workgroup's rows are those of the static split.
    unsigned* const ticket = (unsigned*)(smem + (size_t)W * p.C * (sizeof(float) + sizeof(long long)));
    if (p.dynamic) {
        if (threadIdx.x == 0) *ticket = 0u;
        __syncthreads();
    }
    const long long wsteps = nsteps * W;                     // steps of the whole workgroup
    auto draw = [&]() __attribute__((always_inline)) -> long long {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(ticket, 1u);
        return (long long)__builtin_amdgcn_readfirstlane(t);
    };
    long long tk = p.dynamic ? draw() : wave;                // step of the workgroup: tk = st * W + (wave slot)
    while (tk < wsteps) {
        const long long st = tk / W, sub = tk - st * W;
        const long long base = (p.contiguous ? (p.dynamic ? (long long)blockIdx.x * wsteps + tk : gw * nsteps + st)
                                             : st * TW + (long long)blockIdx.x * W + sub) * RPS;
        const long long tk_next = p.dynamic ? draw() : tk + W;
        u32x4 x[U][NCH];
        float nn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = base + u * RPI + grp;
            const long long rc = r < last ? r : last;
            const char* ptr = p.rows + (size_t)rc * rowbytes;
#pragma unroll
            for (int c = 0; c < NCH; ++c) x[u][c] = load16<NT>(ptr + off[c]);
            nn[u] = p.l2 ? p.n2[rc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (PK) acc = dot_unit_pk(x[u][c], qh[c], ql[c], acc);
                else acc = dot_unit<DT>(x[u][c], qv[c], acc);
            }
            acc = group_sum<LPR>(acc);
            const long long r = base + u * RPI + grp;
            const float key = p.l2 ? (2.f * acc - (rbon ? rowbound_adj(nn[u], ea, es) : nn[u])) : acc;
            const bool pass = (r <= last) && (lir == 0) && better(key, r, sel.thr_s, sel.thr_t);
            sel.push(pass, key, r, lane);
        }
        tk = tk_next;
    }

#ifdef VDB_SCAN_STAMPS
    {
        const unsigned long long stamp1 = wall_clock64();
        const unsigned w = (unsigned)(blockIdx.x * W + wave);
        if (lane == 0 && w < 16384 && qy == 0) {
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            g_scan_wave_end[w] = (unsigned)(stamp1 - stamp0);
            g_scan_wave_xcc[w] = (xcc & 0xf) | (hwid << 4);
        }
    }
#endif
    // every wave hands over its own sorted list: no merge inside the block
    sel.compact(lane);
    {
        const size_t o = (((size_t)qy * gridDim.x + blockIdx.x) * W + wave) * p.K;
        for (int i = lane; i < p.K; i += 64) { p.cand_s[o + i] = sel.s[i]; p.cand_t[o + i] = sel.t[i]; }
        if (lane == 0) p.cand_h[o / p.K] = sel.s[0];
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
    int qy = blockIdx.y;
    if (p.only) { if (qy >= p.only[0]) return; qy = p.only[1 + qy]; }
    const float* q = p.q + (size_t)qy * p.dpad;
    const int nch = (p.m + 63) >> 6;
    // query staged as nch chunks of 64 units, zero beyond the row
    for (int i = threadIdx.x; i < nch * 64 * EPL; i += blockDim.x) qs[i] = (i < p.dpad) ? q[i] : 0.f;
    __syncthreads();

    WaveSel sel;
    sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane,
             p.floor ? p.floor[qy] : neg_inf());
    const bool rbon = p.l2 && p.rb.qinfo != nullptr;
    const float ea = rbon ? p.rb.a : 0.f;
    const float es = rbon ? rowbound_s(p.rb, lane) : 0.f;

    const long long TW = (long long)gridDim.x * W;
    const long long gw = (long long)blockIdx.x * W + wave;
    const long long nsteps = (p.n + TW * U - 1) / (TW * U);
    const size_t rowbytes = (size_t)p.m * 16;
    const long long last = p.n - 1;

    // steps of the workgroup drawn from a counter in LDS, as in scan_kernel (the counter sits behind the query)
    unsigned* const ticket = (unsigned*)(qs + (size_t)nch * 64 * EPL);
    if (p.dynamic) {
        if (threadIdx.x == 0) *ticket = 0u;
        __syncthreads();
    }
    const long long wsteps = nsteps * W;
    auto draw = [&]() __attribute__((always_inline)) -> long long {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(ticket, 1u);
        return (long long)__builtin_amdgcn_readfirstlane(t);
    };
    long long tk = p.dynamic ? draw() : wave;
    while (tk < wsteps) {
        const long long st = tk / W, sub = tk - st * W;
        const long long base = (p.contiguous ? (p.dynamic ? (long long)blockIdx.x * wsteps + tk : gw * nsteps + st)
                                             : st * TW + (long long)blockIdx.x * W + sub) * U;
        const long long tk_next = p.dynamic ? draw() : tk + W;
        float acc[U], nn[U];
        const char* ptr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = base + u;
            const long long rc = r < last ? r : last;
            ptr[u] = p.rows + (size_t)rc * rowbytes;
            acc[u] = 0.f;
            nn[u] = p.l2 ? p.n2[rc] : 0.f;
        }
        // three chunks of every row at a time: U x 3 loads in flight per wave (two for 16-bit rows, whose
        // eight query values per chunk and conversions would not fit the registers otherwise)
        constexpr int CG = DT == VDB_F32 ? 3 : 2;
        for (int c = 0; c < nch; c += CG) {
            float qv[CG][EPL];
            u32x4 x[CG][U];
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const int unit = (c + j) * 64 + lane;
                const bool live = (c + j) < nch;                    // chunks past the row: first unit again, zero query
                const int o = (live && unit < p.m) ? unit * 16 : 0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) qv[j][e] = live ? qs[unit * EPL + e] : 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) x[j][u] = load16<true>(ptr[u] + o);
            }
#pragma unroll
            for (int j = 0; j < CG; ++j)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = dot_unit<DT>(x[j][u], qv[j], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = acc[u];
            a = group_sum<64>(a);
            const long long r = base + u;
            float key = a;
            if (p.l2) key = 2.f * a - (rbon ? rowbound_adj(nn[u], ea, es) : nn[u]);
            const bool pass = (r <= last) && (lane == 0) && better(key, r, sel.thr_s, sel.thr_t);
            sel.push(pass, key, r, lane);
        }
        tk = tk_next;
    }

    // every wave hands over its own sorted list: no merge inside the block
    sel.compact(lane);
    {
        const size_t o = (((size_t)qy * gridDim.x + blockIdx.x) * W + wave) * p.K;
        for (int i = lane; i < p.K; i += 64) { p.cand_s[o + i] = sel.s[i]; p.cand_t[o + i] = sel.t[i]; }
        if (lane == 0) p.cand_h[o / p.K] = sel.s[0];
    }
}

// Shape of the specialised kernel chosen for a row of m 16-byte units.
struct ScanShape { int lpr, nch; bool generic; };
static ScanShape scan_shape(int m, int dt) {
    ScanShape s{64, 1, false};
    // rows of three quarters of a power of two (768 16-bit or 384 fp32 elements: 96 units) take three
    // instructions of half that many lanes: no lane re-reads the row's head, as a quarter of them would
    // with the next power of two
    if (m > 64 && m <= 96) { s.lpr = 32; s.nch = 3; return s; }
    if (m > 32 && m <= 48) { s.lpr = 16; s.nch = 3; return s; }
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
    long long cap = (long long)n_cu * (g.W >= 16 ? 1 : (16 / g.W));
    if (g_tune.blocks_per_cu > 0) {
        // Tables of up to about 28 GB do best with only the blocks that are resident at once (two of
        // 8 waves per CU): a second round of blocks costs its start-up and the first round's tail
        // again (1 M x 768 fp32: 78.8 -> 83.2 % of the HBM peak; with the step counter 5 M x 768:
        // 86.9 -> 88.1 %, 8.5 M: 87.2 -> 87.8 %, 10 M: even, 12 M: 87.2 against 87.6 %); longer scans
        // gain more from the finer balance of four blocks per CU.
        const double bytes = (double)t.n * t.m * 16.0;
        cap = (long long)n_cu * (bytes < g_tune.short_bytes && g_tune.blocks_per_cu > 2 ? 2 : g_tune.blocks_per_cu);
        // blocks of four to seven waves (candidate lists of 512 to 1024 entries, 48 KiB of lists per block):
        // three such blocks are resident per CU, and that one round is launched
        if (g.W >= 2 && g.W < 8) cap = (long long)n_cu * 3;
    }
    return (int)(need < cap ? need : cap);
}

template <int DT>
static hipError_t launch_scan_dt(const ScanShape& s, dim3 grid, dim3 block, size_t lds,
                                 const ScanParams& p, hipStream_t st) {
#define VDB_SCAN_CASE(L, N)                                                                       \
    if (s.lpr == L && s.nch == N) {                                                               \
        if (g_tune.nt) hipLaunchKernelGGL((scan_kernel<DT, L, N, true>), grid, block, lds, st, p);  \
        else hipLaunchKernelGGL((scan_kernel<DT, L, N, false>), grid, block, lds, st, p);          \
        return hipGetLastError();                                                                 \
    }
    if (!s.generic) {
        VDB_SCAN_CASE(1, 1) VDB_SCAN_CASE(2, 1) VDB_SCAN_CASE(4, 1) VDB_SCAN_CASE(8, 1)
        VDB_SCAN_CASE(16, 1) VDB_SCAN_CASE(32, 1) VDB_SCAN_CASE(64, 1)
        VDB_SCAN_CASE(64, 2) VDB_SCAN_CASE(64, 3) VDB_SCAN_CASE(64, 4)
        VDB_SCAN_CASE(32, 3) VDB_SCAN_CASE(16, 3)
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
                       const SelGeom& g, int G, float* cand_s, long long* cand_t, float* cand_h,
                       const float* floor, const int* only, hipStream_t st) {
    ScanParams p;
    p.only = only;
    p.rows = t.rows; p.n2 = t.n2; p.q = qprep; p.n = t.n; p.m = t.m; p.dpad = dpad;
    p.K = g.K; p.C = g.C; p.l2 = (t.metric == VDB_METRIC_L2) ? 1 : 0;
    p.rb = t.rb;
    // each wave scans one contiguous range of a long table; over short ones the waves advance through the
    // table side by side, so that they all end together (1 M x 768 fp32: 83.3 -> 84.0 % of the HBM peak)
    p.contiguous = g_tune.contiguous && (double)t.n * t.m * 16.0 >= g_tune.short_bytes;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cand_h = cand_h; p.floor = floor;
    const ScanShape s = scan_shape(t.m, t.dt);
    const dim3 grid((unsigned)G, (unsigned)nq), block((unsigned)g.W * 64);
    const size_t lds = (size_t)g.W * g.C * (sizeof(float) + sizeof(long long)) + 16;      // lists + the step counter
    p.dynamic = g_tune.dynamic;
    switch (t.dt) {
        case VDB_F32:  return launch_scan_dt<VDB_F32>(s, grid, block, lds, p, st);
        case VDB_BF16: return launch_scan_dt<VDB_BF16>(s, grid, block, lds, p, st);
        default:       return launch_scan_dt<VDB_F16>(s, grid, block, lds, p, st);
    }
}

// ------------------------------------------------------------------------------ read ceiling
// Streams a buffer with the scan's access shape (16 bytes per lane, 12 loads in flight per wave)
// and almost no arithmetic: the read bandwidth this chip grants such a kernel, measured beside
// the scan by the tuning script.
template <bool NT>
__global__ void __launch_bounds__(1024)
read_bw_kernel(const char* __restrict__ buf, long long units, unsigned int* __restrict__ sink) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    unsigned int acc = 0;
    long long i = tid;
    for (; i + 11 * nthreads < units; i += 12 * nthreads) {
        u32x4 x[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) x[j] = load16<NT>(buf + (size_t)(i + j * nthreads) * 16);
#pragma unroll
        for (int j = 0; j < 12; ++j) acc ^= x[j].x ^ x[j].y ^ x[j].z ^ x[j].w;
    }
    for (; i < units; i += nthreads) { const u32x4 x = load16<NT>(buf + (size_t)i * 16); acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x9e3779b9u) sink[0] = acc;     // keeps the loads alive, practically never taken
}

hipError_t launch_read_bw(const void* buf, long long bytes, int nt, int blocks, int threads,
                          unsigned int* sink, hipStream_t st) {
    const long long units = bytes / 16;
    if (nt) hipLaunchKernelGGL((read_bw_kernel<true>), dim3((unsigned)blocks), dim3((unsigned)threads), 0, st, (const char*)buf, units, sink);
    else hipLaunchKernelGGL((read_bw_kernel<false>), dim3((unsigned)blocks), dim3((unsigned)threads), 0, st, (const char*)buf, units, sink);
    return hipGetLastError();
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
    const float*     cand_h;     // lists mode: [nq][G] list heads
    long long*       out_ids;
    float*           out_scores;
    int*             out_counts;
    int*             need_exact;
    const int*       only;       // lists mode: null, or {count, query indices...}: block b works on query only[1 + b]
    long long*       counters;
    // flat mode: one unsorted candidate array per query (batched path)
    const int*       flat_cnt;   // [nq] candidates appended (may exceed flat_cap: the excess was lost)
    int              flat_cap;
    const float*     flat_tau;   // [nq] every row that is not a candidate has a fast key <= tau
    const int*       flat_lost;  // [nq]
    const int*       flat_bad;   // [nq] query not representable on the batched path (may be null)
    double           dd_mult, extra_rel, extra_abs, l2_bias;   // see FlatBound
    int              force_exact;
    float*           floor_out;  // lists mode, non-null: only write the K-th best fast key per query and return
    int              fin_cap;    // lists mode: survivors the kernel's LDS holds (FIN_CAP or FIN_CAP_LONG)
    int              pth;        // lists mode: which entry of every list the first bound is taken from (fin_pth; 0: the heads, per thread)
    int*             host_flag = nullptr;   // lists mode, non-null: [query] set once the answer is written: 1 final, 2 the query goes on to the exact pass (plain store; may be host memory)
};

__device__ __forceinline__ int pow2_ceil_dev(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Order-preserving map of an fp32 score onto uint32 (larger score -> larger key).
__device__ __forceinline__ uint32_t fkey(float s) {
    const uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return (k & 0x80000000u) ? __uint_as_float(k & 0x7fffffffu) : __uint_as_float(~k);
}

// One histogram increment per distinct bin per wave instead of one per lane: scores of the
// candidates share their leading bits, so a plain per-lane LDS atomic would serialise 64-fold.
// Bins that few lanes share are cheaper left to the LDS atomics themselves (one increment per lane, spread over the
// banks): after four rounds of aggregation the lanes still waiting add their own.
__device__ __forceinline__ void hist_add_aggregated(int* hist, bool active, int bin) {
    unsigned long long todo = __ballot(active);
    for (int round = 0; todo && round < 4; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader);
        const unsigned long long same = __ballot(active && bin == b) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b], __popcll(same));
        todo &= ~same;
    }
    if ((todo >> (threadIdx.x & 63)) & 1ull) atomicAdd(&hist[bin], 1);
}

// Block-wide K-th largest of `count` fp32 values by an 8-bit MSD radix select (four histogram
// passes, no sorting).  `get(i, u)` returns value i (-inf = empty, never counted); u is the slot
// of the 8-wide batch, a compile-time constant after unrolling, so that a source of at most
// 8 * blockDim.x values may sit in registers (slot u <-> value u * blockDim.x + tid).  On return:
// ctl[4] = number of real values; if that is <= K the caller takes them all (returns true);
// otherwise `thr` is the key of the K-th largest and `need` how many values equal to it belong
// to the K (all values with a larger key do).  With npass < 4 only the leading 8*npass bits are
// resolved and `thr` is a lower bound of the K-th largest key (`need` is then meaningless).
// Called by every thread of the block.
template <typename F>
__device__ __forceinline__ bool select_kth(F get, long long count, int K, int* hist, int* ctl,
                                           uint32_t& thr, int& need, int npass = 4) {
    const int nth = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t prefix = 0;
    need = K;
    for (int pass = 0; pass < npass; ++pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = 24 - 8 * pass;
        for (long long i0 = 0; i0 < count; i0 += (long long)nth * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const long long i = i0 + (long long)u * nth + tid; v[u] = (i < count) ? get(i, u) : neg_inf(); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + (long long)u * nth < count) {             // block-uniform
                    const uint32_t key = fkey(v[u]);
                    const bool act = (v[u] > neg_inf()) && (pass == 0 || (key >> (shift + 8)) == prefix);
                    hist_add_aggregated(hist, act, (int)((key >> shift) & 255u));
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            int h[4], tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[j] = hist[255 - 4 * lane - j]; tot += h[j]; }
            int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            const int excl = incl - tot;
            if (pass == 0 && lane == 63) ctl[4] = incl;
            if (excl < need && need <= incl) {
                int acc = excl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (acc < need && need <= acc + h[j]) { ctl[0] = (int)((prefix << 8) | (uint32_t)(255 - 4 * lane - j)); ctl[1] = need - acc; }
                    acc += h[j];
                }
            }
        }
        __syncthreads();
        if (pass == 0 && ctl[4] <= K) return true;
        prefix = (uint32_t)ctl[0];
        need = ctl[1];
    }
    // with fewer than four passes the undecided low bits are zero: the lower edge of the bin
    // that holds the K-th largest, i.e. a lower bound of it
    thr = npass < 4 ? prefix << (32 - 8 * npass) : prefix;
    return false;
}

constexpr int FIN_CAP = 2048;     // survivors of the pre-filter kept in LDS (FIN_CAP_LONG for lists of more than 512 entries)
constexpr int FIN_CAP_LONG = 4096;
constexpr int FIN_RANK_MAX = 256;  // up to this many values are ranked directly (quadratic) instead of radix-selected

// Position of value i among V[0..n) in descending order (ties by index).  Broadcast LDS reads.
__device__ __forceinline__ int rank_of(const float* V, int n, int i, float vi) {
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const float vj = V[j];
        rank += (vj > vi || (vj == vi && j < i)) ? 1 : 0;
    }
    return rank;
}

// One block per query.  The G per-block candidate lists arrive sorted, so the K-th largest list
// head is a lower bound of the K-th best key overall; everything below it is dropped with one
// compare per candidate, and the exact K best are radix-selected among the few survivors.  The
// K rows are then re-scored canonically, sorted by (score, id) and the answer is certified.
__global__ void __launch_bounds__(1024)
finalize_kernel(FinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nth = blockDim.x;
    const int W = nth >> 6;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int K = p.K;
    const int KR = pow2_ceil_dev(K);
    int*       hist = (int*)smem;                       // 256 bins
    int*       ctl  = hist + 256;                       // 16 control words
    long long* ST   = (long long*)(smem + 1088);        // K selected tags (row indices)
    float*     SS   = (float*)(ST + K);                 // K selected fast keys
    long long* RT   = (long long*)(SS + K);             // KR re-scored ids
    float*     RS   = (float*)(RT + KR);                // KR re-scored keys
    float*     VS   = RS + KR;                          // fin_cap survivor keys
    int*       VI   = (int*)(VS + p.fin_cap);           // fin_cap survivor positions

    int qi = blockIdx.x;
    if (p.only) { if (qi >= p.only[0]) return; qi = p.only[1 + qi]; }     // block-uniform
    const long long M = (long long)p.G * K;
    const size_t qoff = (size_t)qi * M;
    const float*     gs = p.cand_s + qoff;
    const long long* gt = p.cand_t + qoff;

    if (tid < 16) ctl[tid] = 0;
    __syncthreads();
    uint32_t thr = 0;
    int need = K;
    bool take_all;
    // what the certificate at the very end needs from memory is asked for now
    const float pre_max_n2 = p.floor_out ? 0.f : *p.t.max_n2;
    QInfo pre_qinfo;
    pre_qinfo.n2 = 0.0; pre_qinfo.bq = 0.f; pre_qinfo.pad = 0.f;
    if (!p.floor_out) pre_qinfo = p.qinfo[qi];
    // diagnostic build only (-DVDB_FIN_STAMPS): stage boundaries of query 0 in 10 ns ticks
#ifdef VDB_FIN_STAMPS
#define STAMP(n) do { if (tid == 0 && qi == 0 && p.counters != nullptr) { p.counters[8 + (n)] = (long long)wall_clock64(); p.counters[16 + (n)] = (long long)clock64(); } } while (0)
#else
#define STAMP(n) do { } while (0)
#endif
    STAMP(0);

    // (A) lower bound from the list heads; (B) walk only the lists whose head survives it
    uint32_t lb = 0;
    const bool heads_in_regs = p.G <= 8 * nth;
    if (heads_in_regs) {
        const float* gh = p.cand_h + (size_t)qi * p.G;
        // thread t owns the lpt consecutive lists t*lpt ... (the waves of one scan block)
        const int lpt = (p.G + nth - 1) / nth;               // <= 8
        float head[8];
        float tmax = neg_inf();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long b = (long long)tid * lpt + u;
            head[u] = (u < lpt && b < p.G) ? gh[b] : neg_inf();
            tmax = fmaxf(tmax, head[u]);
        }
        // Every thread's best head is a distinct candidate: the K-th largest of those (lower edge of its
        // 16-bit radix bin) is a lower bound of the K-th best key overall.  When K comes near the number
        // of threads that hold lists that bound lets everything through; then (p.pth >= 1, chosen by
        // fin_pth) every list takes part with its pth entry (the lists are sorted, best first):
        // ceil(K / pth) lists whose pth entry reaches a value hold pth entries each that reach it, K or
        // more in all.
        const int T = (p.G + lpt - 1) / lpt;
        if (p.pth >= 1) {
            float pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long b = (long long)tid * lpt + u;
                pv[u] = neg_inf();
                if (u < lpt && b < p.G) pv[u] = p.pth == 1 ? head[u] : gs[b * K + (p.pth - 1)];
            }
            const int Kp = (K + p.pth - 1) / p.pth;
            int nd;
            // value (tid, u) is enumerated as index u * nth + tid: u < lpt for every index below nth * lpt
            const bool all = select_kth([&](long long, int u) { return pv[u]; }, (long long)nth * lpt, Kp, hist, ctl, lb, nd, 2);
            if (all) lb = 0;
            __syncthreads();
        } else if (T > K) {
            int nd;
            const bool all = select_kth([&](long long, int) { return tmax; }, (long long)T, K, hist, ctl, lb, nd, 2);
            if (all) lb = 0;
            __syncthreads();
        }
        STAMP(1);
        // the lists are sorted: a list contributes its leading entries down to the bound
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long b = (long long)tid * lpt + u;
            if (u < lpt && b < p.G && head[u] > neg_inf() && fkey(head[u]) >= lb) {
                bool more = true;
                for (int i = 0; more && i < K; i += 4) {
                    const f32x4s v = *(const f32x4s*)(gs + b * K + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e];
                        if (more && x > neg_inf() && fkey(x) >= lb) {
                            const int pos = atomicAdd(&ctl[5], 1);
                            if (pos < p.fin_cap) { VS[pos] = x; VI[pos] = (int)(b * K + i + e); }
                        } else {
                            more = false;
                        }
                    }
                }
            }
        }
    } else {
        if (p.G > K) {
            int nd;
            const bool all = select_kth([&](long long b, int) { return gs[b * K]; }, (long long)p.G, K, hist, ctl, lb, nd, 2);
            if (all) lb = 0;
            __syncthreads();
        }
        STAMP(1);
        for (long long i0 = 0; i0 < M; i0 += (long long)nth * 32) {
            float v[32];                             // 32 loads in flight: one round trip per 32*nth candidates
#pragma unroll
            for (int u = 0; u < 32; ++u) { const long long i = i0 + (long long)u * nth + tid; v[u] = (i < M) ? gs[i] : neg_inf(); }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (i0 + (long long)u * nth >= M) continue;              // block-uniform
                const bool pass = (v[u] > neg_inf()) && fkey(v[u]) >= lb;
                const unsigned long long m = __ballot(pass);
                if (m) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ctl[5], __popcll(m));
                    base = __shfl(base, 0);
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (pass && pos < p.fin_cap) { VS[pos] = v[u]; VI[pos] = (int)(i0 + (long long)u * nth + tid); }
                }
            }
        }
    }
    __syncthreads();
    STAMP(2);
    const int S = ctl[5];
    if (S <= FIN_RANK_MAX) {
        // (C) few survivors: rank them directly; the K best land in SS/ST already ordered
        take_all = S <= K;
        for (int j = tid; j < S; j += nth) {
            const float sc = VS[j];
            const int r = rank_of(VS, S, j, sc);
            if (r < K) { SS[r] = sc; ST[r] = gt[VI[j]]; }
            if (r == K - 1) ctl[7] = (int)fkey(sc);
        }
        __syncthreads();
        thr = (uint32_t)ctl[7];
        if (tid == 0) ctl[4] = S;
    } else if (S <= p.fin_cap) {
        // (C) exact K best among the survivors
        take_all = select_kth([&](long long j, int) { return VS[j]; }, (long long)S, K, hist, ctl, thr, need);
        for (int j = tid; j < S; j += nth) {
            const float sc = VS[j];
            const uint32_t key = fkey(sc);
            int pos = -1;
            if (take_all || key > thr) pos = atomicAdd(&ctl[2], 1);
            else if (key == thr) { const int e = atomicAdd(&ctl[3], 1); if (e < need) pos = (K - need) + e; }
            if (pos >= 0 && pos < K) { SS[pos] = sc; ST[pos] = gt[VI[j]]; }
        }
    } else {
        // more survivors than the buffer holds (massive ties): select over all candidates
        take_all = select_kth([&](long long i, int) { return gs[i]; }, M, K, hist, ctl, thr, need);
        for (long long i = tid; i < M; i += nth) {
            const float sc = gs[i];
            if (!(sc > neg_inf())) continue;
            const uint32_t key = fkey(sc);
            int pos = -1;
            if (take_all || key > thr) pos = atomicAdd(&ctl[2], 1);
            else if (key == thr) { const int e = atomicAdd(&ctl[3], 1); if (e < need) pos = (K - need) + e; }
            if (pos >= 0 && pos < K) { SS[pos] = sc; ST[pos] = gt[i]; }
        }
    }
    __syncthreads();
    STAMP(3);
    if (p.floor_out) {
        // threshold-only use: the K-th best key of these lists bounds the K-th best key of any
        // superset of their rows from below
        if (tid == 0) p.floor_out[qi] = take_all ? neg_inf() : fkey_inv(thr);
        return;
    }
    const int nvalid = take_all ? ctl[4] : K;
    const uint32_t prefix = thr;

    const bool l2 = p.t.metric == VDB_METRIC_L2;
    const float* q = p.q + (size_t)qi * p.dpad;
    const size_t rowbytes = (size_t)p.t.m * 16;
    // four rows per wave and round, their loads in flight together (slots past the selected rows read row 0 and
    // are dropped)
    for (int i0 = wave; i0 < KR; i0 += 4 * W) {
        const char* rp[4];
        long long id[4];
        double sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * W;
            const long long r = i < nvalid ? ST[i] : 0ll;
            rp[j] = p.t.rows + (size_t)r * rowbytes;
            id[j] = p.t.ids[r];
        }
#if defined(VDB_FIN_EXP) && VDB_FIN_EXP == 1
        for (int j = 0; j < 4; ++j) sc[j] = (double)j;
#else
        if (i0 < nvalid) canon_score_rows_dt<4>(rp, q, p.t.d, p.t.dt, l2, lane, sc);      // wave-uniform
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * W;
            float key = neg_inf();
            long long tag = TAG_NONE;
            if (i < nvalid) { const float f = (float)sc[j]; key = l2 ? -f : f; tag = id[j]; }
            if (lane == 0 && i < KR) { RS[i] = key; RT[i] = tag; }
        }
    }
    __syncthreads();
    STAMP(4);
    if (wave == 0) {
        // ids may legitimately equal TAG_NONE; empties are told apart by position (>= nvalid)
        // because a real entry is never worse than an empty one with the same key and id.
        if (KR <= 64) {
            // rank sort: every lane counts the entries that precede its own (broadcast LDS reads)
            const bool has = lane < KR;
            const float ms = has ? RS[lane] : neg_inf();
            const long long mt = has ? RT[lane] : TAG_NONE;
            int rank = 0;
            for (int j = 0; j < KR; ++j) {
                const float sj = RS[j];
                const long long tj = RT[j];
                rank += (better(sj, tj, ms, mt) || (sj == ms && tj == mt && j < lane)) ? 1 : 0;
            }
            lds_wave_sync();
            if (has) { RS[rank] = ms; RT[rank] = mt; }
            lds_wave_sync();
        } else {
            wave_bitonic_sort(RS, RT, KR, lane);
        }
        const int cnt = nvalid < p.k ? nvalid : p.k;
        for (int i = lane; i < p.k; i += 64) {
            const bool ok = i < cnt;
            p.out_ids[(size_t)qi * p.k + i] = ok ? RT[i] : -1ll;
            const float key = RS[i < KR ? i : 0];
            p.out_scores[(size_t)qi * p.k + i] = ok ? (l2 ? -key : key) : (l2 ? __builtin_huge_valf() : neg_inf());
        }
        if (lane == 0) {
            p.out_counts[qi] = cnt;
            // lists mode with fewer candidates than K: every stored row is among them
            bool cert = nvalid < K;
            if (!cert && cnt > 0) {
                // theta: no row outside the candidate set has a larger fast key
                double theta = (double)fkey_inv(prefix);
                if (take_all) { float mn = SS[0]; for (int i = 1; i < nvalid; ++i) mn = fminf(mn, SS[i]); theta = (double)mn; }
                const double bx = sqrt((double)pre_max_n2) * (1.0 + 1e-6);
                const double bq = (double)pre_qinfo.bq;
                const double u = 5.9604644775390625e-08;   // 2^-24
                const double dd = (double)p.t.d * p.dd_mult + 16.0;
                const double xq = p.extra_rel * bx * bq + p.extra_abs * bx;     // query rounding of the batched path
                if (!l2) {
                    const double eps = 1.05 * dd * u * bx * bq + xq + 4.0 * u * fabs(theta) + 1e-37;
                    cert = (double)RS[cnt - 1] > theta + eps;
                } else {
                    // keys that carry their own row's error term bound the canonical keys by themselves
                    const double eps = (p.t.rb.qinfo ? 4.0 * u * pre_qinfo.n2
                                                     : 1.05 * u * (2.0 * dd * bx * bq + 4.0 * (bx * bx + 2.0 * bx * bq)) + 2.0 * xq)
                                       + 4.0 * u * fabs(theta) + 1e-37;
                    const double dist_k = -(double)RS[cnt - 1];
                    const double bound = pre_qinfo.n2 - theta - eps;     // outside rows are farther than this
                    cert = dist_k * (1.0 + 8.0 * u) + 1e-37 < bound;
                }
            }
            if (p.force_exact) cert = false;
            if (!cert) { const int pos = atomicAdd(&p.need_exact[0], 1); p.need_exact[1 + pos] = qi; }
            if (p.host_flag) {
                // the answer above is out before the word that announces it
                __threadfence_system();
                *(volatile int*)&p.host_flag[qi] = cert ? 1 : 2;
            }
            STAMP(5);
            if (!p.only) atomicAdd((unsigned long long*)&p.counters[0], 1ull);      // a second tier answers queries already counted
        }
    }
}

constexpr int BAND_CAP = 2048;    // most rows one query re-scores on the batched path

// Batched path, one block per query.  The flat candidate array holds every row whose fast key
// exceeds tau, unsorted.  Let f_k be the k-th best fast key and eps a bound of
// |fast key - canonical key| (canonical keys shifted by the query's squared norm for L2).  The k rows
// with the best fast keys have canonical keys >= f_k - eps, so the k-th best canonical key is at
// least that, and a row whose canonical key reaches it has a fast key >= f_k - 2 eps.  Every
// candidate in that band is re-scored canonically and the band is ranked by (score, id): the first
// k are the answer and nothing is left to certify, however coarse the fast keys are.  A query goes to
// the exact pass only when the candidates do not cover the band (f_k - 2 eps <= tau, candidates lost
// to a full array, an unrepresentable query) or the band holds more than BAND_CAP rows.
__global__ void __launch_bounds__(1024)
finalize_band_kernel(FinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nth = blockDim.x;
    const int W = nth >> 6;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int*       hist = (int*)smem;                       // 256 bins
    int*       ctl  = hist + 256;                       // 16 control words
    long long* ST   = (long long*)(smem + 1088);        // BAND_CAP rows of the band
    long long* RT   = ST + BAND_CAP;                    // their ids
    float*     RS   = (float*)(RT + BAND_CAP);          // their canonical keys

    const int qi = blockIdx.x;
    const long long M = p.flat_cnt[qi] < p.flat_cap ? p.flat_cnt[qi] : p.flat_cap;
    const float*     gs = p.cand_s + (size_t)qi * p.flat_cap;
    const long long* gt = p.cand_t + (size_t)qi * p.flat_cap;
    const bool l2 = p.t.metric == VDB_METRIC_L2;
    if (tid < 16) ctl[tid] = 0;
    __syncthreads();

    // every decision below is block-uniform
    bool ok = p.flat_lost[qi] == 0 && (p.flat_bad == nullptr || p.flat_bad[qi] == 0) && M > p.k && !p.force_exact;
    uint32_t thr = 0;
    int need = 0;
    const bool rowb = l2 && p.t.rb.qinfo != nullptr;
    double band = 0.0;
    if (ok && rowb) {
        // Keys with their row's error term e(x) added: a key u bounds its canonical key c from above
        // (c <= u + D) and u - 2 e(x) - D bounds it from below, D = 4 u ||q||^2 for the shift by the
        // query's norm.  The k-th largest lower bound L is a lower bound of the k-th best canonical
        // key, and every row that can reach it has a key >= L - D.
        const double u = 5.9604644775390625e-08;        // 2^-24
        const double ea = (double)p.t.rb.a, es = (double)rowbound_s(p.t.rb, lane);
        const double bq = (double)p.qinfo[qi].bq;
        const double D = 4.0 * u * bq * bq;
        auto lower = [&](long long i, int) -> float {
            const float key = gs[i];
            if (!(key > neg_inf())) return neg_inf();
            const double n2 = (double)p.t.n2[gt[i]];
            const double e = (ea * n2 + es * sqrt(n2)) * (1.0 + 1e-6) + 4.0 * u * n2;      // what the pass added, at most
            const double lo = (double)key - 2.0 * e - D - 4.0 * u * fabs((double)key) - 1e-37;
            float f = (float)lo;
            if ((double)f > lo) f = nextafterf(f, neg_inf());                                   // round down
            return f;
        };
        const bool all = select_kth(lower, M, p.k, hist, ctl, thr, need);
        if (all) ok = false;
        if (ok) {
            const double L = (double)fkey_inv(thr);
            band = L - D - 4.0 * u * fabs(L) - 1e-37;
            if (!(band > (double)p.flat_tau[qi])) ok = false;
        }
    }
    // A few rows far longer than the rest (IP tables): the bound below is taken over the rest, the long
    // rows' fast keys are left out (their own error bars dwarf the scores), and all of them are re-scored
    // canonically with the band, whatever their fast keys say.
    const bool side = !l2 && p.t.n_out > 0;
    auto long_row = [&](long long i) -> bool { return side && p.t.n2[gt[i]] > p.t.out_limit; };
    if (ok && !rowb) {
        const bool all = select_kth([&](long long i, int) { return long_row(i) ? neg_inf() : gs[i]; }, M, p.k, hist, ctl, thr, need);
        if (all) ok = false;                            // fewer than k usable keys
    }
    if (ok && !rowb) {
        const double fk = (double)fkey_inv(thr);
        const double bx = sqrt((double)(side ? p.t.max_rest : *p.t.max_n2)) * (1.0 + 1e-6);
        const double bq = (double)p.qinfo[qi].bq;
        const double u = 5.9604644775390625e-08;        // 2^-24
        const double dd = (double)p.t.d * p.dd_mult + 16.0;
        const double xq = p.extra_rel * bx * bq + p.extra_abs * bx;
        double eps;
        if (!l2) eps = 1.05 * dd * u * bx * bq + xq;
        else eps = 1.05 * u * (2.0 * dd * bx * bq + 4.0 * (bx * bx + 2.0 * bx * bq)) + 2.0 * xq + 4.0 * u * (bx + bq) * (bx + bq)
                   + p.l2_bias * 1.05 * u * dd * bx * bx;       // keys accumulated on top of ||x||^2: partial sums of that size
        eps = (eps + 4.0 * u * fabs(fk) + 1e-37) * 1.001;
        eps += 8.0 * u * eps;                           // keys in the band are up to 2 eps away from f_k
        band = fk - 2.0 * eps;
        if (!(band > (double)p.flat_tau[qi])) ok = false;     // also catches a non-finite bound
    }
    if (ok) {
        for (long long i0 = 0; i0 < M; i0 += nth) {
            const long long i = i0 + tid;
            const bool in = i < M && (double)gs[i] >= band && !long_row(i);
            const unsigned long long m = __ballot(in);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&ctl[5], __popcll(m));
                base = __shfl(base, 0);
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (in && pos < BAND_CAP) ST[pos] = gt[i];
            }
        }
        if (side) {
            // the rows on the list, as far as this search sees them (rows of an insert still under way do not count)
            for (int j = tid; j < p.t.n_out; j += nth) {
                const long long r = p.t.outliers[j];
                if (r >= 0 && r < p.t.n) {
                    const int pos = atomicAdd(&ctl[5], 1);
                    if (pos < BAND_CAP) ST[pos] = r;
                }
            }
        }
        __syncthreads();
        if (ctl[5] > BAND_CAP) ok = false;
    }
    if (!ok) {
        if (tid == 0) {
            const int pos = atomicAdd(&p.need_exact[0], 1);
            p.need_exact[1 + pos] = qi;
            atomicAdd((unsigned long long*)&p.counters[0], 1ull);
        }
        return;
    }
    const int S = ctl[5];                               // >= k: the k best fast keys are in the band
    const float* q = p.q + (size_t)qi * p.dpad;
    const size_t rowbytes = (size_t)p.t.m * 16;
    for (int i = wave; i < S; i += W) {
        const long long r = ST[i];
        const double sc = canon_score(p.t.rows + (size_t)r * rowbytes, q, p.t.d, p.t.dt, l2, lane);
        const float f = (float)sc;
        if (lane == 0) { RS[i] = l2 ? -f : f; RT[i] = p.t.ids[r]; }
    }
    __syncthreads();
    for (int j = tid; j < S; j += nth) {
        const float ms = RS[j];
        const long long mt = RT[j];
        int rank = 0;
        for (int i = 0; i < S; ++i) {
            const float si = RS[i];
            const long long ti = RT[i];
            rank += (better(si, ti, ms, mt) || (si == ms && ti == mt && i < j)) ? 1 : 0;
        }
        if (rank < p.k) {
            p.out_ids[(size_t)qi * p.k + rank] = mt;
            p.out_scores[(size_t)qi * p.k + rank] = l2 ? -ms : ms;
        }
    }
    if (tid == 0) {
        p.out_counts[qi] = p.k;
        atomicAdd((unsigned long long*)&p.counters[0], 1ull);
    }
}

RowBound l2_row_bound(const FlatBound& fb, int d, const QInfo* qinfo, int nq) {
    // the terms of the table-wide bounds of the finalize kernels, sorted by their power of ||x||
    const double u = 5.9604644775390625e-08;            // 2^-24
    const double dd = (double)d * fb.dd_mult + 16.0;
    const double a = u * (8.2 + 1.05 * dd * fb.l2_bias);
    const double brel = u * (1.05 * (2.0 * dd + 8.0) + 8.0) + 2.0 * fb.extra_rel;
    const double cabs = 2.0 * fb.extra_abs;
    auto up = [](double v) { float f = (float)(v * 1.0001); if ((double)f < v) f = nextafterf(f, __builtin_huge_valf()); return f; };
    RowBound rb;
    rb.qinfo = qinfo; rb.nq = nq;
    rb.a = up(a + 4.0 * u);                              // + the roundings of ||x||^2 - e(x) in fp32
    rb.brel = up(brel); rb.cabs = up(cabs);
    return rb;
}

// Which entry of the lists the finalize kernel's first bound is taken from.  0: the K-th largest of the per-thread
// best heads, which is cheapest and tight while the threads holding lists outnumber K two to one.  Otherwise every
// list takes part with its p-th entry, p in {1, 2, 4, 8, 16} chosen for the fewest expected survivors: with rows
// dealt to the G lists at random, a list holds Poisson(lambda) rows above a value that N = lambda * G rows reach,
// and the bound is the value that ceil(K / p) of the G lists reach with p rows.
static int g_fin_threads = 512;        // measured on the blocking path over small tables: 1024 threads 41.7 us per call, 512 39.5, 256 41.3
void set_fin_threads(int v) { g_fin_threads = (v == 256 || v == 512) ? v : 1024; }
// threads of the lists-mode finalize launch: the option's, as long as every thread owns at most eight lists
static int fin_threads(int G) {
    int nth = g_fin_threads;
    while (nth < 1024 && (long long)G > 8LL * nth) nth <<= 1;
    return nth;
}

static int fin_pth(int G, int K, int nth) {
    if (G > 8 * nth) return 0;                           // the kernel ranks the heads from memory
    const int lpt = (G + nth - 1) / nth;
    if ((G + lpt - 1) / lpt >= 2 * K) return 0;
    int best = 0;
    double best_n = 1e300;
    for (int p = 1; p <= 16; p <<= 1) {
        const double f = (double)((K + p - 1) / p) / (double)G;      // share of the lists that must reach the bound
        if (f >= 0.95) continue;
        double lo = 0.0, hi = 64.0;
        for (int it = 0; it < 60; ++it) {
            const double lam = 0.5 * (lo + hi);
            double term = std::exp(-lam), below = 0.0;               // P(X < p)
            for (int i = 0; i < p; ++i) { below += term; term *= lam / (i + 1); }
            if (1.0 - below < f) lo = lam; else hi = lam;
        }
        const double n = hi * G;
        if (n < best_n) { best_n = n; best = p; }
    }
    return best;
}

hipError_t launch_finalize(const TableView& t, const float* qprep, const QInfo* qinfo, int64_t nq,
                           int dpad, const SelGeom& g, int G, int k,
                           const float* cand_s, const long long* cand_t, const float* cand_h,
                           long long* out_ids, float* out_scores, int* out_counts,
                           int* need_exact, long long* counters, const FlatBound& fb, const int* only, hipStream_t st,
                           int* host_flag) {
    FinParams p;
    p.t = t; p.q = qprep; p.qinfo = qinfo; p.dpad = dpad;
    p.K = g.K; p.C = g.C; p.G = G; p.k = k;
    p.only = only; p.host_flag = host_flag;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cand_h = cand_h;
    p.out_ids = out_ids; p.out_scores = out_scores; p.out_counts = out_counts;
    p.need_exact = need_exact; p.counters = counters;
    p.flat_cnt = nullptr; p.flat_cap = 0; p.flat_tau = nullptr; p.flat_lost = nullptr; p.flat_bad = nullptr;
    p.dd_mult = fb.dd_mult; p.extra_rel = fb.extra_rel; p.extra_abs = fb.extra_abs; p.l2_bias = 0.0; p.force_exact = fb.force_exact; p.floor_out = nullptr;
    const int nth = fin_threads(G);
    p.pth = fin_pth(G, g.K, nth);
    int kr = 1; while (kr < g.K) kr <<= 1;
    p.fin_cap = g.K > 512 ? FIN_CAP_LONG : FIN_CAP;
    const size_t lds = 1088 + (size_t)g.K * 12 + (size_t)kr * 12 + (size_t)p.fin_cap * 8;
    hipError_t e = allow_lds(finalize_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nq), dim3((unsigned)nth), lds, st, p);
    return hipGetLastError();
}

// K-th best fast key of G sorted K-lists per query (the scan of a row sample) -> floor_out[q].
hipError_t launch_lists_floor(const SelGeom& g, int G, int64_t nq, const float* cand_s, const long long* cand_t,
                              const float* cand_h, float* floor_out, hipStream_t st) {
    FinParams p;
    std::memset(&p, 0, sizeof p);
    p.K = g.K; p.C = g.C; p.G = G; p.k = 1;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cand_h = cand_h;
    p.floor_out = floor_out;
    const int nth = fin_threads(G);
    p.pth = fin_pth(G, g.K, nth);
    int kr = 1; while (kr < g.K) kr <<= 1;
    p.fin_cap = g.K > 512 ? FIN_CAP_LONG : FIN_CAP;
    const size_t lds = 1088 + (size_t)g.K * 12 + (size_t)kr * 12 + (size_t)p.fin_cap * 8;
    hipError_t e = allow_lds(finalize_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nq), dim3((unsigned)nth), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_finalize_flat(const TableView& t, const float* qprep, const QInfo* qinfo, int64_t nq,
                                int dpad, const SelGeom& g, int k,
                                const float* cand_s, const long long* cand_t, const int* cnt, int cap,
                                const float* tau, const int* lost, const int* bad, const FlatBound& fb,
                                long long* out_ids, float* out_scores, int* out_counts,
                                int* need_exact, long long* counters, hipStream_t st) {
    FinParams p;
    p.t = t; p.q = qprep; p.qinfo = qinfo; p.dpad = dpad;
    p.K = g.K; p.C = g.C; p.G = 0; p.k = k;
    p.only = nullptr;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cand_h = nullptr;
    p.out_ids = out_ids; p.out_scores = out_scores; p.out_counts = out_counts;
    p.need_exact = need_exact; p.counters = counters;
    p.flat_cnt = cnt; p.flat_cap = cap; p.flat_tau = tau; p.flat_lost = lost; p.flat_bad = bad; p.pth = 0; p.fin_cap = 0;
    p.dd_mult = fb.dd_mult; p.extra_rel = fb.extra_rel; p.extra_abs = fb.extra_abs; p.l2_bias = fb.l2_bias; p.force_exact = fb.force_exact; p.floor_out = nullptr;
    const size_t lds = 1088 + (size_t)BAND_CAP * 20;
    hipError_t e = allow_lds(finalize_band_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_band_kernel, dim3((unsigned)nq), dim3(1024), lds, st, p);
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

// need_exact[0] = number of flagged queries, need_exact[1..] = their indices (written by the
// finalize kernels, reset by prep_query).  Both kernels run on a small fixed grid and loop over
// the flagged list, so an empty list costs one early-exit launch however many queries there are.
__global__ void __launch_bounds__(1024)
exact_scan_kernel(ExactParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));
    const bool l2 = p.t.metric == VDB_METRIC_L2;
    const size_t rowbytes = (size_t)p.t.m * 16;
    unsigned* const ticket = (unsigned*)(smem + (size_t)W * p.C * 12);
    const int nflag = p.need_exact[0];
    for (int f = blockIdx.y; f < nflag; f += gridDim.y) {
        const int qi = p.need_exact[1 + f];
        const float* q = p.q + (size_t)qi * p.dpad;
        __syncthreads();                       // the previous query's lists have been written out
        if (threadIdx.x == 0) *ticket = 0u;
        __syncthreads();
        WaveSel sel;
        sel.init(S_all + (size_t)wave * p.C, T_all + (size_t)wave * p.C, p.C, p.K, lane);
        const long long TW = (long long)gridDim.x * W;
        constexpr int R = 4;                   // rows per wave and round
        // the rounds of the workgroup (W per TW * R rows) are drawn from a counter in LDS, as in the scan: the
        // waves of a workgroup get through their rows at different speeds
        const long long wrounds = (p.t.n + TW * R - 1) / (TW * R) * W;
        while (true) {
            unsigned tk = 0;
            if (lane == 0) tk = atomicAdd(ticket, 1u);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if ((long long)tk >= wrounds) break;
            const long long r0 = (long long)(tk / W) * TW * R + (long long)blockIdx.x * W + (tk % W);
            if (r0 >= p.t.n) continue;
            const char* rp[R];
            long long id[R];
            double sc[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const long long r = r0 + j * TW;
                const long long rc = r < p.t.n ? r : p.t.n - 1;
                rp[j] = p.t.rows + (size_t)rc * rowbytes;
                id[j] = p.t.ids[rc];
            }
            canon_score_rows_dt<R>(rp, q, p.t.d, p.t.dt, l2, lane, sc);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float fs = (float)sc[j];
                const float key = l2 ? -fs : fs;
                const bool pass = (lane == 0) && (r0 + j * TW < p.t.n) && better(key, id[j], sel.thr_s, sel.thr_t);
                sel.push(pass, key, id[j], lane);
            }
        }
        block_merge(sel, S_all, T_all, W, wave, lane);
        if (wave == 0) {
            const size_t o = ((size_t)f * gridDim.x + blockIdx.x) * p.K;
            for (int i = lane; i < p.K; i += 64) { p.cand_s[o + i] = S_all[i]; p.cand_t[o + i] = T_all[i]; }
        }
    }
}

// A stored id may equal TAG_NONE, which block_topk_global treats as "empty"; the exact pass
// therefore merges with its own loop that keeps every entry of the first min(K, rows) per block.
__global__ void __launch_bounds__(1024)
exact_finalize_kernel(ExactParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*     S_all = (float*)smem;
    long long* T_all = (long long*)(smem + (size_t)W * p.C * sizeof(float));
    const bool l2 = p.t.metric == VDB_METRIC_L2;
    const int nflag = p.need_exact[0];
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int qi = p.need_exact[1 + f];
        const long long ncand = (long long)p.G * p.K;
        const float* gs = p.cand_s + (size_t)f * ncand;
        const long long* gt = p.cand_t + (size_t)f * ncand;
        __syncthreads();
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
                atomicAdd((unsigned long long*)&p.counters[1], 1ull);
            }
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
    const unsigned gy = (unsigned)(nq < 8 ? nq : 8);
    hipLaunchKernelGGL(exact_scan_kernel, dim3((unsigned)G, gy), dim3((unsigned)g.W * 64), lds + 16, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned gf = (unsigned)(nq < 64 ? nq : 64);
    hipLaunchKernelGGL(exact_finalize_kernel, dim3(gf), dim3((unsigned)g.W * 64), lds, st, p);
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

#ifdef VDB_SCAN_STAMPS
extern "C" int vdb_debug_scan_waves(unsigned int* ends, unsigned int* xcc, int n) {
    if (hipMemcpyFromSymbol(ends, HIP_SYMBOL(vdb::g_scan_wave_end), (size_t)n * 4) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(xcc, HIP_SYMBOL(vdb::g_scan_wave_xcc), (size_t)n * 4) != hipSuccess) return -1;
    return 0;
}
#endif
