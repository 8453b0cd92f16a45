// Batched-query path: scores[rows, queries] = X . Q^T as an MFMA GEMM whose epilogue keeps only
// the scores that beat a per-query threshold.  The score matrix never reaches HBM.  Three kernels:
// gemm32s_kernel (16-bit MFMA; every pass but the first, every table type), and the two small
// kernels of the first threshold pass, gemm_thresh_kernel (fp32 MFMA, also the gemm32_split=0 mode
// of fp32 tables) and gemm16_dense_kernel (16-bit tables).
//
// Selection without in-kernel lists: a threshold tau_q that is known to be at or below the K-th
// best score of query q lets the epilogue drop almost every score with one compare and append the
// few survivors to a per-query candidate array with one global atomic.  tau_q comes from the same
// kernel run on a small strided sample of the row tiles (whose own threshold comes from an exact
// top-K over a few thousand rows), so three passes of geometrically growing size end with about
// two thousand candidates per query out of millions of rows.  If a candidate array overflows
// (rows ordered adversarially), the query is flagged and answered by the exact pass instead.
//
// gemm_thresh_kernel: 128 rows x 128 queries x 32 k per workgroup of four waves (2 x 2); each wave owns
// 64 x 64 as 2 x 2 MFMA tiles of v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate in
// k order).  Operand tiles are staged through registers into LDS rows padded by 16 bytes
// (conflict-free 16-byte fragment reads); the loads of the next k-step are issued before the
// MFMAs of the current one.  Workgroups that share a row range (one per query tile) are placed
// on one XCD so that the row tile is fetched from HBM once and served from that XCD's L2.
#include "vdb_internal.h"
#include "select.h"
#include "gemm_pp.h"
#include "gemm_pp384.h"

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_M = 128, GB_N = 128, GB_K = 32, GB_LD = GB_K + 4;   // LDS row = 36 floats

struct GemmParams {
    const float* X;          // [n][ldx]
    const float* n2;         // [n] (L2 only)
    const float* Q;          // [nq][dpad]
    const float* thr;        // [nq] a candidate must have key > thr
    long long    n;
    int          nq;
    int          dpad;       // floats per query row (multiple of 4, zero padded)
    int          ldx;        // floats per stored row
    int          l2;
    long long    tile0, tile_stride, ntiles;   // row tiles visited: tile0 + t * tile_stride, t < ntiles
    float*       cand_s;     // [nq][cap]
    long long*   cand_t;
    int*         cnt;        // [nq]
    int*         lost;       // [nq] set when a candidate could not be stored
    int          cap;
    RowBound     rb;         // L2: error term added to every key
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const f32x4*)p; }

// Source of the query operand for k positions past the end of a row.  The operand is switched by
// ADDRESS, never by masking loaded data: a select on a loaded register makes the wave wait for the
// load where it is issued, which turns every prefetch into a synchronous load.  A zero query
// operand is enough (the row operand then re-reads the row's first bytes: finite data times zero).
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};     // never written

template <bool DENSE>
__global__ void __launch_bounds__(256, 2)
gemm_thresh_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[GB_M * GB_LD];
    __shared__ __attribute__((aligned(16))) float Bs[GB_N * GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.y;
    const bool rbon = p.l2 && p.rb.qinfo != nullptr;
    const float ea = rbon ? p.rb.a : 0.f;
    const float es = rbon ? rowbound_s(p.rb, lane) : 0.f;
    const int nks = (p.dpad + GB_K - 1) / GB_K;
    const int d4 = p.dpad >> 2;                    // float4 units per query row (= per stored row)

    // staging map: float4 slot f = tid + 256*i, i < 4: row f/8, column block f%8
    int srow[4], sc4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int f = tid + 256 * i; srow[i] = f >> 3; sc4[i] = f & 7; }

    // query rows of this tile (clamped; masked in the epilogue)
    const float* qptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int q = qt * GB_N + srow[i];
        if (q > p.nq - 1) q = p.nq - 1;
        qptr[i] = p.Q + (size_t)q * p.dpad;
    }
    float thr[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = qt * GB_N + wn * 64 + ni * 32 + li;
        thr[ni] = (q < p.nq) ? p.thr[q] : __builtin_huge_valf();
    }

    for (long long t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const long long row0 = (p.tile0 + t * p.tile_stride) * GB_M;
        const float* xptr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long r = row0 + srow[i];
            if (r > p.n - 1) r = p.n - 1;
            xptr[i] = p.X + (size_t)r * p.ldx;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        f32x4 ra[4], rb[4];
        auto fetch = [&](int ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c4 = ks * (GB_K / 4) + sc4[i];
                const bool ok = c4 < d4;
                const int c = ok ? c4 * 4 : 0;
                ra[i] = ldg4(xptr[i] + c);
                rb[i] = ldg4(ok ? qptr[i] + c : (const float*)g_zero16);
            }
        };
        fetch(0);
        for (int ks = 0; ks < nks; ++ks) {
            __syncthreads();                               // previous k-step's fragment reads are done
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(f32x4*)&As[srow[i] * GB_LD + sc4[i] * 4] = ra[i];
                *(f32x4*)&Bs[srow[i] * GB_LD + sc4[i] * 4] = rb[i];
            }
            __syncthreads();
            if (ks + 1 < nks) fetch(ks + 1);               // in flight during the MFMAs below
#pragma unroll
            for (int kk = 0; kk < GB_K / 8; ++kk) {
                f32x4 a4[2], b4[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a4[mi] = *(const f32x4*)&As[(wm * 64 + mi * 32 + li) * GB_LD + kk * 8 + lh * 4];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b4[ni] = *(const f32x4*)&Bs[(wn * 64 + ni * 32 + li) * GB_LD + kk * 8 + lh * 4];
                // element s of the two lane halves forms one k-pair; A and B use the same pairing
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mi][s], b4[ni][s], acc[mi][ni], 0, 0, 0);
            }
        }
        // epilogue: C/D layout col = lane & 31 (query), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        // squared norms of this lane's 32 rows, fetched as one batch of independent loads
        float rn2[2][16];
        if (p.l2) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = row0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    rn2[mi][r] = p.n2[row < p.n ? row : p.n - 1];
                }
            if (rbon) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rn2[mi][r] = rowbound_adj(rn2[mi][r], ea, es);
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = qt * GB_N + wn * 64 + ni * 32 + li;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = row0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float key = acc[mi][ni][r];
                    if (p.l2) key = 2.f * key - rn2[mi][r];
                    if constexpr (DENSE) {
                        if (row < p.n && q < p.nq) { p.cand_s[(size_t)q * p.cap + row] = key; p.cand_t[(size_t)q * p.cap + row] = row; }
                    } else if (key > thr[ni] && row < p.n) {
                        const int pos = atomicAdd(&p.cnt[q], 1);
                        if (pos < p.cap) { p.cand_s[(size_t)q * p.cap + pos] = key; p.cand_t[(size_t)q * p.cap + pos] = row; }
                        else p.lost[q] = 1;
                    }
                }
            }
        }
    }
}

hipError_t launch_gemm_thresh(const TableView& t, const float* qprep, int nq, int dpad, const float* thr,
                              long long tile0, long long tile_stride, long long ntiles,
                              float* cand_s, long long* cand_t, int* cnt, int* lost, int cap, int dense,
                              int n_cu, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    GemmParams p;
    p.X = (const float*)t.rows; p.n2 = t.n2; p.rb = t.rb; p.Q = qprep; p.thr = thr; p.n = t.n; p.nq = nq;
    p.dpad = dpad; p.ldx = t.m * 4; p.l2 = t.metric == VDB_METRIC_L2 ? 1 : 0;
    p.tile0 = tile0; p.tile_stride = tile_stride; p.ntiles = ntiles;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cnt = cnt; p.lost = lost; p.cap = cap;
    const int qtiles = (nq + GB_N - 1) / GB_N;
    // two workgroups per CU are resident; gx a multiple of 8 keeps the query tiles of one row
    // range on one XCD (workgroups are dealt round-robin over the 8 XCDs by linear index)
    long long gx = (2LL * n_cu) / qtiles;
    gx = gx / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > ntiles) gx = ntiles;
    if (dense) hipLaunchKernelGGL(gemm_thresh_kernel<true>, dim3((unsigned)gx, (unsigned)qtiles), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gemm_thresh_kernel<false>, dim3((unsigned)gx, (unsigned)qtiles), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K-th best key per query among `count[q]` (or a fixed `count_all`) candidates: the threshold
// of the next, larger pass.  out[q] = max(in[q], K-th best) -- or in[q] when fewer than K exist.
__device__ __forceinline__ uint32_t fkey_g(float s) {
    const uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv_g(uint32_t k) {
    return (k & 0x80000000u) ? __uint_as_float(k & 0x7fffffffu) : __uint_as_float(~k);
}

__global__ void __launch_bounds__(1024)
kth_threshold_kernel(const float* __restrict__ cand_s, long long stride, const int* __restrict__ cnt,
                     long long count_all, int cap, int K, const float* __restrict__ thr_in,
                     float* __restrict__ thr_out, const float* __restrict__ n2, float out_limit, long long nrows,
                     const long long* __restrict__ cand_t) {
    __shared__ int hist[256];
    __shared__ int ctl[8];
    const int q = blockIdx.x;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    long long count = cnt ? (long long)cnt[q] : count_all;
    if (cnt && count > cap) count = cap;
    const float* gs = cand_s + (size_t)q * stride;
    const long long* gt = cand_t ? cand_t + (size_t)q * stride : nullptr;
    const float tin = thr_in ? thr_in[q] : -__builtin_huge_valf();
    uint32_t prefix = 0;
    int need = K;
    bool short_list = false;
    // Up to eight candidates per thread are read once, with unguarded loads (a guarded load becomes a branch
    // and a wait of its own), and stay in registers for the four passes; longer arrays are re-read per pass.
    constexpr int U = 8;
    const bool in_regs = count <= (long long)U * nth;
    float v[U];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = (long long)u * nth + tid;
            const long long ic = i < count ? i : 0;
            float x = gs[ic];
            if (n2 != nullptr) {
                // rows on the table's list of long rows do not take part (every query re-scores them anyway)
                const long long r = gt ? gt[ic] : ic;
                const bool known = r >= 0 && r < nrows;
                const float nn = n2[known ? r : 0];
                if (known && nn > out_limit) x = -__builtin_huge_valf();
            }
            v[u] = i < count ? x : -__builtin_huge_valf();
        }
    }
    // one increment per distinct bin and wave for the bins many lanes share (four rounds), the LDS atomics for the rest
    auto hist_add = [&](bool act, int bin) __attribute__((always_inline)) {
        unsigned long long todo = __ballot(act);
        for (int round = 0; todo && round < 4; ++round) {
            const int leader = __ffsll((long long)todo) - 1;
            const int b = __shfl(bin, leader);
            const unsigned long long same = __ballot(act && bin == b) & todo;
            if (lane == leader) atomicAdd(&hist[b], __popcll(same));
            todo &= ~same;
        }
        if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1);
    };
    for (int pass = 0; pass < 4; ++pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = 24 - 8 * pass;
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((long long)u * nth < count) {                  // block-uniform
                    const uint32_t key = fkey_g(v[u]);
                    const bool act = (v[u] > -__builtin_huge_valf()) && (pass == 0 || (key >> (shift + 8)) == prefix);
                    hist_add(act, (int)((key >> shift) & 255u));
                }
            }
        } else {
            for (long long i0 = 0; i0 < count; i0 += nth) {
                const long long i = i0 + tid;
                float x = (i < count) ? gs[i] : -__builtin_huge_valf();
                if (n2 != nullptr && i < count) {
                    const long long r = gt ? gt[i] : i;
                    if (r >= 0 && r < nrows && n2[r] > out_limit) x = -__builtin_huge_valf();
                }
                const uint32_t key = fkey_g(x);
                const bool act = (x > -__builtin_huge_valf()) && (pass == 0 || (key >> (shift + 8)) == prefix);
                hist_add(act, (int)((key >> shift) & 255u));
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
        if (pass == 0 && ctl[4] < K) { short_list = true; break; }
        prefix = (uint32_t)ctl[0];
        need = ctl[1];
    }
    if (tid == 0) {
        float out = tin;
        if (!short_list) { const float kth = fkey_inv_g(prefix); out = kth > tin ? kth : tin; }
        thr_out[q] = out;
    }
}

hipError_t launch_kth_threshold(const float* cand_s, long long stride, const int* cnt, long long count_all,
                                int cap, int K, const float* thr_in, float* thr_out, int nq, hipStream_t st,
                                const TableView* t, const long long* cand_t) {
    const bool side = t != nullptr && t->n_out > 0;
    hipLaunchKernelGGL(kth_threshold_kernel, dim3((unsigned)nq), dim3(1024), 0, st,
                       cand_s, stride, cnt, count_all, cap, K, thr_in, thr_out,
                       side ? t->n2 : (const float*)nullptr, side ? t->out_limit : 0.f, side ? t->n : 0LL,
                       side ? cand_t : (const long long*)nullptr);
    return hipGetLastError();
}

}  // namespace vdb

// ---------------------------------------------------------------------------------------------
// 16-bit MFMA (v_mfma_f32_32x32x16_{bf16,f16}: exact products, fp32 accumulate, 16x the fp32 MFMA
// rate).  The fp32 query is presented to the MFMA as hi + lo in the rows' 16-bit type
// (hi = round(q), lo = round(q - hi)); with SPLIT the lo part is accumulated too (relative query
// error 2^-16 for bf16, 2^-22 for fp16), without it only hi is used (bf16: 2^-9, fp16: 2^-11).
// That error is part of the bound the finalize step uses, so the result stays exact either way.
// First the small kernel of the first threshold pass over 16-bit tables (every score of the
// first rows is kept), then the kernel of all other passes over every table type.
namespace vdb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));

constexpr int HB_M = 128, HB_N = 128, HB_K = 64, HB_LD = HB_K + 8;   // LDS row = 72 halfs = 144 bytes

struct Gemm16Params {
    const unsigned short* X;     // [n][ldx] 16-bit elements
    const float* n2;
    const unsigned short* Qhi;   // [nq][ldq]
    const unsigned short* Qlo;   // [nq][ldq] (SPLIT only)
    int          ldq;            // halfs per query row (>= dpad, zero padded)
    const float* thr;
    long long    n;
    int          nq;
    int          dpad;           // elements per row (multiple of 8)
    int          ldx;
    int          l2;
    long long    tile0, tile_stride, ntiles;
    float*       cand_s;
    long long*   cand_t;
    int*         cnt;
    int*         lost;
    int          cap;
    RowBound     rb;
};

template <int DT>
__device__ __forceinline__ f32x16 mfma16(const u32x4g& a, const u32x4g& b, const f32x16& c) {
    if constexpr (DT == VDB_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT, bool SPLIT>
__global__ void __launch_bounds__(256, 2)
gemm16_dense_kernel(Gemm16Params p) {
    constexpr int D = 1;                           // k-steps of operands in registers
    __shared__ __attribute__((aligned(16))) unsigned short As[HB_M * HB_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bh[HB_N * HB_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bl[SPLIT ? HB_N * HB_LD : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.y;
    const bool rbon = p.l2 && p.rb.qinfo != nullptr;
    const float ea = rbon ? p.rb.a : 0.f;
    const float es = rbon ? rowbound_s(p.rb, lane) : 0.f;
    const int nks = (p.dpad + HB_K - 1) / HB_K;
    const int d8 = p.dpad >> 3;                    // 16-byte units per row

    int srow[4], sc8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int f = tid + 256 * i; srow[i] = f >> 3; sc8[i] = f & 7; }
    const unsigned short* qh[4];
    const unsigned short* ql[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int q = qt * HB_N + srow[i];
        if (q > p.nq - 1) q = p.nq - 1;
        qh[i] = p.Qhi + (size_t)q * p.ldq;
        ql[i] = SPLIT ? p.Qlo + (size_t)q * p.ldq : nullptr;
    }

    // The (tile, k-step) pairs of this workgroup form one flat sequence of stages.  A ring of D
    // stages of operands lives in registers: while stage s is multiplied out of LDS the loads of
    // stages s+1 .. s+D are in flight, across tile boundaries too.  Every stage issues exactly
    // one fetch (stages past the end read a zero block), so the number of loads issued after
    // any given one is the same on every path and the waits before the LDS stores can leave the
    // younger stages outstanding (loads complete in issue order).
    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long total = my_tiles * nks;
    const long long padded = (total + D - 1) / D * D;

    u32x4g ra[D][4], rh[D][4], rl[D][4];
    long long f_tile = blockIdx.x;          // fetch cursor
    int f_ks = 0;
    long long f_left = total;
    auto fetch = [&](int u) {
        const bool live = f_left > 0;
        const long long row0 = (p.tile0 + f_tile * p.tile_stride) * HB_M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long r = row0 + srow[i];
            if (r > p.n - 1) r = p.n - 1;
            const int c8 = f_ks * (HB_K / 8) + sc8[i];
            const bool ok = live && c8 < d8;
            const int c = ok ? c8 * 8 : 0;
            const unsigned short* xa = live ? p.X + (size_t)r * p.ldx + c : (const unsigned short*)g_zero16;
            ra[u][i] = __builtin_nontemporal_load((const u32x4g*)xa);                 // rows stream once
            rh[u][i] = *(const u32x4g*)(ok ? qh[i] + c : (const unsigned short*)g_zero16);
            if (SPLIT) rl[u][i] = *(const u32x4g*)(ok ? ql[i] + c : (const unsigned short*)g_zero16);
        }
        --f_left;
        if (++f_ks == nks) { f_ks = 0; f_tile += gridDim.x; }
    };
#pragma unroll
    for (int u = 0; u < D; ++u) fetch(u);

    f32x16 acc[2][2];
    long long c_tile = blockIdx.x;          // compute cursor
    int c_ks = 0;
    for (long long s0 = 0; s0 < padded; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(u32x4g*)&As[srow[i] * HB_LD + sc8[i] * 8] = ra[u][i];
                *(u32x4g*)&Bh[srow[i] * HB_LD + sc8[i] * 8] = rh[u][i];
                if (SPLIT) *(u32x4g*)&Bl[srow[i] * HB_LD + sc8[i] * 8] = rl[u][i];
            }
            __syncthreads();
            fetch(u);
            if (c_ks == 0) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < HB_K / 16; ++kk) {
                u32x4g a8[2], h8[2], l8[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a8[mi] = *(const u32x4g*)&As[(wm * 64 + mi * 32 + li) * HB_LD + kk * 16 + lh * 8];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    h8[ni] = *(const u32x4g*)&Bh[(wn * 64 + ni * 32 + li) * HB_LD + kk * 16 + lh * 8];
                    if (SPLIT) l8[ni] = *(const u32x4g*)&Bl[(wn * 64 + ni * 32 + li) * HB_LD + kk * 16 + lh * 8];
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni] = mfma16<DT>(a8[mi], h8[ni], acc[mi][ni]);
                        if (SPLIT) acc[mi][ni] = mfma16<DT>(a8[mi], l8[ni], acc[mi][ni]);
                    }
            }
            if (++c_ks == nks) {
                c_ks = 0;
                // padded stages beyond the last tile multiply zeros; with few k-steps per tile they
                // complete a tile that does not exist
                const long long row0 = (p.tile0 + c_tile * p.tile_stride) * HB_M;
                c_tile += gridDim.x;
                if (row0 >= p.n) continue;
                const long long left = p.n - row0;
                const int nloc = left < HB_M ? (int)left : HB_M;       // rows of this tile that exist
                const int q0 = qt * HB_N + wn * 64 + li;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int lr0 = wm * 64 + mi * 32 + 4 * lh;         // this lane's rows: lr0 + (r&3) + 8*(r>>2)
                    float rn2[16];
                    if (p.l2) {
                        const float* n2p = p.n2 + row0;                 // row0 < n here
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = lr0 + (r & 3) + 8 * (r >> 2);
                            rn2[r] = n2p[o < nloc ? o : 0];
                        }
                        if (rbon) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) rn2[r] = rowbound_adj(rn2[r], ea, es);
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int q = q0 + ni * 32;
                        const size_t base = (size_t)q * p.cap + (size_t)row0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float key = acc[mi][ni][r];
                            if (p.l2) key = 2.f * key - rn2[r];
                            const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                            if (lr < nloc && q < p.nq) { p.cand_s[base + lr] = key; p.cand_t[base + lr] = row0 + lr; }
                        }
                    }
                }
            }
        }
    }
}

// The kernel of every pass but the first, for every table type, on the 16-bit matrix unit.  A fast
// key only has to be accurate to the bound the finalize step uses, which re-scores every candidate
// within twice that bound of the k-th best key; so fp32 rows are rounded to bf16 in registers on the
// way to LDS and the query likewise (one product per term, relative error 2^-8), or both are
// presented as hi + lo (hi = rne(x), lo = rne(x - hi)) with three products xh*qh + xh*ql + xl*qh
// (3 * 2^-16); 16-bit rows go in as stored.  A workgroup of 8 waves owns 256 rows x 32..256 queries,
// LDS tiles are double-buffered with one barrier per k-step, two or four k-steps of operands ride
// in registers across tile boundaries.
// Elements per k-step: 64 over 16-bit rows, 32 over fp32 rows.  A row tile over 16-bit rows is then
// fetched in pieces of 128 instead of 64 contiguous bytes per row, which reads faster where the
// pass is bound by the rows (20 M x 768 fp16: 5.76 -> 5.28 ms at 2 queries, 6.79 -> 6.09 ms at 64),
// and halves the barriers per byte for the wide tiles (7.65 -> 6.52 ms at 128 queries, 10.1 -> 9.0 ms
// at 256, the latter with one k-step of operands in registers instead of two).  fp32 rows already
// come in 128-byte pieces: 256 measured no gain on the narrow tiles and spills on the 256-query tile
// (19.8 -> 21.2 ms).
constexpr int gemm32s_kstep(int nb, int mode) { return (mode == 1 || mode == 2) ? 64 : 32; }
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2g __attribute__((ext_vector_type(2)));

struct Gemm32sParams {
    const void*  X;              // [n][ldx] rows: fp32 (MODE 0) or 16-bit (MODE 1, 2)
    const float* n2;
    const unsigned short* Qhi;   // [nq][qstride], zero padded
    const unsigned short* Qlo;
    const float* thr;
    long long    n;
    int          nq;
    int          dpad;           // elements per row (multiple of 4 for fp32 rows, of 8 for 16-bit rows)
    int          qstride;        // halfs per query row: a multiple of 64 (whole k-steps of every tile shape), >= dpad
    int          ldx;
    int          l2;
    long long    ntiles;         // 256-row tiles visited: tile t * tile_stride, t < ntiles
    long long    tile_stride;
    float*       cand_s;
    long long*   cand_t;
    int*         cnt;
    int*         lost;
    int          cap;
    RowBound     rb;
};

// 4 fp32 -> 4 bf16 hi (packed in 8 bytes) and 4 bf16 lo
__device__ __forceinline__ void split4_bf16(const f32x4& x, u32x2g& hi, u32x2g& lo) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hb = (__bf16)x[e];
        const float r = x[e] - (float)hb;
        const __bf16 lb = (__bf16)r;
        h[e] = __builtin_bit_cast(unsigned short, hb);
        l[e] = __builtin_bit_cast(unsigned short, lb);
    }
    hi[0] = (unsigned int)h[0] | ((unsigned int)h[1] << 16); hi[1] = (unsigned int)h[2] | ((unsigned int)h[3] << 16);
    lo[0] = (unsigned int)l[0] | ((unsigned int)l[1] << 16); lo[1] = (unsigned int)l[2] | ((unsigned int)l[3] << 16);
}

// NB = 32-query blocks per workgroup tile: 8 (256 queries, 64 x 128 per wave; single-product modes
// only, where the pass is bound by re-reading the rows once per query tile), 4 (128 queries, 64 x 64
// per wave), 2 (64 queries, 64 x 32 per wave) or 1 (32 queries, 32 x 32 per wave; small batches then
// cost little more than reading the rows once).  The tile is 256 rows in every case.
// MODE 0: fp32 rows split into bf16 hi + lo, split query, three products per term;
// MODE 1: 16-bit rows as stored, split query, two products; MODE 2: 16-bit rows, query hi only;
// MODE 3: fp32 rows rounded to bf16 on the way to LDS, query hi only, one product per term (relative
// error 2^-8 of ||x|| ||q||: the re-scored band is wider, the pass three times cheaper).
// DT is the 16-bit type the MFMA multiplies in (bf16 for MODE 0, the table's type otherwise).
// 4 fp32 -> 4 bf16 (round to nearest even), packed in 8 bytes
__device__ __forceinline__ u32x2g round4_bf16(const f32x4& x) {
    unsigned short h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = __builtin_bit_cast(unsigned short, (__bf16)x[e]);
    u32x2g r;
    r[0] = (unsigned int)h[0] | ((unsigned int)h[1] << 16);
    r[1] = (unsigned int)h[2] | ((unsigned int)h[3] << 16);
    return r;
}

template <int NB, int DT, int MODE, int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 2 : 1)
gemm32s_kernel(Gemm32sParams p) {
    constexpr int TM = NT / 2;                     // rows per tile: 256 (8 waves) or 128 (4 waves, two workgroups per CU)
    constexpr bool F32ROWS = MODE == 0 || MODE == 3;
    constexpr bool ALO = MODE == 0;                // rows have a lo part
    constexpr bool BLO = MODE == 0 || MODE == 1;   // queries have a lo part
    constexpr int KS = gemm32s_kstep(NB, MODE);
    constexpr int LD = KS + 8;                     // LDS row in halfs: 80 or 144 bytes, conflict-free b128 reads
    constexpr int EPC = F32ROWS ? 4 : 8;           // elements per 16-byte chunk
    constexpr int CPR = KS / EPC;                  // 16-byte chunks per tile row and k-step
    constexpr int NA = TM * CPR / NT;              // 16-byte row chunks staged per thread and k-step
    constexpr int QPC = KS / 8;                    // 16-byte chunks per query and k-step
    constexpr int RD = (NB == 8 && KS == 64) ? 1 : (NB >= 4 || KS == 64) ? 2 : 4;   // k-steps of operands in registers (narrow tiles have
                                                   // little MFMA time per step to cover the load latency)
    constexpr int TN = 32 * NB;                    // queries per tile
    constexpr int NI = NB == 8 ? 4 : NB == 4 ? 2 : 1;   // 32-query blocks per wave
    constexpr int QCH = TN * QPC > NT ? TN * QPC / NT : 1;   // 16-byte query chunks staged per thread and k-step
    constexpr int RDB = QCH >= 4 ? 1 : RD;         // k-steps of query chunks in registers (they come from L2)
    constexpr int MI = NB == 1 ? 1 : 2;            // 32-row blocks per wave
    constexpr int WN = NB / NI;                    // waves across the queries
    extern __shared__ __attribute__((aligned(16))) unsigned short ssm[];
    unsigned short* Ah = ssm;                                  // [2][TM * LD]
    unsigned short* Al = Ah + 2 * TM * LD;                  // rows with a lo part only
    unsigned short* Bh = Al + (ALO ? 2 * TM * LD : 0);      // [2][TN * LD]
    unsigned short* Bl = Bh + 2 * TN * LD;                  // queries with a lo part only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.y;
    const bool rbon = p.l2 && p.rb.qinfo != nullptr;
    const float ea = rbon ? p.rb.a : 0.f;
    const float es = rbon ? rowbound_s(p.rb, lane) : 0.f;
    const int nks = (p.dpad + KS - 1) / KS;
    const int dch = p.dpad / EPC;                  // 16-byte chunks per row

    // staging: row chunks f = tid + NT * i (i < NA): tile row f / CPR, chunk f % CPR;
    //          query chunks g = tid + NT * j (j < QCH): tile query g / QPC, 8-half chunk g % QPC
    // NT is a multiple of CPR: chunk i of a thread lies NT / CPR rows below chunk 0, same column
    static_assert(NT % CPR == 0, "staging map");
    const int arow0 = tid / CPR, ac0 = tid % CPR;
    const int brow = tid / QPC, bc = tid % QPC;    // chunk j: tile query brow + (NT / QPC) * j
    const bool bstage = tid < TN * QPC;            // threads that stage a query chunk
    const unsigned short* qh[QCH];
    const unsigned short* ql[QCH];
#pragma unroll
    for (int j = 0; j < QCH; ++j) {
        int qb = qt * TN + (bstage ? brow + (NT / QPC) * j : 0);
        if (qb > p.nq - 1) qb = p.nq - 1;
        qh[j] = p.Qhi + (size_t)qb * p.qstride;
        ql[j] = p.Qlo + (size_t)qb * p.qstride;
    }
    float thr[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int q = qt * TN + (wn * NI + ni) * 32 + li;
        thr[ni] = (q < p.nq) ? p.thr[q] : __builtin_huge_valf();
    }

    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long total = my_tiles * nks;
    constexpr int UN = RD < 2 ? 2 : RD;            // stages per trip of the unrolled loop (LDS buffers alternate)
    const long long padded = (total + UN - 1) / UN * UN;

    u32x4g ra[RD][NA];                             // raw 16 bytes: 4 fp32 or 8 16-bit values
    u32x4g rh[RDB][QCH], rl[RDB][QCH];
    long long f_tile = blockIdx.x;
    int f_ks = 0;
    // No load below selects between two base addresses: the compiler turns such a select into
    // branches whose join waits for every outstanding load (s_waitcnt vmcnt(0)), which serialises
    // the prefetch.  Padded stages beyond the last tile re-read the table's last row (their tile is
    // never written out), chunks past the row's end re-read the row's first bytes (the query
    // operand, padded with zeros to a multiple of the k-step, is zero there).
    auto fetch = [&](int u) {
        const long long row0 = f_tile * p.tile_stride * TM;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            long long r = row0 + (arow0 + (NT / CPR) * i);
            if (r > p.n - 1) r = p.n - 1;
            const int ch = f_ks * CPR + ac0;
            const char* xa = (const char*)p.X + ((size_t)r * p.ldx + (ch < dch ? ch * EPC : 0)) * (F32ROWS ? 4 : 2);
            // rows stream once past a single query tile; several query tiles of one row range run
            // on one XCD side by side and meet in its L2, so their loads leave the lines there
            ra[u][i] = gridDim.y == 1 ? __builtin_nontemporal_load((const u32x4g*)xa) : *(const u32x4g*)xa;
        }
        if (++f_ks == nks) { f_ks = 0; f_tile += gridDim.x; }
    };
    // the query chunks of a k-step are the same for every tile: a cursor of their own, cycling over
    // the k-steps
    int b_ks = 0;
    auto fetch_q = [&](int v) {
        const int c8 = b_ks * QPC + bc;     // < qstride / 8: query rows are padded to whole k-steps
        if (TN * QPC >= NT || bstage) {
#pragma unroll
            for (int j = 0; j < QCH; ++j) {
                rh[v][j] = *(const u32x4g*)(qh[j] + c8 * 8);
                if (BLO) rl[v][j] = *(const u32x4g*)(ql[j] + c8 * 8);
            }
        }
        if (++b_ks == nks) b_ks = 0;
    };
    // Issued in the order of the loop below and kept in it (the scheduler would otherwise be free
    // to reorder these independent loads): the loop's wait for a stage's operands is a counted
    // s_waitcnt vmcnt(N), valid only if the younger stages' loads really are younger on every path
    // into the loop; a prologue in another order degrades it to vmcnt(0), which drains the prefetch.
#pragma unroll
    for (int u = 0; u < RD; ++u) {
        if (RDB == RD || u == RD - 1) fetch_q(RDB == 1 ? 0 : u);
        fetch(u);
        __builtin_amdgcn_sched_barrier(0);
    }

    // stage(w, b): operands of register set w go to LDS buffer b (fp32 rows converted on the way),
    // then the set is refilled with the operands RD k-steps ahead
    auto stage = [&](int w, int b) {
        unsigned short* Ahw = Ah + b * (TM * LD);
        unsigned short* Alw = Al + b * (TM * LD);
        unsigned short* Bhw = Bh + b * (TN * LD);
        unsigned short* Blw = Bl + b * (TN * LD);
        const int wb = RDB == 1 ? 0 : w;            // query register set
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (MODE == 0) {
                u32x2g h2, l2;
                split4_bf16(__builtin_bit_cast(f32x4, ra[w][i]), h2, l2);
                *(u32x2g*)&Ahw[(arow0 + (NT / CPR) * i) * LD + ac0 * 4] = h2;
                *(u32x2g*)&Alw[(arow0 + (NT / CPR) * i) * LD + ac0 * 4] = l2;
            } else if constexpr (MODE == 3) {
                *(u32x2g*)&Ahw[(arow0 + (NT / CPR) * i) * LD + ac0 * 4] = round4_bf16(__builtin_bit_cast(f32x4, ra[w][i]));
            } else {
                *(u32x4g*)&Ahw[(arow0 + (NT / CPR) * i) * LD + ac0 * 8] = ra[w][i];
            }
        }
        if (TN * QPC >= NT || bstage) {
#pragma unroll
            for (int j = 0; j < QCH; ++j) {
                *(u32x4g*)&Bhw[(brow + (NT / QPC) * j) * LD + bc * 8] = rh[wb][j];
                if (BLO) *(u32x4g*)&Blw[(brow + (NT / QPC) * j) * LD + bc * 8] = rl[wb][j];
            }
        }
        // queries first: with one k-step of them in registers they are needed a stage before the
        // rows issued next to them, and a counted wait covers everything older than what it waits for
        fetch_q(wb);
        fetch(w);
    };
    stage(0, 0);

    // One barrier per k-step.  The operands of k-step t+1 are written to the other LDS buffer
    // between the two halves of k-step t's MFMA work, not before the barrier: the two waves of a
    // SIMD drift half a step apart, and one's conversions, LDS stores and load issue run under
    // the other's MFMAs.  (Buffer (t+1)&1 was last read in k-step t-1, which every wave finished
    // before the barrier of k-step t.)
    f32x16 acc[MI][NI];
    long long c_tile = blockIdx.x;
    int c_ks = 0;
    for (long long s0 = 0; s0 < padded; s0 += UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const unsigned short* Ahu = Ah + (u & 1) * (TM * LD);
            const unsigned short* Alu = Al + (u & 1) * (TM * LD);
            const unsigned short* Bhu = Bh + (u & 1) * (TN * LD);
            const unsigned short* Blu = Bl + (u & 1) * (TN * LD);
            __syncthreads();
            if (c_ks == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < KS / 16; ++kk) {
                if (kk == KS / 32) stage((u + 1) % RD, (u + 1) & 1);
                u32x4g ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    ah[mi] = *(const u32x4g*)&Ahu[((wm * MI + mi) * 32 + li) * LD + kk * 16 + lh * 8];
                    if (ALO) al[mi] = *(const u32x4g*)&Alu[((wm * MI + mi) * 32 + li) * LD + kk * 16 + lh * 8];
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    bh[ni] = *(const u32x4g*)&Bhu[((wn * NI + ni) * 32 + li) * LD + kk * 16 + lh * 8];
                    if (BLO) bl[ni] = *(const u32x4g*)&Blu[((wn * NI + ni) * 32 + li) * LD + kk * 16 + lh * 8];
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if (ALO) acc[mi][ni] = mfma16<DT>(al[mi], bh[ni], acc[mi][ni]);
                        if (BLO) acc[mi][ni] = mfma16<DT>(ah[mi], bl[ni], acc[mi][ni]);
                        acc[mi][ni] = mfma16<DT>(ah[mi], bh[ni], acc[mi][ni]);
                    }
                // scheduling hint: interleave the LDS reads with the MFMAs of this block (2 % on the
                // 256-query tile; variant 1 of the hint sends the compiler out of memory on this kernel)
                __builtin_amdgcn_iglp_opt(0);
            }
            if (++c_ks == nks) {
                c_ks = 0;
                // padded stages beyond the last tile: with few k-steps per tile they
                // complete a tile that does not exist
                const long long row0 = c_tile * p.tile_stride * TM;
                c_tile += gridDim.x;
                if (row0 >= p.n) continue;
                const long long left = p.n - row0;
                const int nloc = left < TM ? (int)left : TM;
                const int q0 = qt * TN + wn * NI * 32 + li;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int lr0 = (wm * MI + mi) * 32 + 4 * lh;     // this lane's rows: lr0 + (r&3) + 8*(r>>2)
                    float rn2[16];
                    if (p.l2) {
                        const float* n2p = p.n2 + row0;                 // row0 < n here
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = lr0 + (r & 3) + 8 * (r >> 2);
                            rn2[r] = n2p[o < nloc ? o : 0];
                        }
                        if (rbon) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) rn2[r] = rowbound_adj(rn2[r], ea, es);
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        // one compare for the 16 scores of a lane first: a lane meets a candidate in
                        // about one tile out of 300
                        float best = -__builtin_huge_valf();
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float key = acc[mi][ni][r];
                            if (p.l2) key = 2.f * key - rn2[r];
                            best = fmaxf(best, key);
                        }
                        if (__builtin_expect(best > thr[ni], 0)) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float key = acc[mi][ni][r];
                                if (p.l2) key = 2.f * key - rn2[r];
                                const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                                if (key > thr[ni] && lr < nloc) {
                                    int lro = lr, q = q0 + ni * 32;
                                    asm volatile("" : "+v"(lro), "+v"(q));
                                    const int pos = atomicAdd(&p.cnt[q], 1);
                                    if (pos < p.cap) { p.cand_s[(size_t)q * p.cap + pos] = key; p.cand_t[(size_t)q * p.cap + pos] = row0 + lro; }
                                    else p.lost[q] = 1;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int NB, int DT, int MODE, int NT>
static hipError_t launch_gemm32s_nb(const Gemm32sParams& p0, int nq, int n_cu, hipStream_t st) {
    constexpr int TM = NT / 2;
    const size_t lds = (size_t)2 * ((MODE == 0 ? 2 : 1) * TM + (MODE == 0 || MODE == 1 ? 2 : 1) * 32 * NB) * (gemm32s_kstep(NB, MODE) + 8) * sizeof(unsigned short);
    static bool allowed = false;
    if (!allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm32s_kernel<NB, DT, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        allowed = true;
    }
    Gemm32sParams p = p0;
    // every tile_stride-th TM-row tile (1 = the full pass; larger = a strided sample of the rows)
    p.ntiles = ((p.n + TM - 1) / TM + p.tile_stride - 1) / p.tile_stride;
    const int qtiles = (nq + 32 * NB - 1) / (32 * NB);
    long long gx = (long long)n_cu * (512 / NT) / qtiles;     // one workgroup of 8 waves or two of 4 per CU
    gx = gx / 8 * 8;                                          // query tiles of one row range share an XCD
    if (gx < 8) gx = 8;
    if (gx > p.ntiles) gx = p.ntiles;
    hipLaunchKernelGGL((gemm32s_kernel<NB, DT, MODE, NT>), dim3((unsigned)gx, (unsigned)qtiles), dim3(NT), lds, st, p);
    return hipGetLastError();
}

static int g_gemm32s_threads = 512;      // workgroup size of the 256-query tile: 256 (two workgroups per CU) or 512
void set_gemm32s_threads(int v) { g_gemm32s_threads = v == 512 ? 512 : 256; }

template <int DT, int MODE>
static hipError_t launch_gemm32s_mode(const Gemm32sParams& p, int nq, int n_cu, hipStream_t st) {
    if (nq <= 32) return launch_gemm32s_nb<1, DT, MODE, 512>(p, nq, n_cu, st);
    if (nq <= 64) return launch_gemm32s_nb<2, DT, MODE, 512>(p, nq, n_cu, st);
    if constexpr (MODE == 2 || MODE == 3) {
        if (nq > 128) return g_gemm32s_threads == 256 ? launch_gemm32s_nb<8, DT, MODE, 256>(p, nq, n_cu, st)
                                                      : launch_gemm32s_nb<8, DT, MODE, 512>(p, nq, n_cu, st);
    }
    return launch_gemm32s_nb<4, DT, MODE, 512>(p, nq, n_cu, st);
}

// Full pass over a table with the kernel above.  fp32 tables: qhi/qlo are bf16 (launch_split_query16,
// stride qstride); `split` = rows and query as hi + lo (three products), else both rounded to bf16.
// 16-bit tables: qhi/qlo in the table's type, `split` says whether qlo is used.
hipError_t launch_gemm32s(const TableView& t, const unsigned short* qhi, const unsigned short* qlo, int qstride,
                          int split, int nq, int dpad, const float* thr, long long tile_stride,
                          float* cand_s, long long* cand_t,
                          int* cnt, int* lost, int cap, int n_cu, hipStream_t st) {
    Gemm32sParams p;
    p.X = t.rows; p.n2 = t.n2; p.rb = t.rb; p.Qhi = qhi; p.Qlo = qlo; p.thr = thr; p.n = t.n; p.nq = nq;
    p.dpad = dpad; p.qstride = qstride; p.ldx = t.dt == VDB_F32 ? t.m * 4 : t.m * 8; p.l2 = t.metric == VDB_METRIC_L2 ? 1 : 0;
    p.tile_stride = tile_stride < 1 ? 1 : tile_stride;
    p.ntiles = 0;                                    // set by the launcher of the chosen tile shape
    p.cand_s = cand_s; p.cand_t = cand_t; p.cnt = cnt; p.lost = lost; p.cap = cap;
    if (t.dt == VDB_F32) return split ? launch_gemm32s_mode<VDB_BF16, 0>(p, nq, n_cu, st) : launch_gemm32s_mode<VDB_BF16, 3>(p, nq, n_cu, st);
    if (t.dt == VDB_BF16) return split ? launch_gemm32s_mode<VDB_BF16, 1>(p, nq, n_cu, st) : launch_gemm32s_mode<VDB_BF16, 2>(p, nq, n_cu, st);
    return split ? launch_gemm32s_mode<VDB_F16, 1>(p, nq, n_cu, st) : launch_gemm32s_mode<VDB_F16, 2>(p, nq, n_cu, st);
}

// One threshold pass over a 16-bit table: the first pass (`dense`: every score of the leading rows
// is kept, slot = row) on the small kernel above, every other pass on the kernel shared with the
// fp32 tables, which visits every tile_stride-th row tile.
hipError_t launch_gemm16_thresh(const TableView& t, const unsigned short* qhi, const unsigned short* qlo, int qstride,
                                int split, int nq, int dpad, const float* thr,
                                long long tile_stride, long long ntiles,
                                float* cand_s, long long* cand_t, int* cnt, int* lost, int cap, int dense,
                                int n_cu, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    if (!dense)
        return launch_gemm32s(t, qhi, qlo, qstride, split, nq, dpad, thr, tile_stride, cand_s, cand_t, cnt, lost, cap, n_cu, st);
    Gemm16Params p;
    p.X = (const unsigned short*)t.rows; p.n2 = t.n2; p.rb = t.rb; p.Qhi = qhi; p.Qlo = qlo; p.thr = thr; p.n = t.n; p.nq = nq;
    p.dpad = dpad; p.ldq = qstride; p.ldx = t.m * 8; p.l2 = t.metric == VDB_METRIC_L2 ? 1 : 0;
    p.tile0 = 0; p.tile_stride = tile_stride; p.ntiles = ntiles;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cnt = cnt; p.lost = lost; p.cap = cap;
    const int qtiles = (nq + HB_N - 1) / HB_N;
    long long gx = (2LL * n_cu) / qtiles;                                       // two workgroups per CU
    gx = gx / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)qtiles), block(256);
    if (t.dt == VDB_BF16) {
        if (split) hipLaunchKernelGGL((gemm16_dense_kernel<VDB_BF16, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm16_dense_kernel<VDB_BF16, false>), grid, block, 0, st, p);
    } else {
        if (split) hipLaunchKernelGGL((gemm16_dense_kernel<VDB_F16, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm16_dense_kernel<VDB_F16, false>), grid, block, 0, st, p);
    }
    return hipGetLastError();
}

// fp32 prepared queries -> hi/lo parts in the table's 16-bit type; a query whose hi part is not
// finite (fp16 overflow) is flagged as lost so that the exact pass answers it.
template <int DT>
__global__ void __launch_bounds__(256)
split_query16_kernel(const float* __restrict__ q, int dpad, int stride_out, unsigned short* __restrict__ hi,
                     unsigned short* __restrict__ lo, int* __restrict__ lost) {
    const size_t base = (size_t)blockIdx.x * dpad;
    const size_t obase = (size_t)blockIdx.x * stride_out;
    bool bad = false;
    for (int j = threadIdx.x; j < stride_out; j += blockDim.x) {
        const float x = j < dpad ? q[base + j] : 0.f;          // zero padding up to the output stride
        unsigned short h, l;
        float hf;
        if constexpr (DT == VDB_BF16) {
            const uint32_t u = __float_as_uint(x);
            h = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            hf = __uint_as_float((uint32_t)h << 16);
            const uint32_t v = __float_as_uint(x - hf);
            l = (unsigned short)((v + 0x7fffu + ((v >> 16) & 1u)) >> 16);
        } else {
            const _Float16 hh = (_Float16)x;
            hf = (float)hh;
            const _Float16 ll = (_Float16)(x - hf);
            __builtin_memcpy(&h, &hh, 2);
            __builtin_memcpy(&l, &ll, 2);
        }
        if (!(fabsf(hf) <= 3.0e38f)) bad = true;
        hi[obase + j] = h;
        lo[obase + j] = l;
    }
    if (bad) lost[blockIdx.x] = 1;
}

hipError_t launch_split_query16(int dt, const float* qprep, int nq, int dpad, int stride_out, unsigned short* hi,
                                unsigned short* lo, int* lost, hipStream_t st) {
    if (dt == VDB_BF16) hipLaunchKernelGGL((split_query16_kernel<VDB_BF16>), dim3((unsigned)nq), dim3(256), 0, st, qprep, dpad, stride_out, hi, lo, lost);
    else hipLaunchKernelGGL((split_query16_kernel<VDB_F16>), dim3((unsigned)nq), dim3(256), 0, st, qprep, dpad, stride_out, hi, lo, lost);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Ping-pong kernel (gemm_pp.h) for batches of more than 128 queries.
// The queries go in as a tiled 16-bit image: [ceil(nq/256)][nks] blocks of 256 queries x 32 elements in
// the LDS stage layout (pp_tiled_index), zero for queries past nq and elements past dpad; L2 tables
// get -2q (exact: a power of two), so that the kernel accumulates ||x||^2 - 2 x.q.
template <int DT>
__global__ void __launch_bounds__(256)
tile_query16_kernel(const float* __restrict__ q, int nq, int dpad, int nks, float scale,
                    unsigned short* __restrict__ image, int* __restrict__ bad) {
    const int qi = blockIdx.x;                     // < ceil(nq/256)*256
    bool isbad = false;
    for (int j = threadIdx.x; j < nks * 32; j += blockDim.x) {
        const float x = (qi < nq && j < dpad) ? q[(size_t)qi * dpad + j] * scale : 0.f;
        unsigned short h;
        float hf;
        if constexpr (DT == VDB_BF16) {
            const __bf16 b = (__bf16)x;
            h = __builtin_bit_cast(unsigned short, b);
            hf = (float)b;
        } else {
            const _Float16 hh = (_Float16)x;
            hf = (float)hh;
            __builtin_memcpy(&h, &hh, 2);
        }
        if (!(fabsf(hf) <= 3.0e38f)) isbad = true;
        image[pp_tiled_index(qi, j, nks)] = h;
    }
    if (isbad && qi < nq) bad[qi] = 1;
}

static int pp_nks(int dpad) { return (dpad + 31) / 32; }

bool pp_eligible(const TableView& t, int dpad) {
    if (!t.rows16) return false;
    // every k-step reads 32 elements of a row: the row must hold them (mirror rows are padded with
    // zeros; 16-bit rows qualify when their length is a whole number of k-steps)
    return (long long)pp_nks(dpad) * 32 <= t.ld16;
}

size_t pp_query_image_bytes(int nq, int dpad) {
    return (size_t)((nq + 255) / 256) * pp_nks(dpad) * 16384;
}

hipError_t launch_tile_query16(int dt, int l2, const float* qprep, int nq, int dpad, unsigned short* image,
                               int* bad, hipStream_t st) {
    const int rows = (nq + 255) / 256 * 256;
    const float scale = l2 ? -2.f : 1.f;
    if (dt == VDB_BF16) hipLaunchKernelGGL((tile_query16_kernel<VDB_BF16>), dim3((unsigned)rows), dim3(256), 0, st, qprep, nq, dpad, pp_nks(dpad), scale, image, bad);
    else hipLaunchKernelGGL((tile_query16_kernel<VDB_F16>), dim3((unsigned)rows), dim3(256), 0, st, qprep, nq, dpad, pp_nks(dpad), scale, image, bad);
    return hipGetLastError();
}

// Rows per tile of the ping-pong pass.  384 (gemm_pp384.h) for launches of one query tile (up to 256
// queries), where every row byte comes from HBM exactly once and the wider tile saves a third of the query
// traffic: 7 % faster.  256 (gemm_pp.h) when several query tiles share the row tiles through an XCD's L2: the
// 32 workgroups of an XCD then work on 8 row tiles at once, 4.6 MB of 384-row tiles against 4 MB of L2, and
// the fabric traffic of a pass rose from 1.11 x to 1.33 x the rows (PMC FETCH_SIZE) for no gain in time.
// Option gemm_pp_tile: 0 = that rule (default), 256 or 384 = that kernel for every launch.
static int g_pp_tile_rows = 0;
void set_pp_tile_rows(int v) { g_pp_tile_rows = v == PP_TM ? PP_TM : (v == PW_TM ? PW_TM : 0); }
int pp_tile_rows(int nq) { return g_pp_tile_rows ? g_pp_tile_rows : (nq <= PW_TN ? PW_TM : PP_TM); }

static long long pp_grid_x(const GemmPPParams& p, int n_cu, int qtiles) {
    long long gx = (long long)n_cu / qtiles;       // one workgroup per CU
    gx = gx / 8 * 8;                               // the query tiles of one row range share an XCD (and its L2)
    if (gx < 8) gx = 8;
    if (gx > p.ntiles) gx = p.ntiles;
    return gx;
}

template <int DT, bool L2M>
static hipError_t launch_gemm_pp_v(const GemmPPParams& p, int n_cu, hipStream_t st) {
    constexpr int lds = L2M ? PP_LDS_L2 : PP_LDS_IP;
    static bool allowed = false;
    if (!allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<DT, L2M, false, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        allowed = true;
    }
    const int qtiles = (p.nq + PP_TN - 1) / PP_TN;
    const long long gx = pp_grid_x(p, n_cu, qtiles);
    hipLaunchKernelGGL((gemm_pp_kernel<DT, L2M, false, 16, false>), dim3((unsigned)gx, (unsigned)qtiles), dim3(512), lds, st, p);
    return hipGetLastError();
}

// 384-row tiles, row-major rows, query loads ahead of the row loads of a stage
template <int DT, bool L2M>
static hipError_t launch_gemm_pp384_v(const GemmPPParams& p, int n_cu, hipStream_t st) {
    static bool allowed = false;
    if (!allowed) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pp384_kernel<DT, L2M, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e != hipSuccess) return e;
        allowed = true;
    }
    const int qtiles = (p.nq + PW_TN - 1) / PW_TN;
    const long long gx = pp_grid_x(p, n_cu, qtiles);
    hipLaunchKernelGGL((gemm_pp384_kernel<DT, L2M, false, false, 2>), dim3((unsigned)gx, (unsigned)qtiles), dim3(512), PW_LDS, st, p);
    return hipGetLastError();
}

hipError_t launch_gemm_pp(const TableView& t, const unsigned short* qimage, int nq, int dpad, const float* thr,
                          long long tile_stride, long long tile0, long long tile_count,
                          float* cand_s, long long* cand_t, int* cnt, int* lost, int cap,
                          int n_cu, hipStream_t st) {
    const int TR = pp_tile_rows(nq);
    GemmPPParams p;
    p.X = t.rows16; p.n2 = t.n2; p.rb = t.rb; p.Q = qimage; p.thr = thr; p.n = t.n; p.ldx = t.ld16; p.nq = nq;
    p.qstride = 0; p.nks = pp_nks(dpad);
    p.tile_stride = tile_stride < 1 ? 1 : tile_stride;
    p.ntiles = ((t.n + TR - 1) / TR + p.tile_stride - 1) / p.tile_stride;
    // a range of the visited tiles: [tile0, tile0 + tile_count) of them (tile_count < 0: to the end)
    p.tile0 = tile0 * p.tile_stride;
    p.ntiles -= tile0;
    if (tile_count >= 0 && tile_count < p.ntiles) p.ntiles = tile_count;
    p.cand_s = cand_s; p.cand_t = cand_t; p.cnt = cnt; p.lost = lost; p.cap = cap; p.dense = nullptr;
    if (p.ntiles <= 0) return hipSuccess;
    // the kernels address a row with 32-bit byte offsets inside a tile and queries with 32-bit indices
    if (t.ld16 <= 0 || t.ld16 > (1 << 20) || nq <= 0) return hipErrorInvalidValue;
    const bool l2 = t.metric == VDB_METRIC_L2;
    const int dt = t.dt == VDB_F16 ? VDB_F16 : VDB_BF16;       // fp32 tables multiply their bf16 mirror
    if (TR == PW_TM) {
        if (dt == VDB_BF16) return l2 ? launch_gemm_pp384_v<VDB_BF16, true>(p, n_cu, st) : launch_gemm_pp384_v<VDB_BF16, false>(p, n_cu, st);
        return l2 ? launch_gemm_pp384_v<VDB_F16, true>(p, n_cu, st) : launch_gemm_pp384_v<VDB_F16, false>(p, n_cu, st);
    }
    if (dt == VDB_BF16) return l2 ? launch_gemm_pp_v<VDB_BF16, true>(p, n_cu, st) : launch_gemm_pp_v<VDB_BF16, false>(p, n_cu, st);
    return l2 ? launch_gemm_pp_v<VDB_F16, true>(p, n_cu, st) : launch_gemm_pp_v<VDB_F16, false>(p, n_cu, st);
}

}  // namespace vdb
