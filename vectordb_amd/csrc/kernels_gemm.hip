// Batched-query path: scores[rows, queries] = X . Q^T as an fp32 MFMA GEMM whose epilogue keeps
// only the scores that beat a per-query threshold.  The score matrix never reaches HBM.
//
// Selection without in-kernel lists: a threshold tau_q that is known to be at or below the K-th
// best score of query q lets the epilogue drop almost every score with one compare and append the
// few survivors to a per-query candidate array with one global atomic.  tau_q comes from the same
// kernel run on a small strided sample of the row tiles (whose own threshold comes from an exact
// top-K over a few thousand rows), so three passes of geometrically growing size end with about
// two thousand candidates per query out of millions of rows.  If a candidate array overflows
// (rows ordered adversarially), the query is flagged and answered by the exact pass instead.
//
// Tile: 128 rows x 128 queries x 32 k per workgroup of four waves (2 x 2); each wave owns
// 64 x 64 as 2 x 2 MFMA tiles of v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate in
// k order).  Operand tiles are staged through registers into LDS rows padded by 16 bytes
// (conflict-free 16-byte fragment reads); the loads of the next k-step are issued before the
// MFMAs of the current one.  Workgroups that share a row range (one per query tile) are placed
// on one XCD so that the row tile is fetched from HBM once and served from that XCD's L2.
#include "vdb_internal.h"
#include "select.h"

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
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const f32x4*)p; }

__global__ void __launch_bounds__(256, 2)
gemm_thresh_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[GB_M * GB_LD];
    __shared__ __attribute__((aligned(16))) float Bs[GB_N * GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.y;
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
                rb[i] = ldg4(qptr[i] + c);
                if (!ok) { ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; rb[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
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
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = qt * GB_N + wn * 64 + ni * 32 + li;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = row0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float key = acc[mi][ni][r];
                    if (p.l2) key = 2.f * key - p.n2[row < p.n ? row : p.n - 1];
                    if (key > thr[ni] && row < p.n) {
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
                              float* cand_s, long long* cand_t, int* cnt, int* lost, int cap,
                              int n_cu, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    GemmParams p;
    p.X = (const float*)t.rows; p.n2 = t.n2; p.Q = qprep; p.thr = thr; p.n = t.n; p.nq = nq;
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
    hipLaunchKernelGGL(gemm_thresh_kernel, dim3((unsigned)gx, (unsigned)qtiles), dim3(256), 0, st, p);
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
                     float* __restrict__ thr_out) {
    __shared__ int hist[256];
    __shared__ int ctl[8];
    const int q = blockIdx.x;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    long long count = cnt ? (long long)cnt[q] : count_all;
    if (cnt && count > cap) count = cap;
    const float* gs = cand_s + (size_t)q * stride;
    const float tin = thr_in ? thr_in[q] : -__builtin_huge_valf();
    uint32_t prefix = 0;
    int need = K;
    bool short_list = false;
    for (int pass = 0; pass < 4; ++pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = 24 - 8 * pass;
        for (long long i0 = 0; i0 < count; i0 += nth) {
            const long long i = i0 + tid;
            const float v = (i < count) ? gs[i] : -__builtin_huge_valf();
            const uint32_t key = fkey_g(v);
            const bool act = (v > -__builtin_huge_valf()) && (pass == 0 || (key >> (shift + 8)) == prefix);
            // aggregate equal bins of a wave into one LDS atomic
            unsigned long long todo = __ballot(act);
            const int bin = (int)((key >> shift) & 255u);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int b = __shfl(bin, leader);
                const unsigned long long same = __ballot(act && bin == b) & todo;
                if (lane == leader) atomicAdd(&hist[b], __popcll(same));
                todo &= ~same;
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
                                int cap, int K, const float* thr_in, float* thr_out, int nq, hipStream_t st) {
    hipLaunchKernelGGL(kth_threshold_kernel, dim3((unsigned)nq), dim3(256), 0, st,
                       cand_s, stride, cnt, count_all, cap, K, thr_in, thr_out);
    return hipGetLastError();
}

}  // namespace vdb
