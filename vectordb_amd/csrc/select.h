// Wave-level partial top-K selection in LDS.
//
// Each wavefront owns a candidate list of C (key, tag) pairs in LDS and keeps the K best of the
// stream it has seen.  New candidates are appended behind a threshold test; when the list is
// full the wave sorts it (bitonic network over LDS, all 64 lanes) and keeps the first K.  After
// the warm-up the threshold rejects almost every row with one compare, so the list costs
// nothing on the streaming path.  "Better" = larger key, ties by smaller tag.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdb {

constexpr long long TAG_NONE = 0x7fffffffffffffffLL;

__device__ __forceinline__ float neg_inf() { return -__builtin_huge_valf(); }

__device__ __forceinline__ bool better(float s, long long t, float rs, long long rt) {
    return (s > rs) || (s == rs && t < rt);
}

// Orders LDS accesses of the 64 lanes of one wave (writes by some lanes, reads by others).
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Sort n (power of two) pairs in LDS, best first, with one wave.
__device__ inline void wave_bitonic_sort(float* s, long long* t, int n, int lane) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = lane; p < (n >> 1); p += 64) {
                const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                const int j = i + stride;
                const float si = s[i], sj = s[j];
                const long long ti = t[i], tj = t[j];
                const bool up = (i & size) == 0;           // this pair wants the better one first
                const bool j_better = better(sj, tj, si, ti);
                const bool i_better = better(si, ti, sj, tj);
                if (up ? j_better : i_better) {
                    s[i] = sj; t[i] = tj;
                    s[j] = si; t[j] = ti;
                }
            }
            lds_wave_sync();
        }
    }
}

struct WaveSel {
    float*     s;
    long long* t;
    int        C, K;
    int        cnt;       // wave-uniform number of live entries
    float      thr_s;     // a candidate must be better than (thr_s, thr_t)
    long long  thr_t;
    float      floor_s;   // keys below this are known to be outside the global K best (-inf: unknown)

    // `floor` is a lower bound of the K-th best key of the whole stream (of all waves): keys below
    // it are rejected from the first row on; keys equal to it still pass.
    __device__ void init(float* s_, long long* t_, int C_, int K_, int lane, float floor = neg_inf()) {
        s = s_; t = t_; C = C_; K = K_;
        for (int i = lane; i < C; i += 64) { s[i] = neg_inf(); t[i] = TAG_NONE; }
        cnt = 0;                       // until K entries have come in, everything at or above the floor passes
        floor_s = floor;
        thr_s = floor; thr_t = TAG_NONE;
        lds_wave_sync();
    }

    // Sort the live entries, keep the K best, refresh the threshold.  Called by the whole wave.
    // Only the smallest power of two that holds the live entries is sorted (a wave that a floor lets
    // end with a handful of entries sorts 64 slots, not C); slots behind the live entries hold
    // (-inf, TAG_NONE) afterwards, up to K at least: the first K slots are always a valid sorted list.
    __device__ void compact(int lane) {
        lds_wave_sync();
        int P = 64;
        while (P < cnt) P <<= 1;                                   // cnt <= C, a power of two >= 128
        for (int i = cnt + lane; i < P; i += 64) { s[i] = neg_inf(); t[i] = TAG_NONE; }
        lds_wave_sync();
        wave_bitonic_sort(s, t, P, lane);
        if (cnt >= K) {
            cnt = K;
            thr_s = s[K - 1];
            thr_t = t[K - 1];
            if (floor_s > thr_s) { thr_s = floor_s; thr_t = TAG_NONE; }
        }
    }

    // Called by the whole wave; lanes with pass == true contribute (key, tag).
    __device__ __forceinline__ void push(bool pass, float key, long long tag, int lane) {
        const unsigned long long m = __ballot(pass);
        if (m == 0ull) return;
        const int n = __popcll(m);
        if (cnt + n > C) compact(lane);        // afterwards cnt <= K and K + 64 <= C
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (pass) { s[pos] = key; t[pos] = tag; }
        cnt += n;
    }
};

// After every wave of the block has finished its stream: merge the W lists (laid out back to
// back, C entries each) into wave 0's list, sorted best first.  Called by the whole block.
__device__ inline void block_merge(WaveSel& sel, float* S_all, long long* T_all, int W,
                                   int wave, int lane) {
    sel.compact(lane);
    const int C = sel.C, K = sel.K;
    for (int stride = 1; stride < W; stride <<= 1) {
        __syncthreads();
        if ((wave & (2 * stride - 1)) == 0 && wave + stride < W) {
            const float*     ps = S_all + (size_t)(wave + stride) * C;
            const long long* pt = T_all + (size_t)(wave + stride) * C;
            for (int i = lane; i < K; i += 64) { sel.s[K + i] = ps[i]; sel.t[K + i] = pt[i]; }
            sel.cnt = 2 * K;
            sel.compact(lane);
        }
    }
    __syncthreads();
}

// Block-wide top-K of `count` (key, tag) pairs held in global memory; result in S_all/T_all[0..K).
__device__ inline void block_topk_global(const float* gs, const long long* gt, long long count,
                                         float* S_all, long long* T_all, int C, int K, int W,
                                         int wave, int lane) {
    WaveSel sel;
    sel.init(S_all + (size_t)wave * C, T_all + (size_t)wave * C, C, K, lane);
    for (long long base = (long long)wave * 64; base < count; base += (long long)W * 64) {
        const long long i = base + lane;
        float key = neg_inf();
        long long tag = TAG_NONE;
        if (i < count) { key = gs[i]; tag = gt[i]; }
        const bool pass = (tag != TAG_NONE) && better(key, tag, sel.thr_s, sel.thr_t);
        sel.push(pass, key, tag, lane);
    }
    block_merge(sel, S_all, T_all, W, wave, lane);
}

}  // namespace vdb
