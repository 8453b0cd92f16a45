// Batched fast pass on the 16-bit matrix unit, "ping-pong" form: 256 rows x 256 queries per workgroup
// of 8 waves, operands straight from global memory into LDS (global_load_lds, no staging registers,
// no LDS stores), a ring of four 32-element k-steps, and the two waves of every SIMD half a step
// apart: while one issues its fragment reads, the loads of the k-step three ahead and (after a
// tile's last k-step) the thresholded epilogue, the other one multiplies.
//
//   scores[row, query] = X16[row, :] . Q16[query, :]        (exact products, fp32 accumulate)
//
// X16: 16-bit rows (bf16 or fp16), row stride `ldx` elements, readable for nks*32 elements per row.
// Q16: queries in the same type, row stride `qstride`, zero padded to nks*32 elements.
// L2 tables: Q16 holds -2q and the accumulators start at the row's squared norm, so that
// acc = ||x||^2 - 2 x.q (smaller is better); the key handed on is -acc, as on every other path.
//
// Each wave owns 128 rows x 64 queries: 8 x 4 blocks of v_mfma_f32_16x16x32 (SHAPE 16, the shape in
// use: higher sustained clock) or 4 x 2 blocks of v_mfma_f32_32x32x16 (SHAPE 32).  LDS stage buffer:
// A tile [256 rows][32 k] then B tile [256 queries][32 k], 64-byte rows; the 16-byte chunk c of row r
// sits in slot c ^ ((-(r >> 2)) & 3), which makes every ds_read_b128 fragment read conflict-free.
// global_load_lds writes 1 KiB per wave-instruction linearly (16 rows x 64 bytes), so the swizzle is
// applied to the per-lane source address.
//
// Synchronisation (s = stage = one k-step of one tile, flat over the workgroup's tiles):
//   LOAD(s):  [epilogue of the tile that ended at s-1]  ds_read fragments of s;  issue loads of s+3
//             into buffer (s+3)&3;  s_waitcnt vmcnt(N) leaving the loads of s+2, s+3 in flight (so
//             this wave's part of s+1 has landed);  s_waitcnt lgkmcnt(0);  s_barrier
//   COMP(s):  32 MFMAs (SHAPE 16);  s_barrier
// Waves 4-7 run one barrier behind waves 0-3.  A buffer is read in LOAD(s) only, one barrier after
// every wave waited for its loads (read after write), and refilled in LOAD(s+1) of the next round,
// at least one barrier after every wave retired its reads of it (write after read).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vdb_internal.h"

namespace vdb {

typedef float pp_f32x16 __attribute__((ext_vector_type(16)));
typedef float pp_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int pp_u32x4 __attribute__((ext_vector_type(4)));

struct GemmPPParams {
    const unsigned short* X;     // row-major rows [n][ldx], or (TILED) the tiled image: see pp_tiled_index
    const float* n2;             // [n] squared row norms (L2 only)
    RowBound     rb;             // L2: error term folded into the accumulators' start values
    const unsigned short* Q;     // tiled image of the queries: [ceil(nq/256)][nks] blocks of 256 x 32
    const float* thr;            // [nq] a candidate must have key > thr
    long long    n;
    long long    ldx;            // elements per stored row
    int          nq;
    int          qstride;
    int          nks;            // k-steps of 32 elements
    long long    ntiles;         // 256-row tiles visited: tile tile0 + t * tile_stride, t < ntiles
    long long    tile_stride;
    long long    tile0;
    float*       cand_s;         // [nq][cap]
    long long*   cand_t;
    int*         cnt;            // [nq]
    int*         lost;           // [nq]
    int          cap;
    float*       dense;          // test hook (DENSE instantiation): every key goes to dense[row * nq + query]
};

constexpr int PP_TM = 256, PP_TN = 256, PP_KS = 32;
constexpr int PP_STAGE = 32768;                    // A tile 16 KiB + B tile 16 KiB
constexpr int PP_RING = 4;
constexpr int PP_N2OFF = PP_RING * PP_STAGE;       // L2: four 1 KiB slots of row norms behind the ring
constexpr int PP_QCAP = 192;                       // candidate queue entries per wave (flushed beyond 64: room for 128)
constexpr int PP_QBYTES = 8 * PP_QCAP * 16;        // 16-byte entries, 8 waves
constexpr int PP_QOFF(bool l2) { return PP_RING * PP_STAGE + (l2 ? PP_RING * 1024 : 0); }
constexpr int PP_LDS_IP = PP_RING * PP_STAGE + PP_QBYTES;
constexpr int PP_LDS_L2 = PP_RING * PP_STAGE + PP_RING * 1024 + PP_QBYTES;

#define PP_GLOBAL __attribute__((address_space(1)))
#define PP_LOCAL __attribute__((address_space(3)))

template <int OFF = 0>
__device__ __forceinline__ void pp_glds16(const void* src, char* lds_dst) {
    // OFF is added to the global and to the LDS address
    __builtin_amdgcn_global_load_lds((const PP_GLOBAL void*)src, (PP_LOCAL void*)lds_dst, 16, OFF, 0);
}

// Tiled image of a [rows][k] 16-bit matrix: blocks of 256 rows x 32 elements, block (tile, ks) at
// element offset (tile * nks + ks) * 8192, laid out exactly as the LDS stage buffer wants it (64-byte
// rows, 16-byte chunks swizzled), so that a block moves into LDS as one linear 16 KiB copy of whole
// cache lines.  Returns the element index of (row, col).
__host__ __device__ __forceinline__ size_t pp_tiled_index(long long row, int col, int nks) {
    const long long tile = row >> 8;
    const int r = (int)(row & 255), ks = col >> 5, c = (col >> 3) & 3, e = col & 7;
    return ((size_t)tile * nks + ks) * 8192 + (size_t)r * 32 + (size_t)((c ^ ((-(r >> 2)) & 3)) << 3) + e;
}
__device__ __forceinline__ void pp_glds4(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const PP_GLOBAL void*)src, (PP_LOCAL void*)lds_dst, 4, 0, 0);
}

template <int SHAPE> struct PPAcc;
template <> struct PPAcc<32> { typedef pp_f32x16 type; };
template <> struct PPAcc<16> { typedef pp_f32x4 type; };

// SHAPE 32: v_mfma_f32_32x32x16 (two per 32-element k-step and block); SHAPE 16: v_mfma_f32_16x16x32
template <int DT, int SHAPE>
__device__ __forceinline__ typename PPAcc<SHAPE>::type pp_mfma(const pp_u32x4& a, const pp_u32x4& b,
                                                              const typename PPAcc<SHAPE>::type& c) {
    if constexpr (SHAPE == 32) {
        if constexpr (DT == 1)       // VDB_BF16
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pp_bf16x8, a), __builtin_bit_cast(pp_bf16x8, b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pp_f16x8, a), __builtin_bit_cast(pp_f16x8, b), c, 0, 0, 0);
    } else {
        if constexpr (DT == 1)
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pp_bf16x8, a), __builtin_bit_cast(pp_bf16x8, b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pp_f16x8, a), __builtin_bit_cast(pp_f16x8, b), c, 0, 0, 0);
    }
}

#ifdef PP_STAMPS
// diagnostic build only (tools/micro/gemm_pp.hip): per workgroup, the time from its start to its end in 10 ns ticks
__device__ unsigned int g_pp_end[4096];
__device__ unsigned int g_pp_clk[4096];          // shader clock cycles over the same span (clock = cycles / time)
#endif

template <int DT, bool L2M, bool DENSE, int SHAPE, bool TILED>
__global__ void __launch_bounds__(512, 1)
gemm_pp_kernel(GemmPPParams p) {
    extern __shared__ __attribute__((aligned(16))) char pp_smem[];
#ifdef PP_STAMPS
    const unsigned long long pp_stamp0 = wall_clock64();
    const unsigned long long pp_clk0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int MB = SHAPE;                      // rows (and queries) per MFMA block
    constexpr int NMI = 128 / MB, NNI = 64 / MB;   // blocks per wave tile
    constexpr int KK = SHAPE == 32 ? 2 : 1;        // MFMAs per block and stage
    constexpr int AR = SHAPE == 32 ? 16 : 4;       // accumulator registers per block
    typedef typename PPAcc<SHAPE>::type accv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;       // wave tile: rows wm*128.., queries wn*64..
    const int li = lane & (MB - 1), lh = lane / MB;
    const int qt = blockIdx.y;
    const int nks = p.nks;

    // fragment read offsets inside a stage buffer (k16 block kk, row block mi at + mi * 2048)
    const int fsw = (-(li >> 2)) & 3;
    int a_off[KK], b_off[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        a_off[kk] = (wm * 128 + li) * 64 + (((kk * (64 / MB) + lh) ^ fsw) << 4);
        b_off[kk] = 16384 + (wn * 64 + li) * 64 + (((kk * (64 / MB) + lh) ^ fsw) << 4);
    }

    // load map: piece pa = wave*2 + i covers tile rows pa*16 .. +15; lane -> row pa*16 + (lane>>2),
    // LDS slot lane&3, which holds source chunk (lane&3) ^ swizzle(row)
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const unsigned voff = (unsigned)(wave * 2048 + lane * 16);     // this lane's bytes of a tiled 16 KiB block
    const char* const btile = (const char*)p.Q + (((size_t)qt * nks) << 14);
    float thr[NNI];
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) {
        const int q = qt * PP_TN + wn * 64 + ni * MB + li;
        const float t = (q < p.nq) ? p.thr[q] : __builtin_huge_valf();
        thr[ni] = L2M ? -t : t;                    // L2: candidates have acc < -thr
    }
    // L2: the rows' error term (see RowBound); two wave-uniform values, fetched before the loop as well
    bool rbon = false;
    float ea = 0.f, es = 0.f;
    if constexpr (L2M) {
        rbon = p.rb.qinfo != nullptr;
        if (rbon) { ea = p.rb.a; es = rowbound_s(p.rb, lane); }
        asm volatile("" : "+v"(ea), "+v"(es));
    }
    // the thresholds are waited for here, ahead of every global_load_lds: a wait at their first use
    // inside the loop would have to be vmcnt(0) and drain the prefetch once per tile
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) asm volatile("" : "+v"(thr[ni]));

    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long total = my_tiles * nks;

    // fetch cursor
    long long f_seq = 0;                           // tiles of this workgroup fetched so far
    int f_ks = 0;
    long long f_left = total;
    const unsigned short* asrc[2] = {nullptr, nullptr};
    const char* atile = nullptr;
    const float* nsrc = nullptr;
    auto set_tile_src = [&]() __attribute__((always_inline)) {
        const long long tile = p.tile0 + (blockIdx.x + f_seq * gridDim.x) * p.tile_stride;
        const long long row0 = tile * PP_TM;
        if constexpr (TILED) {
            atile = (const char*)p.X + (((size_t)tile * nks) << 14);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long long r = row0 + (wave * 2 + i) * 16 + lrow;
                if (r > p.n - 1) r = p.n - 1;
                asrc[i] = p.X + (size_t)r * p.ldx + lchunk * 8;
            }
        }
        if constexpr (L2M) {
            long long r = row0 + (wave & 3) * 64 + lane;
            if (r > p.n - 1) r = p.n - 1;
            nsrc = p.n2 + r;
        }
    };
    set_tile_src();
    // Every call issues the same number of operand loads (stages past the end re-read the last one's
    // operands into a buffer nobody reads), so that the counted waits below hold on every path.
    auto issue_a = [&](int buf) __attribute__((always_inline)) {
        char* dst = pp_smem + buf * PP_STAGE + wave * 2048;
        if constexpr (TILED) {
            const char* a = atile + ((size_t)f_ks << 14) + voff;
            pp_glds16<0>(a, dst);
            pp_glds16<1024>(a, dst);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) pp_glds16(asrc[i] + f_ks * PP_KS, dst + i * 1024);
        }
        // L2: the rows' norms travel with the first k-step of their tile only (one more load in one stage of
        // nks: the counted waits below allow for none, which is the stricter side)
        if constexpr (L2M) { if (f_ks == 0) pp_glds4(nsrc, pp_smem + PP_N2OFF + buf * 1024 + (wave & 3) * 256); }
    };
    auto issue_b = [&](int buf) __attribute__((always_inline)) {
        char* dst = pp_smem + buf * PP_STAGE + wave * 2048;
        const char* b = btile + ((size_t)f_ks << 14) + voff;
        pp_glds16<0>(b, dst + 16384);
        pp_glds16<1024>(b, dst + 16384);
        if (--f_left > 0) {
            if (++f_ks == nks) { f_ks = 0; ++f_seq; set_tile_src(); }
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) { issue_a(buf); issue_b(buf); };
    constexpr int NA = 2, NB = 2;
    constexpr int WAITN = 2 * (NA + NB);
    issue(0); issue(1); issue(2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * (NA + NB)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();     // waves 4-7 run one barrier behind

    accv acc[NMI][NNI];
    long long c_seq = 0;
    int c_ks = 0;
    bool epi = false;
    long long epi_row0 = 0;

    // Candidates go to a queue of PP_QCAP entries {key, row, query} per wave in LDS and leave it in
    // batches: the atomic that reserves a slot in a query's candidate array returns a value, and
    // waiting for it drains every load in flight, which is affordable once per batch only.
    char* const queue = pp_smem + PP_QOFF(L2M) + wave * (PP_QCAP * 16);
    int qn = 0;                                    // entries queued (wave-uniform)
    auto flush = [&]() __attribute__((always_inline)) {
        for (int i0 = 0; i0 < qn; i0 += 64) {
            const int i = i0 + lane;
            if (i < qn) {
                const pp_u32x4 e = *(const pp_u32x4*)(queue + i * 16);
                const int q = (int)e[3];
                const int pos = atomicAdd(&p.cnt[q], 1);
                if (pos < p.cap) {
                    p.cand_s[(size_t)q * p.cap + pos] = __uint_as_float(e[0]);
                    p.cand_t[(size_t)q * p.cap + pos] = (long long)(((unsigned long long)e[2] << 32) | e[1]);
                } else {
                    p.lost[q] = 1;
                }
            }
        }
        qn = 0;
    };

    auto epilogue = [&](long long row0) __attribute__((always_inline)) {
        const long long left = p.n - row0;
        if (left <= 0) return;
        const int nloc = left < PP_TM ? (int)left : PP_TM;
        const int q0 = qt * PP_TN + wn * 64 + li;
        if (!DENSE && qn > 64) flush();            // room for 128 or more entries in every tile
        if constexpr (DENSE) {
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
                    for (int r = 0; r < AR; ++r) {
                        const int lr = wm * 128 + mi * MB + 4 * lh + (r & 3) + 8 * (r >> 2);
                        const int q = q0 + ni * MB;
                        if (lr < nloc && q < p.nq) p.dense[(size_t)(row0 + lr) * p.nq + q] = L2M ? -acc[mi][ni][r] : acc[mi][ni][r];
                    }
            return;
        }
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) {
            // one compare for all scores of a lane and query first: candidates are rare
            float best = L2M ? __builtin_huge_valf() : -__builtin_huge_valf();
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int r = 0; r < AR; ++r) best = L2M ? fminf(best, acc[mi][ni][r]) : fmaxf(best, acc[mi][ni][r]);
            const bool any = L2M ? best < thr[ni] : best > thr[ni];
            if (__builtin_expect(__ballot(any) != 0ull, 0)) {
                // Everything this rare path needs is derived here from two opaque registers: values the
                // compiler could compute ahead of the loop would be spilled, and a reload from scratch waits
                // with vmcnt(0), which drains the loads in flight
                int lb = wm * 128 + 4 * lh, qq = q0 + ni * MB;
                asm volatile("" : "+v"(lb), "+v"(qq));
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) {
                    float bm = acc[mi][ni][0];
#pragma unroll
                    for (int r = 1; r < AR; ++r) bm = L2M ? fminf(bm, acc[mi][ni][r]) : fmaxf(bm, acc[mi][ni][r]);
                    if (__ballot(L2M ? bm < thr[ni] : bm > thr[ni]) == 0ull) continue;
#pragma unroll
                    for (int r = 0; r < AR; ++r) {
                        const float v = acc[mi][ni][r];
                        const int lr = lb + (mi * MB + (r & 3) + 8 * (r >> 2));
                        const bool hit = (L2M ? v < thr[ni] : v > thr[ni]) && lr < nloc;
                        const unsigned long long mask = __ballot(hit);
                        if (mask == 0ull) continue;
                        // the wave's own queue in LDS: no atomic, nothing to wait for
                        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (hit) {
                            if (slot < PP_QCAP) {
                                const unsigned long long row = (unsigned long long)row0 + (unsigned)lr;
                                pp_u32x4 e;
                                e[0] = __float_as_uint(L2M ? -v : v);
                                e[1] = (unsigned)row;
                                e[2] = (unsigned)(row >> 32);
                                e[3] = (unsigned)qq;
                                *(pp_u32x4*)(queue + slot * 16) = e;
                            } else {
                                p.lost[qq] = 1;        // more than 128 hits in one 128 x 64 tile: the exact pass answers
                            }
                        }
                        qn += __builtin_popcountll(mask);
                        if (qn > PP_QCAP) qn = PP_QCAP;
                    }
                }
            }
        }
    };

    pp_u32x4 af[NMI][KK], bf[NNI][KK];
    for (long long s = 0; s < total; ++s) {
        const int buf = (int)(s & 3);
        const char* sb = pp_smem + buf * PP_STAGE;
        // ---------------- LOAD(s)
        if (epi) { epilogue(epi_row0); epi = false; }
#if defined(PP_ABL) && (PP_ABL & 2)
        if (s == 0)                    // timing ablation: fragments read once
#endif
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int ni = 0; ni < NNI; ++ni) bf[ni][kk] = *(const pp_u32x4*)(sb + b_off[kk] + ni * (MB * 64));
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) af[mi][kk] = *(const pp_u32x4*)(sb + a_off[kk] + mi * (MB * 64));
        }
        if constexpr (L2M) {
            if (c_ks == 0) {
                const char* nb = pp_smem + PP_N2OFF + buf * 1024 + (wm * 128 + 4 * lh) * 4;
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                    for (int j = 0; j < AR / 4; ++j) {
                        pp_f32x4 v = *(const pp_f32x4*)(nb + (mi * MB + 8 * j) * 4);
                        if (rbon) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = rowbound_adj(v[e], ea, es);
                        }
#pragma unroll
                        for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mi][ni][4 * j + e] = v[e];
                    }
            }
        }
#if !defined(PP_ABL) || !(PP_ABL & 1)
        issue((int)((s + 3) & 3));
#endif
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(WAITN) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMP(s)
        __builtin_amdgcn_s_setprio(1);
#if defined(PP_ABL) && (PP_ABL & 4)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {          // timing ablation: no MFMAs, operands kept alive
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) asm volatile("" :: "v"(af[mi][kk]));
#pragma unroll
            for (int ni = 0; ni < NNI; ++ni) asm volatile("" :: "v"(bf[ni][kk]));
        }
        if (s == 0) {
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
                    for (int r = 0; r < AR; ++r) acc[mi][ni][r] = 0.f;
        }
#else
        if (!L2M && c_ks == 0) {
            accv z;
#pragma unroll
            for (int r = 0; r < AR; ++r) z[r] = 0.f;
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni) acc[mi][ni] = pp_mfma<DT, SHAPE>(af[mi][0], bf[ni][0], z);
        } else {
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni) acc[mi][ni] = pp_mfma<DT, SHAPE>(af[mi][0], bf[ni][0], acc[mi][ni]);
        }
#pragma unroll
        for (int kk = 1; kk < KK; ++kk)
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni) acc[mi][ni] = pp_mfma<DT, SHAPE>(af[mi][kk], bf[ni][kk], acc[mi][ni]);
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (++c_ks == nks) {
            c_ks = 0;
            epi = true;
            epi_row0 = (p.tile0 + (blockIdx.x + c_seq * gridDim.x) * p.tile_stride) * PP_TM;
            ++c_seq;
        }
    }
    if (epi) epilogue(epi_row0);
    if (!DENSE) flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wm == 0) __builtin_amdgcn_s_barrier();     // matches the extra barrier of waves 4-7
#ifdef PP_STAMPS
    if (tid == 0) {
        g_pp_clk[(blockIdx.y * gridDim.x + blockIdx.x) & 4095] = (unsigned)(__builtin_amdgcn_s_memtime() - pp_clk0);
        g_pp_end[(blockIdx.y * gridDim.x + blockIdx.x) & 4095] = (unsigned)(wall_clock64() - pp_stamp0);
    }
#endif
}


}  // namespace vdb
