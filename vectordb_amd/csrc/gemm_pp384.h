// Batched fast pass on the 16-bit matrix unit, wide form of the ping-pong kernel (gemm_pp.h):
// 384 rows x 256 queries per workgroup of 8 waves.  The kernel is bound by the rate at which a CU
// moves operands from L2 into LDS; a 384 x 256 tile needs 40 KiB per k-step of 32 elements for
// 6.3 MFLOP where the 256 x 256 tile needs 32 KiB for 4.2 MFLOP (17 % fewer bytes per flop).
//
//   scores[row, query] = X16[row, :] . Q16[query, :]        (exact products, fp32 accumulate)
//
// Operands as in gemm_pp.h: X16 row-major 16-bit rows (stride ldx elements, readable for nks*32
// elements per row), Q16 the tiled image of the queries (blocks of 256 queries x 32 elements in LDS
// order, pp_tiled_index), L2 tables with -2q and accumulators that start at ||x||^2 - e(x).
//
// Each wave owns 96 rows x 128 queries = 6 x 8 blocks of v_mfma_f32_16x16x32: 192 accumulator
// registers of the 256 a wave has at two waves per SIMD.  The fragments of a whole k-step (6 row
// blocks, 8 query blocks: 56 registers) do not fit beside them, so a k-step is two half-steps over
// 64 queries each: the row fragments (24 registers) stay, the query fragments (16) are read twice.
//
// LDS: ring of three stages of 40 KiB (A tile [384 rows][32 k] then B tile [256 queries][32 k],
// 64-byte rows, the 16-byte chunk c of row r in slot c ^ ((-(r >> 2)) & 3)), three slots of row
// norms (L2), the thresholds of the 256 queries, the candidate queues of the eight waves.
//
// Synchronisation (s = stage = one k-step of one tile, flat over the workgroup's tiles; waves 4-7 run
// one barrier behind waves 0-3, so that the two waves of a SIMD alternate between loading and
// multiplying):
//   LOAD(s,0): [epilogue of the tile that ended at s-1]  ds_read row fragments and query fragments
//              0-3 of s;  issue the row loads of s+2 into buffer (s+2) % 3;  lgkmcnt(0);  s_barrier
//   COMP(s,0): 24 MFMAs;  s_barrier
//   LOAD(s,1): ds_read query fragments 4-7 of s;  issue the query loads of s+2;  s_waitcnt vmcnt(5)
//              (the loads of s+2 stay in flight: this wave's part of s+1 has landed);  lgkmcnt(0);
//              s_barrier
//   COMP(s,1): 24 MFMAs;  s_barrier
// A buffer is read in LOAD phases only.  Stage s+1 is read from LOAD(s+1,0) on, two barriers after
// every wave waited for its loads of it (one would do for the waves' own group; the other group is a
// barrier behind).  Buffer s % 3 is refilled from LOAD(s+1,0) on, after the barrier that ends the
// slot in which the trailing group made its last reads of it (LOAD(s,1)).
#pragma once
#include "gemm_pp.h"

namespace vdb {

constexpr int PW_TM = 384, PW_TN = 256;
constexpr int PW_ASTAGE = PW_TM * 64;                   // 24 KiB of rows per stage
constexpr int PW_BSTAGE = PW_TN * 64;                   // 16 KiB of queries
constexpr int PW_STAGE = PW_ASTAGE + PW_BSTAGE;
constexpr int PW_RING = 3;
constexpr int PW_N2OFF = PW_RING * PW_STAGE;            // three slots of 384 row norms
constexpr int PW_THROFF = PW_N2OFF + PW_RING * PW_TM * 4;
constexpr int PW_QOFF = PW_THROFF + PW_TN * 4;
constexpr int PW_QCAP = 192;                            // candidate queue entries per wave (flushed beyond 64)
constexpr int PW_LDS = PW_QOFF + 8 * PW_QCAP * 16;      // 153088 bytes

// The lane number, computed where it is used: a value derived from threadIdx ahead of the main loop
// would be kept in a register (or spilled) across it, and the kernel has none to spare.
__device__ __forceinline__ int pw_lane_here() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Tiled image of a [rows][k] 16-bit matrix for this kernel: blocks of 384 rows x 32 elements, block
// (tile, ks) at element offset (tile * nks + ks) * 12288, in the LDS stage layout.
__host__ __device__ __forceinline__ size_t pw_tiled_index(long long row, int col, int nks) {
    const long long tile = row / PW_TM;
    const int r = (int)(row % PW_TM), ks = col >> 5, c = (col >> 3) & 3, e = col & 7;
    return ((size_t)tile * nks + ks) * (PW_TM * 32) + (size_t)r * 32 + (size_t)((c ^ ((-(r >> 2)) & 3)) << 3) + e;
}

// VAR bit 0: row loads non-temporal; bit 1: query loads in LOAD(s,0), row loads in LOAD(s,1)
template <int DT, bool L2M, bool DENSE, bool TILED = false, int VAR = 0>
__global__ void __launch_bounds__(512, 1)
gemm_pp384_kernel(GemmPPParams p) {
    extern __shared__ __attribute__((aligned(16))) char pw_smem[];
#ifdef PP_STAMPS
    const unsigned long long pp_stamp0 = wall_clock64();
    const unsigned long long pp_clk0 = __builtin_amdgcn_s_memtime();
#endif
    typedef pp_f32x4 accv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wave tile: rows wm*96.., queries wn*128..
    const int grp = wave >> 2;                     // waves 4-7 trail by one barrier
    const int li = lane & 15, lh = lane >> 4;
    const int qt = blockIdx.y;
    const int nks = p.nks;

    // thresholds of this workgroup's queries, in LDS (read by the epilogue only); L2: candidates have acc < -thr
    if (tid < PW_TN) {
        const int q = qt * PW_TN + tid;
        const float t = (q < p.nq) ? p.thr[q] : __builtin_huge_valf();
        ((float*)(pw_smem + PW_THROFF))[tid] = L2M ? -t : t;
    }
    // L2: the rows' error term (see RowBound); two wave-uniform values
    bool rbon = false;
    float ea = 0.f, es = 0.f;
    if constexpr (L2M) {
        rbon = p.rb.qinfo != nullptr;
        if (rbon) {
            ea = p.rb.a;
            es = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rowbound_s(p.rb, lane))));
        }
        asm volatile("" : "+s"(ea), "+s"(es));
    }
    // ordinary loads and LDS stores are done with before the first global_load_lds: a wait for them
    // inside the loop would have to be vmcnt(0) and drain the prefetch
    __syncthreads();

    // fragment read offsets inside a stage buffer (row block mi at + mi * 1024, query block ni at + ni * 1024)
    const int fsw = (-(li >> 2)) & 3;
    const int a_off = (wm * 96 + li) * 64 + ((lh ^ fsw) << 4);
    const int b_off = PW_ASTAGE + (wn * 128 + li) * 64 + ((lh ^ fsw) << 4);

    // load map, rows: 24 pieces of 16 rows x 64 bytes per stage, piece 3 * wave + i; lane -> row
    // 48 * wave + 16 * i + (lane >> 2), LDS slot lane & 3, which holds source chunk (lane & 3) ^ swizzle(row)
    const unsigned qvoff = (unsigned)(wave * 2048 + lane * 16);    // this lane's bytes of a tiled 16 KiB query block
    const char* const btile = (const char*)p.Q + (((size_t)qt * nks) << 14);
    const unsigned ldx2 = (unsigned)p.ldx * 2u;

    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long total = my_tiles * nks;

    // fetch cursor
    long long f_seq = 0;                           // tiles of this workgroup fetched so far
    int f_ks = 0;
    long long f_left = total;
    const char* abase = nullptr;                   // first row of the tile being fetched (wave-uniform)
    unsigned avoff[3] = {0u, 0u, 0u};              // this lane's byte offsets from abase, one per piece
    const char* nbase = nullptr;
    unsigned nvoff = 0u;
    auto set_tile_src = [&]() __attribute__((always_inline)) {
        const long long tile = p.tile0 + (blockIdx.x + f_seq * gridDim.x) * p.tile_stride;
        const long long row0 = tile * PW_TM;
        const long long left = p.n - row0;         // >= 1: the host visits no tile past the table
        const int last = left < PW_TM ? (int)left - 1 : PW_TM - 1;
        const int ln = pw_lane_here();
        if constexpr (TILED) {
            abase = (const char*)p.X + (size_t)tile * nks * PW_ASTAGE;
        } else {
            abase = (const char*)p.X + (size_t)row0 * ldx2;
            const int lrow = ln >> 2;
            const int lchunk = (ln & 3) ^ ((-(ln >> 4)) & 3);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int r = wave * 48 + i * 16 + lrow;
                if (r > last) r = last;            // rows past the table re-read its last row (never written out)
                avoff[i] = (unsigned)r * ldx2 + (unsigned)(lchunk * 16);
            }
        }
        if constexpr (L2M) {
            int r = (wave % 6) * 64 + ln;
            if (r > last) r = last;
            nbase = (const char*)(p.n2 + row0);
            nvoff = (unsigned)r * 4u;
        }
    };
    set_tile_src();
    // Every call issues the same number of operand loads (stages past the end re-read the last one's
    // operands into a buffer nobody reads), so that the counted wait below holds on every path.
    auto issue_a = [&](int buf) __attribute__((always_inline)) {
        char* dst = pw_smem + buf * PW_STAGE + wave * 3072;
        constexpr int AUX = (VAR & 1) ? 2 : 0;
        if constexpr (TILED) {
            const char* a = abase + (size_t)f_ks * PW_ASTAGE + wave * 3072 + qvoff - wave * 2048;
            __builtin_amdgcn_global_load_lds((const PP_GLOBAL void*)a, (PP_LOCAL void*)dst, 16, 0, AUX);
            __builtin_amdgcn_global_load_lds((const PP_GLOBAL void*)a, (PP_LOCAL void*)dst, 16, 1024, AUX);
            __builtin_amdgcn_global_load_lds((const PP_GLOBAL void*)a, (PP_LOCAL void*)dst, 16, 2048, AUX);
        } else {
            const char* a = abase + f_ks * 64;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_global_load_lds((const PP_GLOBAL void*)(a + avoff[i]), (PP_LOCAL void*)(dst + i * 1024), 16, 0, AUX);
        }
        // L2: the rows' norms travel with the first k-step of their tile (one more load in one stage of
        // nks, in six of the eight waves: the counted wait allows for none, which is the stricter side)
        if constexpr (L2M) {
            if (f_ks == 0 && wave < 6) pp_glds4(nbase + nvoff, pw_smem + PW_N2OFF + buf * (PW_TM * 4) + wave * 256);
        }
    };
    auto issue_b = [&](int buf) __attribute__((always_inline)) {
        char* dst = pw_smem + buf * PW_STAGE + PW_ASTAGE + wave * 2048;
        const char* b = btile + ((size_t)f_ks << 14) + qvoff;
        pp_glds16<0>(b, dst);
        pp_glds16<1024>(b, dst);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (--f_left > 0) {
            if (++f_ks == nks) { f_ks = 0; ++f_seq; set_tile_src(); }
        }
    };
    constexpr bool QFIRST = (VAR & 2) != 0;
    issue_a(0); issue_b(0); advance();
    issue_a(1); issue_b(1); advance();
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();    // waves 4-7 run one barrier behind

    accv acc[6][8];
    long long c_seq = 0;
    int c_ks = 0;
    bool epi = false;
    long long epi_row0 = 0;

    // Candidates go to a queue of PW_QCAP entries {key, row, query} per wave in LDS and leave it in
    // batches: the atomic that reserves a slot in a query's candidate array returns a value, and
    // waiting for it drains every load in flight, which is affordable once per batch only.
    char* const queue = pw_smem + PW_QOFF + wave * (PW_QCAP * 16);
    int qn = 0;                                    // entries queued (wave-uniform)
    auto flush = [&]() __attribute__((always_inline)) {
        const int fl = pw_lane_here();
        for (int i0 = 0; i0 < qn; i0 += 64) {
            const int i = i0 + fl;
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
        const int nloc = left < PW_TM ? (int)left : PW_TM;
        const int eln = pw_lane_here();
        const int eli = eln & 15, elh = eln >> 4;
        const int q0 = qt * PW_TN + wn * 128 + eli;
        if (!DENSE && qn > 64) flush();            // room for 128 or more entries in every tile
        if constexpr (DENSE) {
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = wm * 96 + mi * 16 + 4 * elh + r;
                        const int q = q0 + ni * 16;
                        if (lr < nloc && q < p.nq) p.dense[(size_t)(row0 + lr) * p.nq + q] = L2M ? -acc[mi][ni][r] : acc[mi][ni][r];
                    }
            return;
        }
        const float* const thr_l = (const float*)(pw_smem + PW_THROFF) + wn * 128 + eli;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const float th = thr_l[ni * 16];
            // one compare for all scores of a lane and query first: candidates are rare
            float best = L2M ? __builtin_huge_valf() : -__builtin_huge_valf();
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) best = L2M ? fminf(best, acc[mi][ni][r]) : fmaxf(best, acc[mi][ni][r]);
            const bool any = L2M ? best < th : best > th;
            if (__builtin_expect(__ballot(any) != 0ull, 0)) {
                // Everything this rare path needs is derived from the lane number taken above: values the
                // compiler could compute ahead of the main loop would be spilled, and a reload from scratch
                // waits with vmcnt(0), which drains the loads in flight
                int lb = wm * 96 + 4 * elh, qq = q0 + ni * 16;
                asm volatile("" : "+v"(lb), "+v"(qq));
#pragma unroll
                for (int mi = 0; mi < 6; ++mi) {
                    float bm = acc[mi][ni][0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) bm = L2M ? fminf(bm, acc[mi][ni][r]) : fmaxf(bm, acc[mi][ni][r]);
                    if (__ballot(L2M ? bm < th : bm > th) == 0ull) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[mi][ni][r];
                        const int lr = lb + (mi * 16 + r);
                        const bool hit = (L2M ? v < th : v > th) && lr < nloc;
                        const unsigned long long mask = __ballot(hit);
                        if (mask == 0ull) continue;
                        // the wave's own queue in LDS: no atomic, nothing to wait for
                        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (hit) {
                            if (slot < PW_QCAP) {
                                const unsigned long long row = (unsigned long long)row0 + (unsigned)lr;
                                pp_u32x4 e;
                                e[0] = __float_as_uint(L2M ? -v : v);
                                e[1] = (unsigned)row;
                                e[2] = (unsigned)(row >> 32);
                                e[3] = (unsigned)qq;
                                *(pp_u32x4*)(queue + slot * 16) = e;
                            } else {
                                p.lost[qq] = 1;        // more than 128 hits in one 96 x 128 tile: the exact pass answers
                            }
                        }
                        qn += __builtin_popcountll(mask);
                        if (qn > PW_QCAP) qn = PW_QCAP;
                    }
                }
            }
        }
    };

    pp_u32x4 af[6], bf[4];
    int buf = 0;                                   // s % 3
    for (long long s = 0; s < total; ++s) {
        const char* sb = pw_smem + buf * PW_STAGE;
        const int nbuf = buf == 0 ? 2 : buf - 1;   // (s + 2) % 3
        // ---------------- LOAD(s,0)
        if (epi) { epilogue(epi_row0); epi = false; }
#if defined(PP_ABL) && (PP_ABL & 2)
        if (s == 0) {                  // timing ablation: fragments read once
#else
        {
#endif
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = *(const pp_u32x4*)(sb + b_off + ni * 1024);
#pragma unroll
            for (int mi = 0; mi < 6; ++mi) af[mi] = *(const pp_u32x4*)(sb + a_off + mi * 1024);
        }
        if constexpr (L2M) {
            if (c_ks == 0) {
                const char* nb = pw_smem + PW_N2OFF + buf * (PW_TM * 4) + (wm * 96 + 4 * (pw_lane_here() >> 4)) * 4;
#pragma unroll
                for (int mi = 0; mi < 6; ++mi) {
                    pp_f32x4 v = *(const pp_f32x4*)(nb + mi * 64);
                    if (rbon) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rowbound_adj(v[e], ea, es);
                    }
#pragma unroll
                    for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = v;
                }
            }
        }
#if !defined(PP_ABL) || !(PP_ABL & 1)
        if constexpr (QFIRST) issue_b(nbuf); else issue_a(nbuf);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMP(s,0)
        __builtin_amdgcn_s_setprio(1);
#if defined(PP_ABL) && (PP_ABL & 4)
#pragma unroll
        for (int mi = 0; mi < 6; ++mi) asm volatile("" :: "v"(af[mi]));      // timing ablation: no MFMAs, operands kept alive
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" :: "v"(bf[ni]));
        if (s == 0) {
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
        }
#else
        if (!L2M && c_ks == 0) {
            accv z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = 0.f;
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = pp_mfma<DT, 16>(af[mi], bf[ni], z);
        } else {
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = pp_mfma<DT, 16>(af[mi], bf[ni], acc[mi][ni]);
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- LOAD(s,1)
#if defined(PP_ABL) && (PP_ABL & 2)
        if (s == 0)
#endif
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = *(const pp_u32x4*)(sb + b_off + (4 + ni) * 1024);
#if !defined(PP_ABL) || !(PP_ABL & 1)
        if constexpr (QFIRST) issue_a(nbuf); else issue_b(nbuf);
        advance();
#endif
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMP(s,1)
        __builtin_amdgcn_s_setprio(1);
#if defined(PP_ABL) && (PP_ABL & 4)
#pragma unroll
        for (int mi = 0; mi < 6; ++mi) asm volatile("" :: "v"(af[mi]));      // timing ablation: no MFMAs, operands kept alive
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" :: "v"(bf[ni]));
        if (s == 0) {
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
        }
#else
        if (!L2M && c_ks == 0) {
            accv z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = 0.f;
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][4 + ni] = pp_mfma<DT, 16>(af[mi], bf[ni], z);
        } else {
#pragma unroll
            for (int mi = 0; mi < 6; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][4 + ni] = pp_mfma<DT, 16>(af[mi], bf[ni], acc[mi][4 + ni]);
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        buf = buf == 2 ? 0 : buf + 1;
        if (++c_ks == nks) {
            c_ks = 0;
            epi = true;
            epi_row0 = (p.tile0 + (blockIdx.x + c_seq * gridDim.x) * p.tile_stride) * PW_TM;
            ++c_seq;
        }
    }
    if (epi) epilogue(epi_row0);
    if (!DENSE) flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grp == 0) __builtin_amdgcn_s_barrier();    // matches the extra barrier of waves 4-7
#ifdef PP_STAMPS
    if (tid == 0) {
        g_pp_clk[(blockIdx.y * gridDim.x + blockIdx.x) & 4095] = (unsigned)(__builtin_amdgcn_s_memtime() - pp_clk0);
        g_pp_end[(blockIdx.y * gridDim.x + blockIdx.x) & 4095] = (unsigned)(wall_clock64() - pp_stamp0);
    }
#endif
}

}  // namespace vdb
