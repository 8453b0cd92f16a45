// EXPERIMENT, not part of the library (measured slower than the eight-wave kernel, see the end of this comment).
// Batched fast pass, four-wave form of the kernel in vectordb_amd/csrc/gemm_pp.h: the same 256 rows x 256 queries per
// workgroup, the same stage buffers (operands straight from global memory into LDS, four 32-element
// k-steps in a ring) and the same thresholded epilogue, but one wave per SIMD owning 128 rows x 128
// queries (8 x 8 blocks of v_mfma_f32_16x16x32).  A wave reads 16 fragments per k-step for 64 MFMAs
// (the eight-wave form: 12 for 32), so the fragment traffic out of LDS falls by a third, and there is
// one barrier per k-step instead of two.  Each wave software-pipelines by itself: while the MFMAs of
// k-step s run, the fragments of s+1 are read into a second register set and the loads of s+4 are issued.
//
//   iteration s:  s_waitcnt vmcnt (own loads of s+1 landed), lgkmcnt(0) (fragments of s have arrived);
//                 s_barrier  -> every wave's part of s+1 is in LDS, nobody reads buffer s & 3 any more
//                 [L2, first k-step of a tile: accumulators start at the rows' norms, read from LDS]
//                 issue loads of s+4 into buffer s & 3;  ds_read fragments of s+1;  64 MFMAs of s
//                 [last k-step of a tile: epilogue]
//
// Needs an even number of k-steps.  The tile loop is outside and the k-loop inside, without branches: with
// one flat loop and a conditional epilogue the register allocator kept accumulators in VGPRs and moved
// them through scratch memory (256 accumulators are all of a wave's AGPRs).
// Measured (4 M x 768 bf16 rows, 1024 queries, MI355X): 1279 TFLOP/s against 1425 for the eight-wave
// kernel; 1804 with the loads of the loop removed.  A wave that is alone on its SIMD issues in order: when
// a global_load_lds has to wait for room in the memory pipeline, which it does whenever the L2 -> LDS fill
// rate is the bound, the MFMAs behind it wait as well; with two waves per SIMD the other one multiplies.
#pragma once
#include "../../vectordb_amd/csrc/gemm_pp.h"

namespace vdb {

constexpr int PP4_QCAP = 384;                      // candidate queue entries per wave (4 waves: the same 24 KiB)

template <int DT, bool L2M, bool DENSE, bool TILED>
__global__ void __launch_bounds__(256, 1)
gemm_pp4_kernel(GemmPPParams p) {
    extern __shared__ __attribute__((aligned(16))) char pp_smem[];
    constexpr int NMI = 8, NNI = 8;
    typedef pp_f32x4 accv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wave tile: rows wm*128.., queries wn*128..
    const int li = lane & 15, lh = lane >> 4;
    const int qt = blockIdx.y;
    const int nks = p.nks;

    const int fsw = (-(li >> 2)) & 3;
    const int a_off = (wm * 128 + li) * 64 + ((lh ^ fsw) << 4);
    const int b_off = 16384 + (wn * 128 + li) * 64 + ((lh ^ fsw) << 4);

    // load map: wave w fills rows w*64 .. w*64+63 of the A tile and of the B tile, four pieces of 16 rows
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const unsigned voff = (unsigned)(wave * 4096 + lane * 16);
    const char* const btile = (const char*)p.Q + (((size_t)qt * nks) << 14);
    float thr[NNI];
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) {
        const int q = qt * PP_TN + wn * 128 + ni * 16 + li;
        const float t = (q < p.nq) ? p.thr[q] : __builtin_huge_valf();
        thr[ni] = L2M ? -t : t;
    }
    bool rbon = false;
    float ea = 0.f, es = 0.f;
    if constexpr (L2M) {
        rbon = p.rb.qinfo != nullptr;
        if (rbon) { ea = p.rb.a; es = rowbound_s(p.rb, lane); }
        asm volatile("" : "+v"(ea), "+v"(es));
    }
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) asm volatile("" : "+v"(thr[ni]));

    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long total = my_tiles * nks;

    long long f_seq = 0;
    int f_ks = 0;
    long long f_left = total;
    const unsigned short* asrc[4] = {nullptr, nullptr, nullptr, nullptr};
    const char* atile = nullptr;
    const float* nsrc = nullptr;
    auto set_tile_src = [&]() __attribute__((always_inline)) {
        const long long tile = p.tile0 + (blockIdx.x + f_seq * gridDim.x) * p.tile_stride;
        const long long row0 = tile * PP_TM;
        if constexpr (TILED) {
            atile = (const char*)p.X + (((size_t)tile * nks) << 14);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                long long r = row0 + (wave * 4 + i) * 16 + lrow;
                if (r > p.n - 1) r = p.n - 1;
                asrc[i] = p.X + (size_t)r * p.ldx + lchunk * 8;
            }
        }
        if constexpr (L2M) {
            long long r = row0 + wave * 64 + lane;
            if (r > p.n - 1) r = p.n - 1;
            nsrc = p.n2 + r;
        }
    };
    set_tile_src();
    // Every k-step issues the same number of loads (stages past the end re-read the last one's operands
    // into a buffer nobody reads), so that the counted waits hold on every path.  The source addresses of a
    // stage are taken first (load_src), the cursor moves on, and the loads themselves go out one by one
    // between the MFMAs (load_piece).
    const char* la[4];
    const char* lb;
    const float* ln = nullptr;
    auto load_src = [&]() __attribute__((always_inline)) {
        if constexpr (TILED) {
            la[0] = atile + ((size_t)f_ks << 14) + voff;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) la[i] = (const char*)(asrc[i] + f_ks * PP_KS);
        }
        if constexpr (L2M) ln = nsrc;
        lb = btile + ((size_t)f_ks << 14) + voff;
        if (--f_left > 0) {
            if (++f_ks == nks) { f_ks = 0; ++f_seq; set_tile_src(); }
        }
    };
    auto load_piece = [&](int buf, int piece) __attribute__((always_inline)) {       // piece 0-3: A, 4-7: B, 8: norms
        char* dst = pp_smem + buf * PP_STAGE + wave * 4096;
        if (piece < 4) {
            if constexpr (TILED) {
                if (piece == 0) pp_glds16<0>(la[0], dst);
                if (piece == 1) pp_glds16<1024>(la[0], dst);
                if (piece == 2) pp_glds16<2048>(la[0], dst);
                if (piece == 3) pp_glds16<3072>(la[0], dst);
            } else {
                pp_glds16(la[piece], dst + piece * 1024);
            }
        } else if (piece < 8) {
            if (piece == 4) pp_glds16<0>(lb, dst + 16384);
            if (piece == 5) pp_glds16<1024>(lb, dst + 16384);
            if (piece == 6) pp_glds16<2048>(lb, dst + 16384);
            if (piece == 7) pp_glds16<3072>(lb, dst + 16384);
        } else {
            if constexpr (L2M) pp_glds4(ln, pp_smem + PP_N2OFF + buf * 1024 + wave * 256);
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        load_src();
#pragma unroll
        for (int piece = 0; piece < 9; ++piece) load_piece(buf, piece);
    };
    constexpr int NLD = L2M ? 9 : 8;               // loads per wave and stage
    // s_waitcnt vmcnt(2 * NLD) lgkmcnt(0), expcnt free: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
    constexpr int PP4_WAIT = ((2 * NLD) & 15) | (7 << 4) | (0 << 8) | (((2 * NLD) >> 4) << 14);
    issue(0); issue(1); issue(2); issue(3);
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(3 * NLD) : "memory");
    __builtin_amdgcn_s_barrier();

    accv acc[NMI][NNI];

    char* const queue = pp_smem + PP_QOFF(L2M) + wave * (PP4_QCAP * 16);
    int qn = 0;
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
        const int q0 = qt * PP_TN + wn * 128 + li;
        if (!DENSE && qn > 64) flush();            // room for 320 entries in every tile
        if constexpr (DENSE) {
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = wm * 128 + mi * 16 + 4 * lh + r;
                        const int q = q0 + ni * 16;
                        if (lr < nloc && q < p.nq) p.dense[(size_t)(row0 + lr) * p.nq + q] = L2M ? -acc[mi][ni][r] : acc[mi][ni][r];
                    }
            return;
        }
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) {
            // the accumulators live in AGPRs: one query block's 32 values at a time come over to VGPRs
            __builtin_amdgcn_sched_barrier(0);
            float best = L2M ? __builtin_huge_valf() : -__builtin_huge_valf();
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) best = L2M ? fminf(best, acc[mi][ni][r]) : fmaxf(best, acc[mi][ni][r]);
            const bool any = L2M ? best < thr[ni] : best > thr[ni];
            if (__builtin_expect(__ballot(any) != 0ull, 0)) {
                int lb = wm * 128 + 4 * lh, qq = q0 + ni * 16;
                asm volatile("" : "+v"(lb), "+v"(qq));
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) {
                    float bm = acc[mi][ni][0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) bm = L2M ? fminf(bm, acc[mi][ni][r]) : fmaxf(bm, acc[mi][ni][r]);
                    if (__ballot(L2M ? bm < thr[ni] : bm > thr[ni]) == 0ull) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[mi][ni][r];
                        const int lr = lb + (mi * 16 + r);
                        const bool hit = (L2M ? v < thr[ni] : v > thr[ni]) && lr < nloc;
                        const unsigned long long mask = __ballot(hit);
                        if (mask == 0ull) continue;
                        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (hit) {
                            if (slot < PP4_QCAP) {
                                const unsigned long long row = (unsigned long long)row0 + (unsigned)lr;
                                pp_u32x4 e;
                                e[0] = __float_as_uint(L2M ? -v : v);
                                e[1] = (unsigned)row;
                                e[2] = (unsigned)(row >> 32);
                                e[3] = (unsigned)qq;
                                *(pp_u32x4*)(queue + slot * 16) = e;
                            } else {
                                p.lost[qq] = 1;        // more than 320 hits in one 128 x 128 tile: the exact pass answers
                            }
                        }
                        qn += __builtin_popcountll(mask);
                        if (qn > PP4_QCAP) qn = PP4_QCAP;
                    }
                }
            }
        }
    };

    pp_u32x4 af[2][NMI], bf[2][NNI];
    auto read_frags = [&](int buf, pp_u32x4 (&a)[NMI], pp_u32x4 (&b)[NNI]) __attribute__((always_inline)) {
        const char* sb = pp_smem + buf * PP_STAGE;
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) b[ni] = *(const pp_u32x4*)(sb + b_off + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) a[mi] = *(const pp_u32x4*)(sb + a_off + mi * 1024);
    };
    read_frags(0, af[0], bf[0]);

    // one k-step: `s` counts the k-steps of this workgroup (its low bits name the ring buffer)
    auto kstep = [&](long long s, bool first, pp_u32x4 (&ca)[NMI], pp_u32x4 (&cb)[NNI], pp_u32x4 (&na)[NMI], pp_u32x4 (&nb)[NNI]) __attribute__((always_inline)) {
        // the builtin, not inline assembly: the compiler's own wait insertion has to see that the fragments
        // read in the previous k-step have arrived, or it waits again right before their first use, behind
        // the reads issued in this k-step
#ifdef PP4_NOLOAD
        __builtin_amdgcn_s_waitcnt(0xC07F);
#else
        __builtin_amdgcn_s_waitcnt(PP4_WAIT);
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (first) {
            if constexpr (L2M) {
                // the rows' norms, before the loads of s+4 overwrite this slot
                const char* nbp = pp_smem + PP_N2OFF + (int)(s & 3) * 1024 + (wm * 128 + 4 * lh) * 4;
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) {
                    pp_f32x4 v = *(const pp_f32x4*)(nbp + (mi * 16) * 4);
                    if (rbon) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rowbound_adj(v[e], ea, es);
                    }
#pragma unroll
                    for (int ni = 0; ni < NNI; ++ni) acc[mi][ni] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        load_src();
        __builtin_amdgcn_sched_barrier(0);
        // 16 groups of four MFMAs; behind each of the first eight, two fragment reads of the next k-step (they
        // have half a k-step to arrive before the wait at the top of the next one); behind the last nine,
        // one load of the k-step four ahead.  A wave is alone on its SIMD: whatever is not issued in the
        // shadow of an MFMA leaves the matrix unit idle
        const int lbuf = (int)(s & 3);
        const char* sb = pp_smem + (int)((s + 1) & 3) * PP_STAGE;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int mi = g >> 1, n0 = (g & 1) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mi][n0 + j] = pp_mfma<DT, 16>(ca[mi], cb[n0 + j], acc[mi][n0 + j]);
            if (g < 8) {
                nb[g] = *(const pp_u32x4*)(sb + b_off + g * 1024);
                na[g] = *(const pp_u32x4*)(sb + a_off + g * 1024);
            }
#ifndef PP4_NOLOAD
            if (!(g & 1)) load_piece(lbuf, g >> 1);
            if (g == 15) load_piece(lbuf, 8);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // nks is even (pp4_eligible): the fragment sets alternate and a tile ends on the second one
    long long s = 0;
    for (long long t = 0; t < my_tiles; ++t) {
        kstep(s, true, af[0], bf[0], af[1], bf[1]);
        kstep(s + 1, false, af[1], bf[1], af[0], bf[0]);
        s += 2;
        for (int ks = 2; ks < nks; ks += 2) {
            kstep(s, false, af[0], bf[0], af[1], bf[1]);
            kstep(s + 1, false, af[1], bf[1], af[0], bf[0]);
            s += 2;
        }
        epilogue((p.tile0 + (blockIdx.x + t * gridDim.x) * p.tile_stride) * PP_TM);
    }
    if (!DENSE) flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace vdb
