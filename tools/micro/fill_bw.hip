// How fast can one CU move operand tiles from global memory into LDS?  The batched kernel is bound by
// this rate, so the variants here (LDS-DMA vs registers + ds_write, waves per workgroup, stages in
// flight, streamed vs cache-resident source) are timed alone, with the kernel's barrier per stage.
//   fill_bw [stream_gb]          prints bytes per clock and CU for every variant
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))
#define LOCAL_AS __attribute__((address_space(3)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int STAGE = 32768;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N < 63 ? N : 63) : "memory"); }

// METHOD 0: global_load_lds_dwordx4.  METHOD 1: global_load_dwordx4 + ds_write_b128.
// METHOD 2: the streamed half by global_load_lds_dwordx4, the hot half by global_load_dwordx4 into
// registers that never go to LDS (what fragment-ordered query loads straight to VGPRs would do).
// METHOD 3: as 2 with every hot piece loaded twice (two waves of a workgroup need the same query fragment).
// Half of every stage comes from `stream` (each byte read once: HBM), half from `hot` (a buffer of
// hot_bytes that every workgroup re-reads: L2), as the row and query operands of the GEMM do.
template <int METHOD, int WAVES, int DEPTH, int SEG = 0>
__global__ void __launch_bounds__(WAVES * 64)
fill_kernel(const char* __restrict__ stream, const char* __restrict__ hot, size_t hot_bytes, long stages,
            int stream_only, int share, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PER = STAGE / (WAVES * 1024);          // 1 KiB wave-instructions per wave and stage
    static_assert(PER >= 2 && PER % 2 == 0, "stage split");
    // share > 1: workgroups b, b + G/share, ... read the same stream (G/share a multiple of 8: one XCD, one L2)
    const char* sbase = stream + (size_t)(blockIdx.x % (gridDim.x / share)) * stages * (stream_only ? STAGE : STAGE / 2);
    unsigned acc = 0;
    u32x4 regs[DEPTH][PER];
    u32x4 regs2[METHOD == 3 ? DEPTH : 1][PER];
    auto issue = [&](long s, auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool from_stream = stream_only || i < PER / 2;
            const size_t piece = (size_t)wave * PER + i;           // 1 KiB piece of the stage
            const char* src;
            if (stream_only) src = sbase + (size_t)s * STAGE + piece * 1024 + lane * 16;
            else if (from_stream && SEG == 0) src = sbase + (size_t)s * (STAGE / 2) + ((size_t)wave * (PER / 2) + i) * 1024 + lane * 16;
            else if (from_stream) {
                // row-major operand: rows of 1536 bytes, a tile of 256 rows is read SEG bytes per row and stage
                // (24 * 64 / SEG stages per tile); a wave-instruction covers 1024 / SEG rows
                // SEG 1128: 128 bytes per row and stage, fetched as two 64-byte instructions back to back
                constexpr bool PAIR = SEG == 1128;
                constexpr int SG = PAIR ? 128 : (SEG ? SEG : 64);
                constexpr int LPS = PAIR ? 4 : SG / 16;              // lanes per row segment of an instruction
                constexpr int RPS = (STAGE / 2) / SG;                // rows a stage covers (a tile has 256)
                constexpr int BLK = 256 / RPS;                       // stages per SG-byte column of a tile
                const size_t tile = (size_t)s / 24, sub = (size_t)s % 24;     // 24 stages of 16 KiB per tile
                const size_t ks = sub / BLK, rowblock = sub % BLK;
                const size_t pieces = (size_t)wave * (PER / 2) + i;  // of the 16 KiB operand half: 1 KiB each
                const size_t row = PAIR ? rowblock * RPS + (pieces >> 1) * 16 + lane / LPS
                                        : rowblock * RPS + pieces * (1024 / SG) + lane / LPS;
                src = sbase + (tile * 256 + row) * 1536 + ks * SG + (PAIR ? (pieces & 1) * 64 : 0) + (lane % LPS) * 16;
            }
            else src = hot + (((size_t)s * (STAGE / 2) + ((size_t)wave * (PER / 2) + (i - PER / 2)) * 1024) % hot_bytes) + lane * 16;
            char* dst = smem + buf * STAGE + piece * 1024;
            if constexpr (METHOD == 0) {
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LOCAL_AS void*)dst, 16, 0, 0);
            } else if constexpr (METHOD >= 2) {
                if (from_stream) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LOCAL_AS void*)dst, 16, 0, 0);
                else {
                    regs[buf][i] = *(const u32x4*)src;
                    if constexpr (METHOD == 3) regs2[buf][i] = *(const u32x4*)(hot + (((size_t)(src - hot) + 4096) % hot_bytes));
                }
            } else {
                regs[buf][i] = __builtin_nontemporal_load((const u32x4*)src);
            }
        }
    };
    static_for<DEPTH - 1>([&](auto d) __attribute__((always_inline)) { issue((long)decltype(d)::value, d); });
    // `stages` is a multiple of DEPTH: the ring position is a compile-time constant in the loop body
    for (long s0 = 0; s0 < stages; s0 += DEPTH) {
        static_for<DEPTH>([&](auto bc) __attribute__((always_inline)) {
            constexpr int buf = decltype(bc)::value;
            const long s = s0 + buf;
            const long nx = s + DEPTH - 1;
            issue(nx < stages ? nx : stages - 1, std::integral_constant<int, (buf + DEPTH - 1) % DEPTH>{});
            if constexpr (METHOD == 0) {
                wait_vm<(DEPTH - 1) * PER>();
            } else if constexpr (METHOD == 2 || METHOD == 3) {
                constexpr int LPS = METHOD == 3 ? PER + PER / 2 : PER;      // loads per stage and wave
                wait_vm<(DEPTH - 1) * LPS>();
                static_for<PER>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    if (i >= PER / 2) { acc ^= regs[buf][i][0] ^ regs[buf][i][3]; if constexpr (METHOD == 3) acc ^= regs2[buf][i][1]; }
                });
            } else {
                // the loads of stage s are the oldest PER in flight
                static_for<PER>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    wait_vm<(DEPTH - 1) * PER + (PER - 1 - i)>();
                    *(u32x4*)(smem + buf * STAGE + ((size_t)wave * PER + i) * 1024 + lane * 16) = regs[buf][i];
                });
            }
            __builtin_amdgcn_s_barrier();
            acc += *(const unsigned*)(smem + buf * STAGE + ((tid * 4) & (STAGE - 1)));
            __builtin_amdgcn_s_barrier();
        });
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int METHOD, int WAVES, int DEPTH, int SEG = 0>
static void run(const char* name, const char* stream, size_t stream_bytes, const char* hot, size_t hot_bytes,
                int blocks, int stream_only, int share, unsigned* sink, double clock_ghz) {
    const size_t per_block = stream_bytes / (blocks / share);
    const long stages = (long)(per_block / (stream_only ? STAGE : STAGE / 2)) / 12 * 12;
    const size_t lds = (size_t)DEPTH * STAGE;
    CK(hipFuncSetAttribute((const void*)fill_kernel<METHOD, WAVES, DEPTH, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fill_kernel<METHOD, WAVES, DEPTH, SEG>), dim3(blocks), dim3(WAVES * 64), lds, 0, stream, hot, hot_bytes, stages, stream_only, share, sink);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    const double bytes = (double)stages * STAGE * blocks;
    printf("%-30s blocks %3d share %d %s  %7.3f ms  %6.2f TB/s  %5.1f GB/s per CU  %5.1f B/clk/CU at %.1f GHz\n", name, blocks, share,
           stream_only ? "stream only  " : "half stream  ", best, bytes / best / 1e9, bytes / blocks / best / 1e6,
           bytes / blocks / (best * 1e-3) / (clock_ghz * 1e9), clock_ghz);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t stream_bytes = (size_t)(gb * 1e9) / (256 * STAGE) * (256 * STAGE);
    const size_t hot_bytes = 3 << 19;     // 1.5 MiB: the 1024 x 768 query image
    char *stream, *hot; unsigned* sink;
    CK(hipMalloc(&stream, stream_bytes + (1 << 20))); CK(hipMalloc(&hot, hot_bytes + (1 << 20))); CK(hipMalloc(&sink, 64));
    CK(hipMemset(stream, 1, stream_bytes)); CK(hipMemset(hot, 2, hot_bytes));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate / 1e6;
    const int cus = prop.multiProcessorCount;
    const double sgb = gb / 4;       // shared streams: a quarter of the bytes, read four times
    const size_t sb4 = (size_t)(sgb * 1e9) / (256 * STAGE) * (256 * STAGE);
    for (int blocks : {cus, cus / 2}) {
        run<0, 8, 4>("lds-dma  8 waves 4 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<0, 8, 3>("lds-dma  8 waves 3 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<0, 4, 4>("lds-dma  4 waves 4 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<0, 16, 4>("lds-dma 16 waves 4 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<1, 8, 2>("registers 8 waves 2 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<1, 8, 3>("registers 8 waves 3 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<1, 16, 3>("registers 16 waves 3 stages", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<2, 8, 4>("rows lds-dma, queries to VGPRs", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
        run<3, 8, 4>("same, query pieces loaded twice", stream, sb4, hot, hot_bytes, blocks, 0, 4, sink, ghz);
    }
    run<2, 8, 4>("rows lds-dma, queries to VGPRs", stream, stream_bytes, hot, hot_bytes, cus, 0, 1, sink, ghz);
    run<3, 8, 4>("same, query pieces loaded twice", stream, stream_bytes, hot, hot_bytes, cus, 0, 1, sink, ghz);
    // row-major row operand, one query tile (nothing shared): 64-byte and 128-byte row segments per stage
    run<0, 8, 4, 64>("lds-dma 8w 4st, rows 64 B/stage", stream, stream_bytes, hot, hot_bytes, cus, 0, 1, sink, ghz);
    run<0, 8, 4, 128>("lds-dma 8w 4st, rows 128 B/stage", stream, stream_bytes, hot, hot_bytes, cus, 0, 1, sink, ghz);
    run<0, 8, 4, 256>("lds-dma 8w 4st, rows 256 B/stage", stream, stream_bytes, hot, hot_bytes, cus, 0, 1, sink, ghz);
    run<0, 8, 4, 1128>("lds-dma 8w 4st, rows 2 x 64 B/stage", stream, stream_bytes, hot, hot_bytes, cus, 0, 1, sink, ghz);
    run<0, 8, 4, 1128>("lds-dma 8w 4st, rows 2 x 64 B, share 4", stream, sb4, hot, hot_bytes, cus, 0, 4, sink, ghz);
    run<0, 8, 4, 64>("lds-dma 8w 4st, rows 64 B, share 4", stream, sb4, hot, hot_bytes, cus, 0, 4, sink, ghz);
    run<0, 8, 4, 128>("lds-dma 8w 4st, rows 128 B, share 4", stream, sb4, hot, hot_bytes, cus, 0, 4, sink, ghz);
    for (int so = 0; so < 2; ++so) {
        for (int blocks : {cus, cus / 2}) {
            run<0, 8, 4>("lds-dma  8 waves 4 stages", stream, stream_bytes, hot, hot_bytes, blocks, so, 1, sink, ghz);
            run<1, 8, 3>("registers 8 waves 3 stages", stream, stream_bytes, hot, hot_bytes, blocks, so, 1, sink, ghz);
        }
    }
    return 0;
}
