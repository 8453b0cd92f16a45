// Measures the MFMA issue rate this GPU sustains with no memory traffic: 4 independent
// accumulators per wave, 1 or 2 waves per SIMD.  Prints TFLOP/s for f16 32x32x16 and f32 32x32x2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    float fa = threadIdx.x * 0.001f, fb = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, double flop_per_mfma, int blocks_per_cu) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 20000;
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<KIND><<<cus * blocks_per_cu, 256>>>(out, 100);
    hipEventRecord(e0);
    mfma_loop<KIND><<<cus * blocks_per_cu, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * blocks_per_cu * 4 /*waves*/ * iters * 4.0 * flop_per_mfma;
    printf("%s, %d waves/SIMD: %.1f TFLOP/s (%.2f ms, %d CUs, clock %d MHz)\n", name, blocks_per_cu,
           flops / (ms * 1e-3) / 1e12, ms, cus, prop.clockRate / 1000);
    hipFree(out);
}

int main() {
    run<0>("mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 1);
    run<0>("mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 2);
    run<1>("mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 1);
    run<1>("mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 2);
    return 0;
}
