// What a blocking call costs before any work: N dependent empty kernels on one stream + hipStreamSynchronize,
// timed from the host.   hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void nop_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void flag_kernel(volatile int* host_word, int v) { if (threadIdx.x == 0) *host_word = v; }
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    int* hw = nullptr;
    hipHostMalloc((void**)&hw, 64, hipHostMallocMapped);
    int* dw = nullptr;
    hipHostGetDevicePointer((void**)&dw, hw, 0);
    for (int nk = 1; nk <= 5; ++nk) {
        std::vector<double> us;
        for (int it = 0; it < 600; ++it) {
            auto t0 = std::chrono::steady_clock::now();
            for (int j = 0; j < nk; ++j) hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
            hipStreamSynchronize(st);
            auto t1 = std::chrono::steady_clock::now();
            if (it >= 100) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(us.begin(), us.end());
        printf("%d dependent empty kernels + hipStreamSynchronize: p50 %.1f us  p10 %.1f  p90 %.1f\n", nk, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
    }
    // the same with the host spinning on a word the kernel writes into pinned host memory instead of synchronising
    {
        std::vector<double> us;
        for (int it = 0; it < 600; ++it) {
            *(volatile int*)hw = 0;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, st, (volatile int*)dw, it + 1);
            while (*(volatile int*)hw != it + 1) { }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            if (it >= 100) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(us.begin(), us.end());
        printf("1 kernel that writes a host word, host spins on the word: p50 %.1f us  p10 %.1f  p90 %.1f\n", us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
    }
    return 0;
}
