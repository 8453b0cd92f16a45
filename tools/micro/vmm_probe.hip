// Probe of the HIP virtual-memory API as the table store uses it: which sequences of chunk sizes can
// be mapped one after the other into one reserved range.  No kernel touches a chunk whose set-up failed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void touch(char* p, size_t n) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = (char)i; }
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
static bool run(const char* name, const std::vector<size_t>& sizes, int mode) {
    // mode 0: set access on the new chunk only; 1: on everything mapped so far
    void* base = nullptr;
    const size_t reserve = (size_t)8 << 30;
    if (hipMemAddressReserve(&base, reserve, 0, nullptr, 0) != hipSuccess) { printf("%s: reserve failed\n", name); (void)hipGetLastError(); return false; }
    size_t mapped = 0;
    bool ok = true;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t i = 0; i < sizes.size() && ok; ++i) {
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, sizes[i], &prop, 0);
        if (e != hipSuccess) { printf("%s: chunk %zu (%zu MiB at %zu MiB): hipMemCreate -> %s\n", name, i, sizes[i] >> 20, mapped >> 20, hipGetErrorString(e)); ok = false; break; }
        e = hipMemMap((char*)base + mapped, sizes[i], 0, h, 0);
        if (e != hipSuccess) { printf("%s: chunk %zu (%zu MiB at %zu MiB): hipMemMap -> %s\n", name, i, sizes[i] >> 20, mapped >> 20, hipGetErrorString(e)); ok = false; break; }
        hs.push_back(h);
        if (mode == 0) e = hipMemSetAccess((char*)base + mapped, sizes[i], &acc, 1);
        else e = hipMemSetAccess(base, mapped + sizes[i], &acc, 1);
        if (e != hipSuccess) { printf("%s: chunk %zu (%zu MiB at %zu MiB): hipMemSetAccess -> %s\n", name, i, sizes[i] >> 20, mapped >> 20, hipGetErrorString(e)); ok = false; break; }
        touch<<<(unsigned)((sizes[i] / 4096 + 255) / 256), 256>>>((char*)base + mapped, sizes[i]);
        e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%s: chunk %zu: kernel -> %s\n", name, i, hipGetErrorString(e)); return false; }
        mapped += sizes[i];
    }
    (void)hipGetLastError();
    printf("%s: %s (%zu MiB mapped in %zu chunks)\n", name, ok ? "ok" : "FAILED", mapped >> 20, hs.size());
    return ok;
}
int main() {
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t g = (size_t)2 << 20;
    run("A equal 2 MiB x6", {g, g, g, g, g, g}, 0);
    run("B equal 8 MiB x3", {4 * g, 4 * g, 4 * g}, 0);
    run("C 2 then 4 MiB", {g, 2 * g}, 0);
    run("D 4 then 2 MiB", {2 * g, g}, 0);
    run("E 2 then 6 MiB", {g, 3 * g}, 0);
    run("F 128 MiB x2, 1 GiB x2", {64 * g, 64 * g, 512 * g, 512 * g}, 0);
    run("G 2,4,8 MiB, access on all mapped", {g, 2 * g, 4 * g}, 1);
    run("H 2,2,4,8,16 MiB", {g, g, 2 * g, 4 * g, 8 * g}, 0);
    run("I 1 GiB then 2 MiB", {512 * g, g}, 0);
    return 0;
}
