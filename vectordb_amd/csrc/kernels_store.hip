// Store-side kernels: device-side append (normalise for COSINE, round to the storage dtype,
// record the stored row's squared norm), the synthetic-embedding generator, and row read-back.
#include "vdb_internal.h"

namespace vdb {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Element (row, col) of the synthetic matrix for `seed`: the sum of the four 16-bit fields of a
// 64-bit counter hash, centred and scaled to unit variance.  Integer arithmetic plus one exact
// int->fp32 conversion and one fp32 multiply, so a host restatement is bit-identical.
__device__ __forceinline__ float synth_value(uint64_t rowkey, int col) {
    const uint64_t h = mix64(rowkey ^ (0xD1B54A32D192ED03ull * (uint64_t)(col + 1)));
    const int a = (int)(h & 0xffff) + (int)((h >> 16) & 0xffff) + (int)((h >> 32) & 0xffff) + (int)(h >> 48);
    return (float)(a - 131070) * 0x1.bb67aep-16f;
}
__device__ __forceinline__ uint64_t synth_rowkey(uint64_t seed, long long row) {
    return mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(row + 1));
}

__device__ __forceinline__ double wave_tree_sum_d(double acc) {
    for (int s = 32; s >= 1; s >>= 1) acc = __dadd_rn(acc, __shfl_xor(acc, s));
    return acc;
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ unsigned short f32_to_f16_rne(float f) {
    const _Float16 h = (_Float16)f;
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ __forceinline__ float f16_to_f32(unsigned short b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

// One wave per appended row.  Lane l owns elements j = l, l+64, ... (the interleaving of the
// canonical fp64 sums), so the norm of the raw row and of the stored row come out in the
// canonical order without any data exchange besides the final tree.
template <int DT>
__global__ void __launch_bounds__(256)
append_kernel(char* __restrict__ rows, unsigned short* __restrict__ rows16, long long ld16,
              long long* __restrict__ ids, float* __restrict__ n2out,
              float* __restrict__ max_n2, int* __restrict__ bad, long long row0, long long n, int d, int m, int metric,
              const float* __restrict__ raw, const long long* __restrict__ raw_ids,
              uint64_t seed, long long synth_first_row) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    constexpr int EPL = (DT == VDB_F32) ? 4 : 8;
    const int dpad = m * EPL;
    const bool synth = raw == nullptr;
    const uint64_t rk = synth ? synth_rowkey(seed, synth_first_row + i) : 0ull;
    const float* src = synth ? nullptr : raw + (size_t)i * d;
    char* dst = rows + (size_t)(row0 + i) * ((size_t)m * 16);

    double inv = 1.0;
    if (metric == VDB_METRIC_COSINE) {
        double acc = 0.0;
        for (int j = lane; j < d; j += 64) {
            const double x = (double)(synth ? synth_value(rk, j) : src[j]);
            acc = __dadd_rn(acc, __dmul_rn(x, x));
        }
        const double n2 = __shfl(wave_tree_sum_d(acc), 0);
        inv = (n2 == 0.0) ? 0.0 : 1.0 / sqrt(n2);
    }
    double acc2 = 0.0;
    bool finite = true;
    for (int j = lane; j < dpad; j += 64) {
        float y = 0.f;
        if (j < d) {
            const float x = synth ? synth_value(rk, j) : src[j];
            y = (metric == VDB_METRIC_COSINE) ? (float)__dmul_rn((double)x, inv) : x;
        }
        float stored;
        if constexpr (DT == VDB_F32) {
            ((float*)dst)[j] = y; stored = y;
            // bf16 mirror for the batched fast pass (finite values only reach this point)
            if (rows16) rows16[(size_t)(row0 + i) * ld16 + j] = f32_to_bf16_rne(y);
        } else if constexpr (DT == VDB_BF16) {
            const unsigned short b = f32_to_bf16_rne(y);
            ((unsigned short*)dst)[j] = b; stored = __uint_as_float((uint32_t)b << 16);
        } else {
            const unsigned short b = f32_to_f16_rne(y);
            ((unsigned short*)dst)[j] = b; stored = f16_to_f32(b);
        }
        finite = finite && (fabsf(stored) <= 3.0e38f);       // false for NaN, infinities and fp16 overflow
        acc2 = __dadd_rn(acc2, __dmul_rn((double)stored, (double)stored));
    }
    // a row that is not finite in the storage type makes the whole insert fail (the host does not
    // publish the rows of this call) and stays out of the table-wide norm bound
    finite = __all(finite);
    if constexpr (DT == VDB_F32) {
        if (rows16) for (int j = dpad + lane; j < ld16; j += 64) rows16[(size_t)(row0 + i) * ld16 + j] = 0;
    }
    const double sn2 = wave_tree_sum_d(acc2);
    if (lane == 0) {
        // round up so that the value bounds the row norm from above
        float f = (float)sn2;
        if ((double)f < sn2) f = __uint_as_float(__float_as_uint(f) + 1u);
        n2out[row0 + i] = (float)sn2;
        if (finite && f <= 3.0e38f) atomicMax((unsigned int*)max_n2, __float_as_uint(f));
        else *bad = 1;
        ids[row0 + i] = synth ? (synth_first_row + i) : raw_ids[i];
    }
}

hipError_t launch_append(char* rows, unsigned short* rows16, long long ld16, long long* ids, float* n2, float* max_n2,
                         int* bad, long long row0, long long n, int d, int m, int dt, int metric,
                         const float* raw, const long long* raw_ids,
                         uint64_t seed, long long synth_first_row, hipStream_t st) {
    // One wave per row: a launch holds at most 2^24 rows, because a grid of 2^32 work-items or more
    // is not launched whole (125 M rows in one launch filled 58 M of them and reported no error).
    const long long chunk = 1ll << 24;
    for (long long c0 = 0; c0 < n; c0 += chunk) {
        const long long nc = n - c0 < chunk ? n - c0 : chunk;
        const dim3 block(256), grid((unsigned)((nc + 3) / 4));
        const float* raw_c = raw ? raw + (size_t)c0 * d : nullptr;
        const long long* ids_c = raw_ids ? raw_ids + c0 : nullptr;
#define VDB_APPEND(DT)                                                                               \
        hipLaunchKernelGGL((append_kernel<DT>), grid, block, 0, st, rows, rows16, ld16, ids, n2, max_n2, bad, row0 + c0, nc, \
                           d, m, metric, raw_c, ids_c, seed, synth_first_row + c0)
        if (dt == VDB_F32) VDB_APPEND(VDB_F32);
        else if (dt == VDB_BF16) VDB_APPEND(VDB_BF16);
        else VDB_APPEND(VDB_F16);
#undef VDB_APPEND
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// An append collects the largest squared norm of its rows in pending[0] (the append kernel's max_n2
// argument); it joins the table-wide bound only if no row of the call was refused, so that a refused
// insert leaves no trace in the bound the certificates (and snapshots) use.
// The table's bound after the fold is also left in bad[1] (host-visible memory) for the host to read.
__global__ void fold_max_kernel(float* __restrict__ table_max, const float* __restrict__ pending,
                                int* __restrict__ bad) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float m = *table_max;
        if (*bad == 0 && *pending > m) { m = *pending; *table_max = m; }      // appends to a table run one at a time
        bad[1] = (int)__float_as_uint(m);
    }
}

hipError_t launch_fold_max(float* table_max, const float* pending, int* bad, hipStream_t st) {
    hipLaunchKernelGGL(fold_max_kernel, dim3(1), dim3(64), 0, st, table_max, pending, bad);
    return hipGetLastError();
}

__device__ __forceinline__ bool norm_sampled(long long r) { return r < 4096 || (r & 63) == 0; }

__global__ void __launch_bounds__(256)
norm_hist_kernel(const float* __restrict__ n2, long long row0, long long n, unsigned int* __restrict__ hist) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = row0 + i;
        if (!norm_sampled(r)) continue;
        const float v = n2[r];
        if (v > 0.f) atomicAdd(&hist[(__float_as_uint(v) >> 23) & 255u], 1u);     // zero rows say nothing about the scale
    }
}

hipError_t launch_norm_hist(const float* n2, long long row0, long long n, unsigned int* hist256, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(norm_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n2, row0, n, hist256);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
collect_outliers_kernel(const float* __restrict__ n2, long long row0, long long n, float limit,
                        long long* __restrict__ list, int* __restrict__ n_out, float* __restrict__ max_rest) {
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = row0 + i;
        const float v = n2[r];
        if (v > limit) {
            const int pos = atomicAdd(n_out, 1);
            if (pos < OUTLIER_CAP) list[pos] = r;
        } else {
            mx = fmaxf(mx, v);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax((unsigned int*)max_rest, __float_as_uint(mx));
}

hipError_t launch_collect_outliers(const float* n2, long long row0, long long n, float limit, long long* list,
                                   int* n_out, float* max_rest, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(collect_outliers_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n2, row0, n, limit, list, n_out, max_rest);
    return hipGetLastError();
}

// bf16 mirror of stored fp32 rows (tables loaded from a snapshot): one wave per row
__global__ void __launch_bounds__(256)
build_mirror_kernel(const char* __restrict__ rows, unsigned short* __restrict__ rows16, long long ld16,
                    long long row0, long long n, int m) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* src = (const float*)(rows + (size_t)(row0 + i) * ((size_t)m * 16));
    unsigned short* dst = rows16 + (size_t)(row0 + i) * ld16;
    const int dpad = m * 4;
    for (int j = lane; j < ld16; j += 64) dst[j] = j < dpad ? f32_to_bf16_rne(src[j]) : (unsigned short)0;
}

hipError_t launch_build_mirror(const char* rows, unsigned short* rows16, long long ld16, long long row0,
                               long long n, int m, hipStream_t st) {
    const long long chunk = 1ll << 24;
    for (long long c0 = 0; c0 < n; c0 += chunk) {
        const long long nc = n - c0 < chunk ? n - c0 : chunk;
        hipLaunchKernelGGL(build_mirror_kernel, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, st, rows, rows16, ld16,
                           row0 + c0, nc, m);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

__global__ void __launch_bounds__(256)
synth_fill_kernel(float* __restrict__ out, long long n, int d, uint64_t seed, long long first_row) {
    const long long total = n * d;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / d;
        const int c = (int)(e - r * d);
        out[e] = synth_value(synth_rowkey(seed, first_row + r), c);
    }
}

hipError_t launch_synth_fill(float* out, long long n, int d, uint64_t seed, long long first_row,
                             hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n * d + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, n, d, seed, first_row);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
fetch_rows_kernel(const char* __restrict__ rows, long long first, long long n, int d, int m, int dt,
                  float* __restrict__ out) {
    const long long total = n * d;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / d;
        const int c = (int)(e - r * d);
        const char* row = rows + (size_t)(first + r) * ((size_t)m * 16);
        float v;
        if (dt == VDB_F32) v = ((const float*)row)[c];
        else {
            const unsigned short b = ((const unsigned short*)row)[c];
            v = (dt == VDB_BF16) ? __uint_as_float((uint32_t)b << 16) : f16_to_f32(b);
        }
        out[e] = v;
    }
}

hipError_t launch_fetch_rows(const char* rows, long long first, long long n, int d, int m, int dt,
                             float* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n * d + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(fetch_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, rows, first, n, d, m, dt, out);
    return hipGetLastError();
}

}  // namespace vdb
