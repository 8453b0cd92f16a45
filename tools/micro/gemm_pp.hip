// Stand-alone check and timing of the ping-pong batched kernel (vectordb_amd/csrc/gemm_pp.h).
//   gemm_pp check            small shapes against a plain reference kernel (all keys, then candidates)
//   gemm_pp time [rows] [nq] [dim] [reps] [l2]
#include "../../vectordb_amd/csrc/gemm_pp.h"
#include "gemm_pp4.h"
#include "../../vectordb_amd/csrc/gemm_pp384.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace vdb;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}
// roughly normal values (sum of four uniforms), rounded to bf16
__global__ void fill_bf16(unsigned short* out, long long count, unsigned long long seed, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const unsigned h = hash32(i * 2654435761ull + seed), g = hash32(i * 40503ull + seed * 7 + 1);
        const float u = ((h & 0xffff) + (h >> 16) + (g & 0xffff) + (g >> 16)) * (1.0f / 65536.0f) - 2.0f;
        const float v = u * scale * 1.7320508f;
        out[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}
__global__ void ref_kernel(const unsigned short* X, long long ldx, const unsigned short* Q, int qstride, const float* n2,
                           long long n, int nq, int K, int l2, float* out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * nq) return;
    const long long r = idx / nq;
    const int q = (int)(idx % nq);
    float acc = 0.f;
    for (int j = 0; j < K; ++j)
        acc = fmaf(__uint_as_float((unsigned)X[r * ldx + j] << 16), __uint_as_float((unsigned)Q[(size_t)q * qstride + j] << 16), acc);
    out[idx] = l2 ? -(n2[r] + acc) : acc;      // Q holds -2q for l2
}
__global__ void norms_kernel(const unsigned short* X, long long ldx, long long n, int K, float* n2) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float a = 0.f;
    for (int j = 0; j < K; ++j) { const float v = __uint_as_float((unsigned)X[r * ldx + j] << 16); a = fmaf(v, v, a); }
    n2[r] = a;
}

static int g_shape = 32, g_tiled = 1, g_kernel = 8;       // g_kernel 4: the four-wave form (gemm_pp4.h); 3: 384 x 256 tiles (gemm_pp384.h)
static long long tile_rows() { return g_kernel == 3 ? 384 : 256; }
template <bool L2M, bool DENSE, int SHAPE, bool TILED>
static void launch_v(const GemmPPParams& p, int gx, int qtiles) {
    const int lds = L2M ? PP_LDS_L2 : PP_LDS_IP;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp_kernel<1, L2M, DENSE, SHAPE, TILED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
    hipLaunchKernelGGL((gemm_pp_kernel<1, L2M, DENSE, SHAPE, TILED>), dim3(gx, qtiles), dim3(512), lds, 0, p);
    CK(hipGetLastError());
}
template <bool L2M, bool DENSE, bool TILED>
static void launch_4(const GemmPPParams& p, int gx, int qtiles) {
    const int lds = L2M ? PP_LDS_L2 : PP_LDS_IP;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp4_kernel<1, L2M, DENSE, TILED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
    hipLaunchKernelGGL((gemm_pp4_kernel<1, L2M, DENSE, TILED>), dim3(gx, qtiles), dim3(256), lds, 0, p);
    CK(hipGetLastError());
}
static int g_var = 0;
template <bool L2M, bool DENSE, bool TILED, int VAR>
static void launch_384v(const GemmPPParams& p, int gx, int qtiles) {
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp384_kernel<1, L2M, DENSE, TILED, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS)); once = true; }
    hipLaunchKernelGGL((gemm_pp384_kernel<1, L2M, DENSE, TILED, VAR>), dim3(gx, qtiles), dim3(512), PW_LDS, 0, p);
    CK(hipGetLastError());
}
template <bool L2M, bool DENSE>
static void launch_384(const GemmPPParams& p, int gx, int qtiles) {
    if (g_tiled) { if (g_var & 2) launch_384v<L2M, DENSE, true, 2>(p, gx, qtiles); else launch_384v<L2M, DENSE, true, 0>(p, gx, qtiles); return; }
    switch (g_var) {
        case 1: launch_384v<L2M, DENSE, false, 1>(p, gx, qtiles); break;
        case 2: launch_384v<L2M, DENSE, false, 2>(p, gx, qtiles); break;
        case 3: launch_384v<L2M, DENSE, false, 3>(p, gx, qtiles); break;
        default: launch_384v<L2M, DENSE, false, 0>(p, gx, qtiles); break;
    }
}
template <bool L2M, bool DENSE>
static void launch_d(const GemmPPParams& p, int gx, int qtiles) {
    if (g_kernel == 3) { launch_384<L2M, DENSE>(p, gx, qtiles); return; }
    if (g_kernel == 4) { if (g_tiled) launch_4<L2M, DENSE, true>(p, gx, qtiles); else launch_4<L2M, DENSE, false>(p, gx, qtiles); return; }
    if (g_shape == 32) { if (g_tiled) launch_v<L2M, DENSE, 32, true>(p, gx, qtiles); else launch_v<L2M, DENSE, 32, false>(p, gx, qtiles); }
    else { if (g_tiled) launch_v<L2M, DENSE, 16, true>(p, gx, qtiles); else launch_v<L2M, DENSE, 16, false>(p, gx, qtiles); }
}
template <bool L2M>
static void launch(const GemmPPParams& p, int gx, int qtiles) {
    if (p.dense) launch_d<L2M, true>(p, gx, qtiles); else launch_d<L2M, false>(p, gx, qtiles);
}
// row-major [n][ld] -> tiled image (rows padded to whole tiles with zeros)
__global__ void tile_kernel(const unsigned short* X, long long ld, long long n, int K, int nks, long long ntile_rows, unsigned short* out, int t384) {
    const long long total = ntile_rows * nks * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / (nks * 32);
        const int col = (int)(i % (nks * 32));
        out[t384 ? pw_tiled_index(row, col, nks) : pp_tiled_index(row, col, nks)] = (row < n && col < K) ? X[row * ld + col] : (unsigned short)0;
    }
}
static unsigned short* make_tiled(const unsigned short* X, long long ld, long long n, int K, int nks, int t384 = 0) {
    const long long tr = t384 ? 384 : 256;
    const long long rows = (n + tr - 1) / tr * tr;
    unsigned short* out;
    CK(hipMalloc(&out, (size_t)rows * nks * 32 * 2));
    tile_kernel<<<4096, 256>>>(X, ld, n, K, nks, rows, out, t384);
    CK(hipGetLastError());
    return out;
}

static int run_check(long long n, int nq, int K, int l2, long long tile_stride, int gx_force) {
    if (g_kernel == 4 && (((K + 31) / 32) & 1)) { printf("check K=%d skipped: the four-wave kernel takes an even number of k-steps\n", K); return 0; }
    const long long ldx = K;
    const int qstride = (K + 31) / 32 * 32;
    unsigned short *X, *Q; float *n2, *thr, *dense, *ref, *cs; long long* ct; int *cnt, *lost;
    const int cap = 4096;
    CK(hipMalloc(&X, (size_t)n * ldx * 2 + 256)); CK(hipMalloc(&Q, (size_t)nq * qstride * 2));
    CK(hipMalloc(&n2, n * 4)); CK(hipMalloc(&thr, nq * 4)); CK(hipMalloc(&dense, (size_t)n * nq * 4)); CK(hipMalloc(&ref, (size_t)n * nq * 4));
    CK(hipMalloc(&cs, (size_t)nq * cap * 4)); CK(hipMalloc(&ct, (size_t)nq * cap * 8)); CK(hipMalloc(&cnt, nq * 4)); CK(hipMalloc(&lost, nq * 4));
    fill_bf16<<<1024, 256>>>(X, n * ldx, 1, 1.0f);
    fill_bf16<<<256, 256>>>(Q, (long long)nq * qstride, 2, l2 ? -2.0f : 1.0f);
    norms_kernel<<<(unsigned)((n + 255) / 256), 256>>>(X, ldx, n, K, n2);
    ref_kernel<<<(unsigned)((n * nq + 255) / 256), 256>>>(X, ldx, Q, qstride, n2, n, nq, K, l2, ref);
    CK(hipMemset(dense, 0xff, (size_t)n * nq * 4));
    std::vector<float> hthr(nq, -1e30f);
    CK(hipMemcpy(thr, hthr.data(), nq * 4, hipMemcpyHostToDevice));
    GemmPPParams p;
    memset(&p, 0, sizeof p);
    unsigned short* Xt = make_tiled(X, ldx, n, K, qstride / 32, g_kernel == 3);
    unsigned short* Qt = make_tiled(Q, qstride, nq, K, qstride / 32);
    p.X = g_tiled ? Xt : X; p.n2 = n2; p.Q = Qt; p.thr = thr; p.n = n; p.ldx = ldx; p.nq = nq; p.qstride = qstride; p.nks = qstride / 32;
    p.tile_stride = tile_stride;
    p.ntiles = ((n + tile_rows() - 1) / tile_rows() + tile_stride - 1) / tile_stride;
    p.cand_s = cs; p.cand_t = ct; p.cnt = cnt; p.lost = lost; p.cap = cap; p.dense = dense;
    const int qtiles = (nq + 255) / 256;
    int gx = gx_force > 0 ? gx_force : 8;
    if (gx > p.ntiles) gx = (int)p.ntiles;
    if (l2) launch<true>(p, gx, qtiles); else launch<false>(p, gx, qtiles);
    CK(hipDeviceSynchronize());
    std::vector<float> hd((size_t)n * nq), hr((size_t)n * nq);
    CK(hipMemcpy(hd.data(), dense, hd.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0; long long bad = 0, visited = 0;
    for (long long r = 0; r < n; ++r) {
        const bool vis = ((r / tile_rows()) % tile_stride) == 0;
        for (int q = 0; q < nq; ++q) {
            const float a = hd[(size_t)r * nq + q], b = hr[(size_t)r * nq + q];
            if (!vis) continue;
            ++visited;
            const double e = fabs((double)a - (double)b);
            if (!(e <= 2e-3 * (l2 ? 4.0 * K : (double)K) * 1e-1 + 1e-3)) { if (bad < 5) printf("  mismatch row %lld q %d: got %g want %g\n", r, q, a, b); ++bad; }
            if (e > maxerr) maxerr = e;
        }
    }
    // candidates: threshold at a value that lets about 1 % through
    std::vector<float> sorted(hr.begin(), hr.begin() + std::min<size_t>(hr.size(), 100000));
    std::sort(sorted.begin(), sorted.end());
    const float t = sorted[(size_t)(sorted.size() * 0.99)];
    for (auto& v : hthr) v = t;
    CK(hipMemcpy(thr, hthr.data(), nq * 4, hipMemcpyHostToDevice));
    CK(hipMemset(cnt, 0, nq * 4)); CK(hipMemset(lost, 0, nq * 4));
    p.dense = nullptr;
    if (l2) launch<true>(p, gx, qtiles); else launch<false>(p, gx, qtiles);
    CK(hipDeviceSynchronize());
    std::vector<int> hcnt(nq), hlost(nq);
    CK(hipMemcpy(hlost.data(), lost, nq * 4, hipMemcpyDeviceToHost));
    std::vector<float> hcs((size_t)nq * cap); std::vector<long long> hct((size_t)nq * cap);
    CK(hipMemcpy(hcnt.data(), cnt, nq * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hcs.data(), cs, hcs.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hct.data(), ct, hct.size() * 8, hipMemcpyDeviceToHost));
    long long cbad = 0, ctot = 0, nlost = 0;
    for (int q = 0; q < nq; ++q) {
        nlost += hlost[q] != 0;
        long long want = 0;
        for (long long r = 0; r < n; ++r) if (((r / tile_rows()) % tile_stride) == 0 && hd[(size_t)r * nq + q] > t) ++want;
        // a query whose wave queue overflowed is flagged lost (the library sends it to the exact pass): it may hold fewer
        if (hlost[q] ? hcnt[q] > want : want != hcnt[q]) { if (cbad < 5) printf("  query %d: %d candidates, want %lld\n", q, hcnt[q], want); ++cbad; }
        for (int i = 0; i < hcnt[q] && i < cap; ++i) {
            const long long r = hct[(size_t)q * cap + i];
            if (r < 0 || r >= n || hcs[(size_t)q * cap + i] != hd[(size_t)r * nq + q]) { if (cbad < 5) printf("  query %d cand %d: row %lld key %g\n", q, i, r, hcs[(size_t)q * cap + i]); ++cbad; }
        }
        ctot += hcnt[q];
    }
    printf("check n=%lld nq=%d K=%d l2=%d stride=%lld gx=%d: %lld keys, max |err| %.3g, %lld bad keys, %lld candidates, %lld queries flagged lost, %lld bad candidate lists -> %s\n",
           n, nq, K, l2, tile_stride, gx, visited, maxerr, bad, ctot, nlost, cbad, (bad || cbad) ? "FAIL" : "ok");
    (void)0;
    return (bad || cbad) ? 1 : 0;
}

static void run_time(long long n, int nq, int K, int reps, int l2) {
    const long long ldx = K;
    const int qstride = (K + 31) / 32 * 32;
    unsigned short *X, *Q; float *n2, *thr, *cs; long long* ct; int *cnt, *lost;
    const int cap = 8192;
    CK(hipMalloc(&X, (size_t)n * ldx * 2 + 256)); CK(hipMalloc(&Q, (size_t)nq * qstride * 2));
    CK(hipMalloc(&n2, n * 4)); CK(hipMalloc(&thr, nq * 4));
    CK(hipMalloc(&cs, (size_t)nq * cap * 4)); CK(hipMalloc(&ct, (size_t)nq * cap * 8)); CK(hipMalloc(&cnt, nq * 4)); CK(hipMalloc(&lost, nq * 4));
    const float fscale = getenv("PP_SCALE") ? (float)atof(getenv("PP_SCALE")) : 1.0f;     // 0: all-zero operands (clock check)
    fill_bf16<<<4096, 256>>>(X, n * ldx, 1, fscale);
    fill_bf16<<<256, 256>>>(Q, (long long)nq * qstride, 2, (l2 ? -2.0f : 1.0f) * fscale);
    norms_kernel<<<(unsigned)((n + 255) / 256), 256>>>(X, ldx, n, K, n2);
    // threshold about 4.2 sigma of the score distribution: ~ 1e-5 of the rows pass
    const float sigma = sqrtf((float)K) * (getenv("PP_SIGMA") ? (float)atof(getenv("PP_SIGMA")) / 4.2f : 1.0f);
    std::vector<float> hthr(nq, l2 ? -(float)K + 4.2f * sqrtf(6.0f * K) : 4.2f * sigma);
    if (getenv("PP_THR_INF")) for (auto& v : hthr) v = 3e38f;
    CK(hipMemcpy(thr, hthr.data(), nq * 4, hipMemcpyHostToDevice));
    GemmPPParams p;
    memset(&p, 0, sizeof p);
    unsigned short* Xt = make_tiled(X, ldx, n, K, qstride / 32, g_kernel == 3);
    unsigned short* Qt = make_tiled(Q, qstride, nq, K, qstride / 32);
    p.X = g_tiled ? Xt : X; p.n2 = n2; p.Q = Qt; p.thr = thr; p.n = n; p.ldx = ldx; p.nq = nq; p.qstride = qstride; p.nks = qstride / 32;
    p.tile_stride = 1; p.ntiles = (n + tile_rows() - 1) / tile_rows();
    p.cand_s = cs; p.cand_t = ct; p.cnt = cnt; p.lost = lost; p.cap = cap; p.dense = nullptr;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int qtiles = (nq + 255) / 256;
    int gx = prop.multiProcessorCount / qtiles / 8 * 8;
    if (gx < 8) gx = 8;
    if (getenv("PP_GX")) gx = atoi(getenv("PP_GX"));
    if (gx > p.ntiles) gx = (int)p.ntiles;
    if (getenv("PP_STRIDE0")) p.tile_stride = 0;      // every tile reads the first 256 rows: the row operand comes from cache
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) { CK(hipMemset(cnt, 0, nq * 4)); if (l2) launch<true>(p, gx, qtiles); else launch<false>(p, gx, qtiles); }
    CK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(cnt, 0, nq * 4, 0));
        CK(hipEventRecord(e0));
        if (l2) launch<true>(p, gx, qtiles); else launch<false>(p, gx, qtiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        sum += ms;
    }
    std::vector<int> hcnt(nq);
    CK(hipMemcpy(hcnt.data(), cnt, nq * 4, hipMemcpyDeviceToHost));
    long long ctot = 0; for (int v : hcnt) ctot += v;
    const double flops = 2.0 * n * K * nq;
    printf("shape %d tiled %d kernel %d ", g_shape, g_tiled, g_kernel); printf("time n=%lld nq=%d K=%d l2=%d grid %dx%d: best %.3f ms, mean %.3f ms -> %.1f TFLOP/s (mean %.1f), %.2f TB/s of rows; %lld candidates\n",
           n, nq, K, l2, gx, qtiles, best, sum / reps, flops / best / 1e9, flops / (sum / reps) / 1e9, (double)n * K * 2 / best / 1e9, ctot);
#ifdef PP_STAMPS
    {
        std::vector<unsigned> h(4096);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pp_end), 4096 * 4));
        const int nb = gx * qtiles;
        std::vector<double> e(nb);
        for (int i = 0; i < nb; ++i) e[i] = h[i] / 100.0;
        std::vector<unsigned> hc(4096);
        CK(hipMemcpyFromSymbol(hc.data(), HIP_SYMBOL(g_pp_clk), 4096 * 4));
        std::vector<double> ghz(nb);
        for (int i = 0; i < nb; ++i) ghz[i] = h[i] ? hc[i] / (h[i] * 10.0) : 0;      // cycles per 10 ns tick -> GHz
        std::sort(ghz.begin(), ghz.end());
        printf("shader clock over the workgroups' run (GHz): min %.3f median %.3f max %.3f\n", ghz[0], ghz[nb / 2], ghz[nb - 1]);
        std::vector<double> srt(e); std::sort(srt.begin(), srt.end());
        printf("workgroup run times (us): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f\n", srt[0], srt[nb / 10], srt[nb / 2], srt[nb * 9 / 10], srt[nb - 1]);
        for (int x = 0; x < 8; ++x) {
            std::vector<double> g;
            for (int i = 0; i < nb; ++i) if ((i % gx) % 8 == x) g.push_back(e[i]);
            std::sort(g.begin(), g.end());
            printf("  blockIdx.x %% 8 == %d: %zu workgroups, min %.0f median %.0f max %.0f\n", x, g.size(), g[0], g[g.size() / 2], g.back());
        }
    }
#endif
    (void)0;
}

int main(int argc, char** argv) {
    if (getenv("PP_SHAPE")) g_shape = atoi(getenv("PP_SHAPE")) == 16 ? 16 : 32;
    if (getenv("PP_TILED")) g_tiled = atoi(getenv("PP_TILED"));
    if (getenv("PP_KERNEL")) g_kernel = atoi(getenv("PP_KERNEL"));
    if (getenv("PP_VAR")) g_var = atoi(getenv("PP_VAR"));
    if (g_kernel == 3) g_shape = 16;
    if (argc >= 2 && !strcmp(argv[1], "check")) {
        int rc = 0;
        printf("MFMA shape %d tiled %d kernel %d\n", g_shape, g_tiled, g_kernel);
        rc |= run_check(256, 256, 64, 0, 1, 1);
        rc |= run_check(1024, 256, 768, 0, 1, 2);
        rc |= run_check(3000, 300, 768, 0, 1, 8);
        rc |= run_check(5000, 512, 96, 0, 1, 8);
        rc |= run_check(2049, 70, 32, 0, 1, 3);
        rc |= run_check(6000, 256, 128, 0, 3, 4);
        rc |= run_check(3000, 300, 768, 1, 1, 8);
        rc |= run_check(2049, 70, 32, 1, 1, 3);
        rc |= run_check(2049, 70, 64, 1, 1, 3);
        rc |= run_check(777, 256, 128, 0, 1, 1);
        rc |= run_check(6000, 256, 192, 1, 3, 4);
        printf(rc ? "CHECK FAILED\n" : "CHECK OK\n");
        return rc;
    }
    const long long n = argc > 2 ? atoll(argv[2]) : 2560000;
    const int nq = argc > 3 ? atoi(argv[3]) : 1024;
    const int K = argc > 4 ? atoi(argv[4]) : 768;
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    const int l2 = argc > 6 ? atoi(argv[6]) : 0;
    run_time(n, nq, K, reps, l2);
    return 0;
}
