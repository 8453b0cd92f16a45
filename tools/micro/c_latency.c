/* Blocking vdb_search latency through the C ABI itself (no binding in between): one query per call over
 * 10 000 x 384 and 100 000 x 384 fp32 rows, IP, top-10.
 *   gcc -O2 -I include -o tools/micro/c_latency tools/micro/c_latency.c -L vectordb_amd -lvdb -Wl,-rpath,'$ORIGIN/../../vectordb_amd' -lm
 *   ./tools/micro/c_latency [0|1] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "vdb.h"

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

int main(int argc, char** argv) {
    if (vdb_init(0) != 0) { fprintf(stderr, "vdb_init: %s\n", vdb_last_error()); return 1; }
    const int64_t shapes[2][2] = {{10000, 384}, {100000, 384}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;          /* 0 or 1: that shape alone */
    for (int s = 0; s < 2; ++s) {
        if (only >= 0 && s != only) continue;
        const int64_t rows = shapes[s][0], dim = shapes[s][1];
        const int k = 10, calls = 512;
        if (vdb_create_table("clat", dim, VDB_METRIC_IP, VDB_F32) != 0 || vdb_insert_synthetic("clat", rows, 0, 0) != 0) {
            fprintf(stderr, "setup: %s\n", vdb_last_error());
            return 1;
        }
        float* q = (float*)malloc(sizeof(float) * dim * calls);
        unsigned x = 12345u;
        for (int64_t i = 0; i < dim * calls; ++i) { x = x * 1664525u + 1013904223u; q[i] = (float)(x >> 8) / 16777216.0f; }
        int64_t ids[10];
        float sc[10];
        int32_t cnt[1];
        double lat[512];
        int n = 0;
        for (int i = 0; i < calls; ++i) {
            const double t0 = now_us();
            if (vdb_search("clat", 1, q + (size_t)i * dim, k, ids, sc, cnt) != 0) { fprintf(stderr, "search: %s\n", vdb_last_error()); return 1; }
            const double t1 = now_us();
            if (i >= 50) lat[n++] = t1 - t0;
        }
        qsort(lat, n, sizeof(double), cmp);
        printf("%7lld x %3lld through the C ABI: p50 %.1f us  p10 %.1f  p90 %.1f  p99 %.1f  (count %d, best id %lld)\n",
               (long long)rows, (long long)dim, lat[n / 2], lat[n / 10], lat[n * 9 / 10], lat[n * 99 / 100], (int)cnt[0], (long long)ids[0]);
        free(q);
        vdb_drop_table("clat");
    }
    return 0;
}
