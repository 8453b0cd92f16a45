/*
 * oracle_c.c -- plain C restatement of the search specification in oracle/oracle.py.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/ (and optionally bench.py's CPU baseline) through
 * oracle/oracle_c.py; never linked or loaded by the product library.
 *
 * PARITY UNPINNED for the search arithmetic, for the reason given in oracle/oracle.py: the
 * reference (/root/reference/app.py:379-409, :391) forwards the query to an external engine and
 * holds no scoring code or expected results.  The interface facts it does pin (best-first
 * order, <= k hits, caller-supplied signed 64-bit ids: app.py:315-325, 392-406) are kept.
 *
 * Canonical score: 64 interleaved fp64 partial sums (element j into partial j % 64, increasing
 * j), combined by the fixed binary tree 32,16,...,1, rounded once to fp32.  Build with
 * -ffp-contract=off: multiply and add are rounded separately.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LANES 64
enum { METRIC_L2 = 1, METRIC_IP = 2, METRIC_COSINE = 3 };

static double tree(double *acc) {
    for (int s = LANES / 2; s >= 1; s /= 2)
        for (int p = 0; p < s; ++p) acc[p] = acc[p] + acc[p + s];
    return acc[0];
}

/* fp64 canonical sum of x[j]*q[j] (l2 == 0) or (x[j]-q[j])^2 (l2 != 0) */
static double canon64(const float *x, const float *q, int d, int l2) {
    double acc[LANES];
    for (int p = 0; p < LANES; ++p) acc[p] = 0.0;
    for (int j = 0; j < d; ++j) {
        const double a = (double)x[j], b = (double)q[j];
        if (l2) { const double df = a - b; const double sq = df * df; acc[j % LANES] = acc[j % LANES] + sq; }
        else { const double pr = a * b; acc[j % LANES] = acc[j % LANES] + pr; }
    }
    return tree(acc);
}

/* scores[i] = fp32(canonical score of row i); rows [n][d] fp32 values as stored */
void oc_scores(const float *rows, int64_t n, int d, const float *q, int metric, float *scores) {
    const int l2 = metric == METRIC_L2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) scores[i] = (float)canon64(rows + (size_t)i * d, q, d, l2);
}

/* canonical L2 normalisation of n fp32 rows, in place semantics via out */
void oc_normalize(const float *rows, int64_t n, int d, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *x = rows + (size_t)i * d;
        const double n2 = canon64(x, x, d, 0);
        const double inv = n2 == 0.0 ? 0.0 : 1.0 / sqrt(n2);
        for (int j = 0; j < d; ++j) out[(size_t)i * d + j] = (float)((double)x[j] * inv);
    }
}

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* the synthetic embedding generator (same definition as oracle.synth_rows) */
void oc_synth_rows(uint64_t seed, int64_t row0, int64_t n, int d, float *out) {
    const float scale = 0x1.bb67aep-16f;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(row0 + i + 1));
        for (int j = 0; j < d; ++j) {
            const uint64_t h = mix64(key ^ (0xD1B54A32D192ED03ull * (uint64_t)(j + 1)));
            const int a = (int)(h & 0xffff) + (int)((h >> 16) & 0xffff) + (int)((h >> 32) & 0xffff) + (int)(h >> 48);
            out[(size_t)i * d + j] = (float)(a - 131070) * scale;
        }
    }
}

/* top-k by (score best first, id ascending): simple selection, k small */
typedef struct { float s; int64_t id; } hit_t;
static int better(float s, int64_t id, float rs, int64_t rid, int l2) {
    if (s != rs) return l2 ? s < rs : s > rs;
    return id < rid;
}
int64_t oc_topk(const float *scores, const int64_t *ids, int64_t n, int k, int metric,
                int64_t *out_ids, float *out_scores) {
    const int l2 = metric == METRIC_L2;
    int64_t cnt = 0;
    hit_t *heap = (hit_t *)malloc(sizeof(hit_t) * (size_t)(k > 0 ? k : 1));
    for (int64_t i = 0; i < n; ++i) {
        if (cnt == k && !better(scores[i], ids[i], heap[cnt - 1].s, heap[cnt - 1].id, l2)) continue;
        int64_t pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && better(scores[i], ids[i], heap[pos - 1].s, heap[pos - 1].id, l2)) {
            heap[pos] = heap[pos - 1];
            --pos;
        }
        heap[pos].s = scores[i];
        heap[pos].id = ids[i];
        if (cnt < k) ++cnt;
    }
    for (int64_t i = 0; i < cnt; ++i) { out_ids[i] = heap[i].id; out_scores[i] = heap[i].s; }
    free(heap);
    return cnt;
}
