"""Cross-check against plain numpy, independent of the oracle package: scores from `X @ q` in fp32 and
in fp64, order from np.lexsort.  Bars (the ones the task statement gives): scores within 1e-5
relative of the numpy value (fp32 tables; 16-bit tables are compared on their stored, already rounded
rows, so the same bar holds), ids equal wherever the numpy scores around a rank are further apart
than that tolerance."""
import numpy as np
import pytest

from vectordb_amd import _lib

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _numpy_scores(stored, q, metric, dtype):
    x = stored.astype(dtype)
    qq = q.astype(dtype)
    if metric == "l2":
        # squared distance; the fp32 variant as numpy users write it
        return ((x - qq[None, :]) ** 2).sum(axis=1)
    return x @ qq


def _check_query(hip_ids, hip_scores, ids, stored, q, k, metric):
    for dt in (np.float32, np.float64):
        s = _numpy_scores(stored, q, metric, dt).astype(np.float64)
        key = s if metric == "l2" else -s
        order = np.lexsort((ids, key))                       # best first, ties by id
        top = order[:k + 1]
        tol = RTOL * np.maximum(np.abs(s[top]), 1e-30)
        # scores: within tolerance of numpy's score of the same row
        pos = {int(i): j for j, i in enumerate(ids)}
        for hid, hs in zip(hip_ids, hip_scores):
            ref = s[pos[int(hid)]]
            assert abs(float(hs) - ref) <= RTOL * max(abs(ref), 1e-30) * (4 if dt == np.float32 else 1) + 1e-12, \
                (metric, dt, hid, hs, ref)
        # ids: position j must match wherever numpy's neighbouring scores are clearly apart
        for j in range(k):
            gap_prev = abs(s[top[j]] - s[top[j - 1]]) if j > 0 else np.inf
            gap_next = abs(s[top[j]] - s[top[j + 1]]) if j + 1 < len(top) else np.inf
            if min(gap_prev, gap_next) > 4 * tol[j]:
                assert int(hip_ids[j]) == int(ids[top[j]]), (metric, dt, j, hip_ids, ids[top[:k]])


@pytest.mark.parametrize("metric", ["cosine", "ip", "l2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_config1_shape_against_plain_numpy(engine, table_name, metric, dtype):
    """10 000 x 384 vectors drawn as /root/reference/createVector.py:6-8 draws them (U[0,1) float64, here
    seeded), 100 queries drawn the same way."""
    n, d, k = 10_000, 384, 10
    raw = np.random.default_rng(0).random((n, d))
    queries = np.random.default_rng(1).random((100, d)).astype(np.float32)
    ids = np.random.default_rng(2).permutation(n).astype(np.int64) - n // 2       # negative ids too
    m = {"cosine": _lib.METRIC_COSINE, "ip": _lib.METRIC_IP, "l2": _lib.METRIC_L2}[metric]
    t = {"f32": _lib.F32, "bf16": _lib.BF16, "f16": _lib.F16}[dtype]
    engine.create_table(table_name, d, m, t)
    engine.insert(table_name, raw.astype(np.float32), ids)
    if dtype == "f32":
        # what a numpy user would hold: the fp32 rows, normalised in fp32 for cosine
        stored = raw.astype(np.float32)
        if metric == "cosine":
            stored = stored / np.linalg.norm(stored, axis=1, keepdims=True)
    else:
        stored, got_ids = engine.fetch_rows(table_name, 0, n)      # the rounded rows as the table holds them
        assert np.array_equal(got_ids, ids)
    got_ids, got_sc, got_cnt = engine.search(table_name, queries, k)
    # single queries go through the scan, the whole batch through the batched path: check both
    one_ids, one_sc, _ = engine.search(table_name, queries[:1], k)
    assert np.array_equal(one_ids[0], got_ids[0]) and np.array_equal(one_sc[0], got_sc[0])
    for qi in range(queries.shape[0]):
        assert int(got_cnt[qi]) == k
        q = queries[qi]
        if metric == "cosine":
            q = q / np.linalg.norm(q)
        _check_query(got_ids[qi], got_sc[qi], ids, stored, q, k, metric)


@pytest.mark.parametrize("nq", [1, 300])
def test_million_rows_against_numpy_on_a_sample(engine, table_name, nq):
    """1 M x 768 fp32 cosine (the second BASELINE configuration): numpy scores a 200 000-row sample of
    the stored rows and the rows the search returned.  No sampled row may beat the k-th hit, and every
    returned score is numpy's score of that row."""
    n, d, k = 1_000_000, 768, 10
    engine.create_table(table_name, d, _lib.METRIC_COSINE, _lib.F32)
    engine.insert_synthetic(table_name, n, seed=41, first_row=0)
    a = 333_000
    sample, sample_ids = engine.fetch_rows(table_name, a, 200_000)
    assert np.array_equal(sample_ids, np.arange(a, a + 200_000))
    rng = np.random.default_rng(3)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    queries[0] = sample[1234] * 3.0                         # a planted neighbour inside the sample
    got_ids, got_sc, got_cnt = engine.search(table_name, queries, k)
    assert int(got_ids[0, 0]) == a + 1234
    for qi in range(0, nq, max(1, nq // 8)):
        q = queries[qi] / np.linalg.norm(queries[qi])
        assert int(got_cnt[qi]) == k
        s32 = sample @ q
        s64 = sample.astype(np.float64) @ q.astype(np.float64)
        kth = float(got_sc[qi, k - 1])
        inlist = np.isin(sample_ids, got_ids[qi])
        tol = RTOL * max(abs(kth), 1e-30)
        assert not np.any(s64[~inlist] > kth + 4 * tol), "a sampled row beats the k-th hit"
        assert not np.any(s32[~inlist] > kth + 16 * tol)
        for j in range(k):
            row, _ = engine.fetch_rows(table_name, int(got_ids[qi, j]), 1)
            ref = float(row[0].astype(np.float64) @ q.astype(np.float64))
            assert abs(float(got_sc[qi, j]) - ref) <= RTOL * abs(ref) + 1e-12
        assert np.all(np.diff(got_sc[qi]) <= 0)


@pytest.mark.parametrize("n,d,dtype,metric,k", [
    (1_000_000, 768, "f32", "cosine", 10),        # second BASELINE configuration
    (10_000_000, 768, "f32", "cosine", 10),       # the headline configuration
    (12_500_000, 128, "bf16", "ip", 100),         # one shard of 8 of the fourth configuration
    (10_000_000, 768, "f16", "l2", 10),           # rows and metric of the fifth configuration
])
def test_whole_table_scored_by_numpy(engine, table_name, n, d, dtype, metric, k):
    """BASELINE's shapes at full size: numpy scores EVERY stored row in fp32 (`X @ q`; for L2
    `|x|^2 - 2 X @ q + |q|^2`), re-scores its best 256 + 8k rows in fp64 from the definition, orders them
    with np.lexsort, and the search has to return those ids (wherever numpy's neighbouring scores are
    further apart than the tolerance) with scores within 1e-5 relative.  16-bit tables are compared on
    the rows as stored (fetched back), so the same bar holds for them."""
    m = {"cosine": _lib.METRIC_COSINE, "ip": _lib.METRIC_IP, "l2": _lib.METRIC_L2}[metric]
    t = {"f32": _lib.F32, "bf16": _lib.BF16, "f16": _lib.F16}[dtype]
    engine.create_table(table_name, d, m, t)
    engine.reserve(table_name, n)
    engine.insert_synthetic(table_name, n, seed=43, first_row=0)
    x = np.empty((n, d), dtype=np.float32)
    step = 500_000
    for a in range(0, n, step):
        rows, got = engine.fetch_rows(table_name, a, min(step, n - a))
        assert got[0] == a and got[-1] == a + rows.shape[0] - 1
        x[a:a + rows.shape[0]] = rows
    ids = np.arange(n, dtype=np.int64)
    rng = np.random.default_rng(5)
    queries = rng.standard_normal((4, d)).astype(np.float32)
    queries[1] = x[n // 3] * 2.0 + 0.02 * queries[1]             # a query with a near neighbour
    rep = 65
    batch_ids, batch_sc, _ = engine.search(table_name, np.repeat(queries, rep, axis=0), k)   # batched path, 260 queries
    n2 = np.einsum("ij,ij->i", x, x) if metric == "l2" else None
    ncand = 256 + 8 * k
    for qi in range(queries.shape[0]):
        one_ids, one_sc, cnt = engine.search(table_name, queries[qi:qi + 1], k)              # scan
        assert int(cnt[0]) == k
        assert np.array_equal(one_ids[0], batch_ids[qi * rep]) and np.array_equal(one_sc[0], batch_sc[qi * rep])
        q = queries[qi] / np.linalg.norm(queries[qi]) if metric == "cosine" else queries[qi]
        q64 = q.astype(np.float64)
        s32 = x @ q
        if metric == "l2":
            s32 = n2 - 2.0 * s32 + np.float32(q @ q)
            key32 = s32                                   # smaller is better
        else:
            key32 = -s32
        cand = np.argpartition(key32, ncand)[:ncand]
        # rows outside the candidates cannot reach the top k: numpy's own fp32 error is far below this gap
        kc = np.sort(key32[cand])
        assert kc[-1] - kc[k] > 2e-4 * np.abs(kc).max()
        xc = x[cand].astype(np.float64)
        s64 = ((xc - q64[None, :]) ** 2).sum(axis=1) if metric == "l2" else xc @ q64
        order = np.lexsort((ids[cand], s64 if metric == "l2" else -s64))
        top = order[:k + 1]
        for j in range(k):
            ref = s64[top[j]]
            tol = RTOL * abs(ref)
            gap_prev = abs(ref - s64[top[j - 1]]) if j > 0 else np.inf
            gap_next = abs(ref - s64[top[j + 1]])
            if min(gap_prev, gap_next) > 4 * tol:
                assert int(one_ids[0, j]) == int(ids[cand[top[j]]]), (qi, j, one_ids[0], ids[cand[top[:k]]])
            hit = np.nonzero(ids[cand] == one_ids[0, j])[0]
            assert hit.size == 1, "a returned row is not among numpy's best"
            assert abs(float(one_sc[0, j]) - s64[hit[0]]) <= RTOL * abs(s64[hit[0]]) + 1e-12
    del x
