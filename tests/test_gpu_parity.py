"""GPU parity tests: every call goes through the C ABI of libvdb.so and is compared with the
numpy oracle on the same inputs.  Bar: ids identical, scores bit-identical."""
import json
import os
import threading

import numpy as np
import pytest

import cases
from oracle import oracle as O
from vectordb_amd import _lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(engine, name, stored, ids, queries, k, metric):
    got_ids, got_sc, got_cnt = engine.search(name, queries, k)
    want = cases.oracle_search(stored, ids, queries, k, metric)
    for qi, (wi, ws) in enumerate(want):
        c = int(got_cnt[qi])
        assert c == len(wi), "query %d: count %d != %d" % (qi, c, len(wi))
        assert np.array_equal(got_ids[qi, :c], wi), "query %d ids\n got %s\nwant %s" % (qi, got_ids[qi, :c], wi)
        assert np.array_equal(got_sc[qi, :c].view(np.uint32), ws.view(np.uint32)), \
            "query %d scores\n got %s\nwant %s" % (qi, got_sc[qi, :c], ws)
        assert np.all(got_ids[qi, c:] == -1)


def _load(engine, name, rows, ids, metric, dtype):
    engine.create_table(name, rows.shape[1], metric, dtype)
    engine.insert(name, rows, ids)
    assert engine.count(name) == rows.shape[0]
    return O.prepare_store(rows, metric, dtype)


# ------------------------------------------------------------------------------- store
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("metric", ["ip", "cosine"])
def test_stored_rows_match_oracle_bitwise(engine, table_name, metric, dtype):
    # many rows of several lengths: covers the device's fp64 sqrt and division (normalisation)
    d = 37
    rows = cases.make_rows("synth", 31, 20000, d)
    rows[5] = 0.0
    ids = cases.make_ids("hash", 31, rows.shape[0])
    stored = _load(engine, table_name, rows, ids, cases.METRICS[metric], cases.DTYPES[dtype])
    got_rows, got_ids = engine.fetch_rows(table_name, 0, rows.shape[0])
    assert np.array_equal(got_ids, ids)
    assert np.array_equal(got_rows.view(np.uint32), stored.view(np.uint32))


def test_synthetic_rows_match_oracle_bitwise(engine, table_name):
    import torch
    engine.create_table(table_name, 96, _lib.METRIC_IP, _lib.F32)
    engine.insert_synthetic(table_name, 3000, seed=5, first_row=0)
    engine.insert_synthetic(table_name, 500, seed=5, first_row=3000)
    got, ids = engine.fetch_rows(table_name, 0, 3500)
    want = O.synth_rows(5, 0, 3500, 96)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(ids, np.arange(3500))
    buf = torch.empty((70, 33), dtype=torch.float32, device="cuda")
    engine.synth_fill_dev(buf.data_ptr(), 70, 33, seed=9, first_row=11,
                          stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), O.synth_rows(9, 11, 70, 33).view(np.uint32))


def test_append_of_more_rows_than_one_launch_holds(engine, table_name):
    """70 M rows appended by one call: one wave per row would need a grid of more than 2^32
    work-items, which is not launched whole, so the append runs in launches of 2^24 rows.  Rows on
    both sides of every launch boundary, the first and the last row hold their own values and ids,
    and a query that is a copy of the last row finds it."""
    n, d = 70_000_000, 4
    engine.create_table(table_name, d, _lib.METRIC_L2, _lib.F32)
    engine.reserve(table_name, n)
    engine.insert_synthetic(table_name, n, seed=3, first_row=0)
    assert engine.count(table_name) == n
    probes = [0, (1 << 24) - 1, 1 << 24, (1 << 25) - 1, 1 << 25, 3 << 24, (1 << 26) - 1, 1 << 26, n - 2]
    for r in probes:
        got, ids = engine.fetch_rows(table_name, r, 2)
        assert np.array_equal(got.view(np.uint32), O.synth_rows(3, r, 2, d).view(np.uint32)), r
        assert ids.tolist() == [r, r + 1]
    q = O.synth_rows(3, n - 1, 1, d)
    ids, sc, cnt = engine.search(table_name, q, 1)
    assert cnt[0] == 1 and sc[0, 0] == 0.0
    assert np.array_equal(O.synth_rows(3, int(ids[0, 0]), 1, d), q)      # a row with exactly these values


def test_synthetic_cosine_table_is_normalised_like_the_oracle(engine, table_name):
    engine.create_table(table_name, 768, _lib.METRIC_COSINE, _lib.F32)
    engine.insert_synthetic(table_name, 2000, seed=0, first_row=100)
    got, ids = engine.fetch_rows(table_name, 0, 2000)
    want = O.normalize_rows(O.synth_rows(0, 100, 2000, 768))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert ids[0] == 100 and ids[-1] == 2099


# ------------------------------------------------------------------------------- search
@pytest.mark.parametrize("d", [1, 8, 100, 128, 384, 768, 1000, 1536, 2500])
@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
def test_search_parity_fp32_dims(engine, table_name, metric, d):
    n = 3000
    rows = cases.make_rows("synth", d, n, d)
    ids = cases.make_ids("hash", d, n)
    q = cases.make_queries("synth", d, 3, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, O.F32)
    _check(engine, table_name, stored, ids, q, 10, m)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("d", [8, 128, 300, 384, 520, 768, 4100])
@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
def test_search_parity_16bit(engine, table_name, metric, d, dtype):
    n = 2000
    rows = cases.make_rows("synth", d + 1, n, d)
    ids = cases.make_ids("index", 0, n)
    q = cases.make_queries("synth", d + 1, 2, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, cases.DTYPES[dtype])
    _check(engine, table_name, stored, ids, q, 10, m)


def test_largest_dimension_scan_and_batched(engine, table_name):
    """dim = 16384, the largest the library accepts (64 KiB per fp32 row): three queries through the
    batched path (512 k-steps per tile), two of them again as single scans; a 16-bit table of the same
    dimension through the scan.  Rows are generated on the device and, for the oracle, by the C
    restatement of the same generator."""
    from oracle import oracle_c as OC
    d, n = 16384, 65_600
    engine.create_table(table_name, d, _lib.METRIC_L2, _lib.F32)
    engine.insert_synthetic(table_name, n, seed=7, first_row=0)
    stored = O.prepare_store(OC.synth_rows(7, 0, n, d), O.L2, O.F32)
    ids = np.arange(n, dtype=np.int64)
    q = OC.synth_rows(8, 0, 3, d)
    q[2] = stored[40_000]
    before = engine.stats()[2]
    _check(engine, table_name, stored, ids, q, 10, O.L2)
    assert engine.stats()[2] - before == 3
    engine.set_option("gemm_path", 0)
    try:
        _check(engine, table_name, stored, ids, q[1:], 10, O.L2)
    finally:
        engine.set_option("gemm_path", 1)
    engine.drop_table(table_name)
    n16 = 3_000
    engine.create_table(table_name, d, _lib.METRIC_COSINE, _lib.BF16)
    engine.insert_synthetic(table_name, n16, seed=9, first_row=0)
    stored16 = O.prepare_store(OC.synth_rows(9, 0, n16, d), O.COSINE, O.BF16)
    _check(engine, table_name, stored16, np.arange(n16, dtype=np.int64), q[:2], 10, O.COSINE)


@pytest.mark.parametrize("n", [1, 2, 9, 31, 32, 33, 63, 64, 65, 1000, 4097])
@pytest.mark.parametrize("k", [1, 10, 100])
def test_search_parity_sizes(engine, table_name, n, k):
    d = 48
    rows = cases.make_rows("synth", n, n, d)
    ids = cases.make_ids("hash", n, n)
    q = cases.make_queries("synth", n, 2, d, rows)
    stored = _load(engine, table_name, rows, ids, O.COSINE, O.F32)
    _check(engine, table_name, stored, ids, q, k, O.COSINE)


def test_search_parity_config1_shape(engine, table_name):
    # 10k x 384 U[0,1) vectors as createVector.py draws them, hashed ids, cosine top-10
    rows = cases.make_rows("uniform", 0, 10000, 384)
    ids = np.array([O.hash_vector(r.tolist()) for r in rows[:200]] + list(cases.make_ids("hash", 0, 10000 - 200)),
                   dtype=np.int64)
    q = cases.make_queries("uniform", 0, 20, 384, rows)
    stored = _load(engine, table_name, rows, ids, O.COSINE, O.F32)
    _check(engine, table_name, stored, ids, q, 10, O.COSINE)


def test_search_parity_large_k(engine, table_name):
    rows = cases.make_rows("synth", 77, 6000, 64)
    ids = cases.make_ids("hash", 77, 6000)
    q = cases.make_queries("synth", 77, 2, 64, rows)
    stored = _load(engine, table_name, rows, ids, O.IP, O.F32)
    for k in (257, 1024):
        _check(engine, table_name, stored, ids, q, k, O.IP)


@pytest.mark.parametrize("case", cases.GOLDEN_CASES, ids=lambda c: "-".join(map(str, c)))
def test_golden_fixtures(engine, table_name, case):
    golden = json.load(open(os.path.join(HERE, "golden", "search_golden.json")))
    g = next(x for x in golden if tuple(x["case"]) == case)
    rk, ik, seed, n, d, metric, dtype, k, nq = case
    rows = cases.make_rows(rk, seed, n, d)
    ids = cases.make_ids(ik, seed, n)
    q = cases.make_queries(rk, seed, nq, d, rows)
    _load(engine, table_name, rows, ids, cases.METRICS[metric], cases.DTYPES[dtype])
    got_ids, got_sc, got_cnt = engine.search(table_name, q, k)
    for qi in range(nq):
        c = int(got_cnt[qi])
        assert [int(x) for x in got_ids[qi, :c]] == g["ids"][qi]
        assert [int(x) for x in got_sc[qi, :c].view(np.uint32)] == g["score_bits"][qi]


# ------------------------------------------------------------------------------- ties
@pytest.mark.parametrize("metric", ["ip", "l2", "cosine"])
@pytest.mark.parametrize("kind", ["ties", "dups"])
def test_ties_are_ordered_by_signed_id(engine, table_name, kind, metric):
    n, d = 5000, 16
    rows = cases.make_rows(kind, 13, n, d)
    ids = cases.make_ids("hash", 13, n)
    q = cases.make_queries(kind, 13, 3, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, O.F32)
    before = engine.stats()[1]
    _check(engine, table_name, stored, ids, q, 10, m)
    if kind == "dups":
        # 5000 equal scores cannot be certified from 32 candidates: the exact pass must have run
        assert engine.stats()[1] > before


def test_duplicate_ids_and_extreme_ids(engine, table_name):
    d = 24
    rows = cases.make_rows("synth", 3, 400, d)
    ids = cases.make_ids("hash", 3, 400)
    ids[:4] = [np.iinfo(np.int64).max, np.iinfo(np.int64).min, -1, 0]
    rows[10] = rows[11]
    ids[10] = ids[11]                     # the same vector under the same id twice
    q = np.concatenate([rows[10:11], rows[0:1], rows[1:2]])
    stored = _load(engine, table_name, rows, ids, O.IP, O.F32)
    _check(engine, table_name, stored, ids, q, 10, O.IP)


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_exact_pass_keeps_rows_that_repeat_a_score_and_an_id(engine, table_name, metric):
    """The exact pass orders by (score, stored id); rows that agree in both are distinct entries of the
    ranking, as in the fast path and in the oracle (option force_exact sends every query there).  Its lists
    drop a row that ties with their K-th entry, which leaves the K best pairs what they are."""
    m = {"ip": O.IP, "l2": O.L2}[metric]
    d = 24
    rows = cases.make_rows("synth", 5, 3000, d)
    ids = cases.make_ids("hash", 5, 3000)
    for j in range(1, 25):                       # 25 copies of row 100 under one id: more than k, spread over blocks
        rows[100 + 97 * j] = rows[100]
        ids[100 + 97 * j] = ids[100]
    rows[7] = rows[8]
    ids[7] = ids[8]
    q = np.concatenate([rows[100:101], rows[7:8], rows[1:2]])
    stored = _load(engine, table_name, rows, ids, m, O.F32)
    engine.set_option("force_exact", 1)
    try:
        _, f0, _ = engine.stats()
        _check(engine, table_name, stored, ids, q, 10, m)
        _check(engine, table_name, stored, ids, q, 40, m)
        _, f1, _ = engine.stats()
        assert f1 - f0 >= 6
    finally:
        engine.set_option("force_exact", 0)


# ------------------------------------------------------------------------------- edges
def test_empty_table_and_zero_queries(engine, table_name):
    engine.create_table(table_name, 8, _lib.METRIC_IP, _lib.F32)
    ids, sc, cnt = engine.search(table_name, np.zeros((2, 8), np.float32), 10)
    assert list(cnt) == [0, 0] and np.all(ids == -1)
    ids, sc, cnt = engine.search(table_name, np.zeros((0, 8), np.float32), 10)
    assert ids.shape == (0, 10)
    engine.insert(table_name, np.zeros((0, 8), np.float32), np.zeros((0,), np.int64))
    assert engine.count(table_name) == 0


def test_error_codes(engine, table_name):
    from vectordb_amd import VdbError
    engine.create_table(table_name, 8, _lib.METRIC_IP, _lib.F32)
    with pytest.raises(VdbError) as e:
        engine.create_table(table_name, 8, _lib.METRIC_IP, _lib.F32)
    assert e.value.code == _lib.EEXISTS
    with pytest.raises(VdbError) as e:
        engine.search("missing", np.zeros((1, 8), np.float32), 10)
    assert e.value.code == _lib.ENOTFOUND
    with pytest.raises(VdbError) as e:
        engine.search(table_name, np.zeros((1, 8), np.float32), 0)
    assert e.value.code == _lib.EINVAL
    with pytest.raises(VdbError) as e:
        engine.search(table_name, np.zeros((1, 8), np.float32), _lib.MAX_TOPK + 1)
    assert e.value.code == _lib.EINVAL
    with pytest.raises(VdbError) as e:
        engine.create_table("bad", 0, _lib.METRIC_IP, _lib.F32)
    assert e.value.code == _lib.EINVAL
    with pytest.raises(VdbError) as e:
        engine.drop_table("missing")
    assert e.value.code == _lib.ENOTFOUND
    engine.create_index(table_name)      # accepted, nothing to build


def test_append_grows_and_stays_searchable(engine, table_name):
    d = 40
    rows = cases.make_rows("synth", 55, 5000, d)
    ids = cases.make_ids("hash", 55, 5000)
    q = cases.make_queries("synth", 55, 2, d, rows)
    engine.create_table(table_name, d, _lib.METRIC_COSINE, _lib.F32)
    stored = O.prepare_store(rows, O.COSINE)
    done = 0
    for step in (1, 1, 1, 500, 700, 2, 3795):      # one row per call (app.py:325), then bulk
        engine.insert(table_name, rows[done:done + step], ids[done:done + step])
        done += step
        assert engine.count(table_name) == done
        _check(engine, table_name, stored[:done], ids[:done], q, 10, O.COSINE)
    assert done == 5000 and engine.table_info(table_name)["capacity"] >= 5000


def test_concurrent_searches_and_inserts(engine, table_name):
    d = 32
    rows = cases.make_rows("synth", 8, 4000, d)
    ids = cases.make_ids("index", 0, 4000)
    q = cases.make_queries("synth", 8, 1, d, rows)
    engine.create_table(table_name, d, _lib.METRIC_IP, _lib.F32)
    engine.insert(table_name, rows[:1000], ids[:1000])
    stored = O.prepare_store(rows, O.IP)
    errors = []

    def searcher():
        try:
            for _ in range(40):
                gi, gs, gc = engine.search(table_name, q, 5)
                n = engine.count(table_name)
                # the answer must be the exact top-5 of some prefix between the two counts
                sc = O.scores(stored, O.prepare_query(q[0], O.IP), O.IP)
                assert gc[0] == 5
                assert np.array_equal(gs[0].view(np.uint32), sc[gi[0]].view(np.uint32))
                assert np.all(gi[0] < n)
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    def inserter():
        try:
            for i in range(1000, 4000, 100):
                engine.insert(table_name, rows[i:i + 100], ids[i:i + 100])
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    ts = [threading.Thread(target=searcher) for _ in range(4)] + [threading.Thread(target=inserter)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    _check(engine, table_name, stored, ids, q, 10, O.IP)


# ------------------------------------------------------------------------------- device entry
def test_concurrent_single_row_inserts_are_merged_and_each_gets_its_own_verdict(engine, table_name):
    """Sixteen threads insert one row per call (the reference's /insert pattern, app.py:315-325); calls that
    arrive together are appended in one launch.  Every tenth call of one thread carries a NaN: exactly those
    calls fail, every other row is stored once, bit for bit what the oracle stores."""
    d, per, nthreads = 96, 60, 16
    rows = cases.make_rows("synth", 77, nthreads * per, d)
    ids = np.arange(nthreads * per, dtype=np.int64) * 3 - 500
    bad = set(range(5 * per, 6 * per, 10))                        # calls of thread 5
    engine.create_table(table_name, d, O.COSINE, _lib.F32)
    failed, errors = set(), []

    def work(t):
        for j in range(t * per, (t + 1) * per):
            row = rows[j:j + 1].copy()
            if j in bad:
                row[0, 7] = np.nan
            try:
                engine.insert(table_name, row, ids[j:j + 1])
            except _lib.VdbError as e:
                (failed.add(j) if e.code == _lib.EINVAL else errors.append((j, e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors and failed == bad
    n = engine.count(table_name)
    assert n == nthreads * per - len(bad)
    got_rows, got_ids = engine.fetch_rows(table_name, 0, n)
    order = np.argsort(got_ids)
    keep = np.array([j for j in range(nthreads * per) if j not in bad])
    assert np.array_equal(got_ids[order], ids[keep])
    want = O.prepare_store(rows[keep], O.COSINE, _lib.F32)
    assert np.array_equal(got_rows[order].view(np.uint32), want.view(np.uint32))


def test_search_dev_and_shard_merge(engine):
    """Row-sharded search on one device: S logical shards, per-shard vdb_search_dev, then
    vdb_merge_topk_dev -- the same data path the multi-GPU host side drives over RCCL."""
    import torch
    d, n, k, nq, S = 64, 6000, 10, 4, 3
    rows = cases.make_rows("ties", 4, n, d)
    ids = cases.make_ids("hash", 4, n)
    q = cases.make_queries("ties", 4, nq, d, rows)
    stored = O.prepare_store(rows, O.L2)
    st = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).cuda()
    p_ids = torch.empty((S, nq, k), dtype=torch.int64, device="cuda")
    p_sc = torch.empty((S, nq, k), dtype=torch.float32, device="cuda")
    p_cnt = torch.empty((S, nq), dtype=torch.int32, device="cuda")
    names = []
    try:
        for s in range(S):
            name = "shard%d" % s
            names.append(name)
            sl = slice(s * n // S, (s + 1) * n // S)
            engine.create_table(name, d, _lib.METRIC_L2, _lib.F32)
            engine.insert(name, rows[sl], ids[sl])
            engine.search_dev(name, nq, qd.data_ptr(), k, p_ids[s].data_ptr(), p_sc[s].data_ptr(),
                              p_cnt[s].data_ptr(), st)
        o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        o_sc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
        engine.merge_topk_dev(S, nq, k, _lib.METRIC_L2, p_ids.data_ptr(), p_sc.data_ptr(), p_cnt.data_ptr(),
                              o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
        torch.cuda.synchronize()
        want = O.search(stored, ids, q, k, O.L2)
        for qi, (wi, ws) in enumerate(want):
            assert int(o_cnt[qi]) == k
            assert np.array_equal(o_ids[qi].cpu().numpy(), wi)
            assert np.array_equal(o_sc[qi].cpu().numpy().view(np.uint32), ws.view(np.uint32))
    finally:
        for name in names:
            if engine.has_table(name):
                engine.drop_table(name)


# ------------------------------------------------------------------------------- full size
def test_one_million_rows_properties(engine, table_name):
    """BASELINE config 2 (1M x 768 fp32 cosine top-10) through size-independent properties:
    returned scores are the canonical scores of the returned rows (rows regenerated on the
    host), order is (score, id), a planted duplicate of the query comes first with its
    own score, and no row of a 20k-row sample beats the k-th hit."""
    import torch
    n, d, k = 1_000_000, 768, 10
    engine.create_table(table_name, d, _lib.METRIC_COSINE, _lib.F32)
    engine.insert_synthetic(table_name, n, seed=0, first_row=0)
    assert engine.count(table_name) == n
    q = O.synth_rows(1, 0, 3, d)
    q[2] = O.synth_rows(0, 123456, 1, d)[0]            # equals stored row 123456 before normalisation
    ids, sc, cnt = engine.search(table_name, q, k)
    assert list(cnt) == [k] * 3
    assert ids[2, 0] == 123456
    for qi in range(3):
        qp = O.prepare_query(q[qi], O.COSINE)
        rows = np.stack([O.normalize_rows(O.synth_rows(0, int(r), 1, d))[0] for r in ids[qi]])
        want = O.scores(rows, qp, O.COSINE)
        assert np.array_equal(want.view(np.uint32), sc[qi].view(np.uint32))
        order = np.lexsort((ids[qi], -sc[qi]))
        assert np.array_equal(order, np.arange(k))
        sample0 = 400_000 + qi * 20_000
        sample = O.normalize_rows(O.synth_rows(0, sample0, 20_000, d))
        ssc = O.scores(sample, qp, O.COSINE)
        better = np.nonzero(ssc > sc[qi, k - 1])[0] + sample0
        assert set(better.tolist()) <= set(ids[qi].tolist())
    # idempotence, and the per-query scan path gives the same answer as the batched path
    ids2, sc2, _ = engine.search(table_name, q, k)
    assert np.array_equal(ids, ids2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    engine.set_option("gemm_path", 0)
    try:
        ids3, sc3, _ = engine.search(table_name, q, k)
    finally:
        engine.set_option("gemm_path", 1)
    assert np.array_equal(ids, ids3) and np.array_equal(sc.view(np.uint32), sc3.view(np.uint32))
    assert engine.stats()[1] >= 0


# ------------------------------------------------------------------------------- batched (GEMM) path
@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
@pytest.mark.parametrize("d,nq", [(96, 33), (100, 130), (768, 24)])
def test_batched_gemm_path_parity(engine, table_name, metric, d, nq):
    """Batches of at least engine_config.use_blas_threshold queries on an fp32 table of at
    least 65536 rows run the MFMA GEMM with the thresholded epilogue; same bar as the scan path."""
    n = 70_000
    rows = cases.make_rows("synth", 900 + d, n, d)
    ids = cases.make_ids("hash", 900 + d, n)
    q = cases.make_queries("synth", 900 + d, nq, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, O.F32)
    before = engine.stats()
    _check(engine, table_name, stored, ids, q, 10, m)
    after = engine.stats()
    assert after[2] - before[2] == nq, "the batch did not take the GEMM path"


def test_batched_gemm_path_ties_and_ordered_rows(engine, table_name):
    """Adversarial inputs for the bootstrapped thresholds: heavy ties, and rows sorted so that
    the best rows come last (the sample underestimates the threshold and candidate arrays
    overflow): the flagged queries must still come back exact through the second pass."""
    n, d, nq = 80_000, 64, 40
    rows = cases.make_rows("ties", 77, n, d)
    q = cases.make_queries("ties", 77, nq, d, rows)
    order = np.argsort(rows @ q[0], kind="stable")          # ascending score for query 0
    rows = np.ascontiguousarray(rows[order])
    ids = cases.make_ids("hash", 77, n)
    stored = _load(engine, table_name, rows, ids, O.IP, O.F32)
    _check(engine, table_name, stored, ids, q, 10, O.IP)


def test_batched_path_matches_scan_path(engine, table_name):
    rows = cases.make_rows("synth", 5, 100_000, 128)
    ids = cases.make_ids("index", 0, 100_000)
    q = cases.make_queries("synth", 5, 256, 128, rows)
    _load(engine, table_name, rows, ids, O.COSINE, O.F32)
    a = engine.search(table_name, q, 10)
    engine.set_option("gemm_path", 0)
    try:
        b = engine.search(table_name, q, 10)
    finally:
        engine.set_option("gemm_path", 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
@pytest.mark.parametrize("d,nq", [(96, 33), (200, 150), (768, 24), (520, 100)])
def test_batched_gemm_path_16bit_parity(engine, table_name, metric, dtype, d, nq):
    """bf16 / fp16 tables: the batch runs on the 16-bit MFMA with the query rounded once to the
    table's type (default) or split into hi + lo parts (`gemm_split_bf16` / `gemm_split_f16`);
    results must be the oracle's bit for bit either way."""
    n = 70_000
    rows = cases.make_rows("synth", 300 + d, n, d)
    ids = cases.make_ids("hash", 300 + d, n)
    q = cases.make_queries("synth", 300 + d, nq, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, cases.DTYPES[dtype])
    before = engine.stats()
    _check(engine, table_name, stored, ids, q, 10, m)
    after = engine.stats()
    assert after[2] - before[2] == nq, "the batch did not take the GEMM path"
    opt = "gemm_split_f16" if dtype == "f16" else "gemm_split_bf16"
    engine.set_option(opt, 1)
    try:
        _check(engine, table_name, stored, ids, q, 10, m)
    finally:
        engine.set_option(opt, 0)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("threads", [256, 512])
def test_256_query_tile_both_workgroup_shapes(engine, table_name, threads, dtype):
    """More than 128 queries take 256-query tiles, as one workgroup of 8 waves over 256 rows or as
    workgroups of 4 waves over 128 rows (`gemm32s_threads`); same exact results.  150 queries: one
    ragged tile; 70 001 rows: a ragged last row tile of either height."""
    n, d, nq = 70_001, 200, 150
    rows = cases.make_rows("synth", 41, n, d)
    ids = cases.make_ids("hash", 41, n)
    q = cases.make_queries("synth", 41, nq, d, rows)
    stored = _load(engine, table_name, rows, ids, O.L2, cases.DTYPES[dtype])
    engine.set_option("gemm32s_threads", threads)
    try:
        before = engine.stats()[2]
        _check(engine, table_name, stored, ids, q, 10, O.L2)
        assert engine.stats()[2] - before == nq
    finally:
        engine.set_option("gemm32s_threads", 512)


def test_band_wider_than_its_buffer_goes_to_the_exact_pass(engine, table_name):
    """3000 stored copies of one vector: for a query equal to it the re-score band of the batched
    path holds more rows than its buffer (2048), so those queries are answered by the exact pass --
    ten of the copies, ordered by id -- while the other queries of the batch stay on the fast path."""
    n, d, nq = 70_000, 64, 40
    rows = cases.make_rows("synth", 45, n, d)
    rows[10_000:13_000] = rows[10_000]
    ids = cases.make_ids("hash", 45, n)
    q = cases.make_queries("synth", 45, nq, d, rows)
    q[3] = rows[10_000]
    q[17] = rows[10_000]
    for metric in (O.COSINE, O.L2):
        name = table_name + "_%d" % metric
        stored = _load(engine, name, rows, ids, metric, O.F32)
        try:
            before = engine.stats()
            _check(engine, name, stored, ids, q, 10, metric)
            after = engine.stats()
            assert after[2] - before[2] == nq
            assert 2 <= after[1] - before[1] < nq
        finally:
            engine.drop_table(name)


def test_wide_batch_default_kernel_choice(engine, table_name):
    """More than one 256-query tile (nq > 256) on an fp16 table and a row count that is not a
    multiple of the 256-row tile; results stay exact."""
    n, d, nq = 66_001, 64, 300
    rows = cases.make_rows("synth", 43, n, d)
    ids = cases.make_ids("index", 43, n)
    q = cases.make_queries("synth", 43, nq, d, rows)
    stored = _load(engine, table_name, rows, ids, O.IP, O.F16)
    before = engine.stats()[2]
    _check(engine, table_name, stored, ids, q, 10, O.IP)
    assert engine.stats()[2] - before == nq


@pytest.mark.parametrize("ordered", [1, 0])
def test_queries_in_flight_on_several_streams(engine, table_name, ordered):
    """Single queries issued round-robin on three HIP streams (the pattern of concurrent clients):
    every answer equals the oracle's whether the library orders the scans or lets them overlap."""
    import torch
    n, d, k, nq = 50_000, 96, 10, 24
    rows = cases.make_rows("synth", 91, n, d)
    ids = cases.make_ids("hash", 91, n)
    stored = _load(engine, table_name, rows, ids, _lib.METRIC_COSINE, _lib.F32)
    q = cases.make_queries("synth", 91, nq, d, rows)
    qd = torch.from_numpy(q).cuda()
    out_ids = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
    out_sc = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    out_cnt = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    engine.set_option("scan_serialize", ordered)
    try:
        for i in range(nq):
            st = streams[i % 3]
            engine.search_dev(table_name, 1, qd[i:i + 1].data_ptr(), k, out_ids[i:i + 1].data_ptr(),
                              out_sc[i:i + 1].data_ptr(), out_cnt[i:i + 1].data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
    finally:
        engine.set_option("scan_serialize", 1)
    want = O.search(stored, ids, q, k, O.COSINE)
    got_ids, got_sc = out_ids.cpu().numpy(), out_sc.cpu().numpy()
    assert out_cnt.cpu().numpy().tolist() == [k] * nq
    for qi, (wi, ws) in enumerate(want):
        assert np.array_equal(got_ids[qi], wi)
        assert np.array_equal(got_sc[qi].view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("n,d,dtype,metric,k", [
    (100_003, 768, "f32", "cosine", 10),      # three loads per row and lane
    (250_001, 128, "bf16", "ip", 100),        # four rows per wave instruction, long lists
    (77_777, 200, "f16", "l2", 10),           # ragged row length
    (5_000, 1000, "f32", "ip", 10),           # fewer steps than waves: most draws come back empty
    (60_001, 2500, "f32", "l2", 10),          # the generic kernel: three chunks of four rows in flight
    (30_001, 4100, "bf16", "cosine", 10),     # the generic kernel over 16-bit rows: two chunks in flight, ragged last chunk
    (20_000, 3072, "f16", "l2", 10),          # the generic kernel, whole chunks
])
def test_scan_step_counter_parity(engine, table_name, n, d, dtype, metric, k):
    """The waves of a scan workgroup draw their steps from a counter in LDS, or scan fixed shares: every row is
    scored exactly once either way, so both settings return the oracle's answer, bit for bit, whatever the
    table's size leaves for the last step."""
    rows = cases.make_rows("synth", 77, n, d)
    ids = cases.make_ids("hash", 77, n)
    q = cases.make_queries("synth", 77, 3, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, cases.DTYPES[dtype])
    engine.set_option("gemm_path", 0)           # three queries would otherwise go to the batched path
    try:
        _check(engine, table_name, stored, ids, q, k, m)
        dynamic = engine.search(table_name, q, k)
        engine.set_option("scan_dynamic", 0)
        _check(engine, table_name, stored, ids, q, k, m)
        fixed = engine.search(table_name, q, k)
    finally:
        engine.set_option("scan_dynamic", 1)
        engine.set_option("gemm_path", 1)
    for a, b in zip(dynamic, fixed):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("k,metric,dtype,kind", [(150, "ip", "f32", "synth"), (256, "l2", "f32", "synth"),
                                                 (500, "cosine", "bf16", "synth"), (256, "ip", "f32", "ties"),
                                                 (600, "l2", "f16", "synth"), (640, "ip", "f32", "synth"),
                                                 (800, "cosine", "f32", "synth"), (1000, "l2", "f32", "synth"),
                                                 (1024, "ip", "bf16", "ties")])
def test_scan_with_lists_of_its_own_size(engine, table_name, k, metric, dtype, kind):
    """Candidate lists of 160 entries and more: the scan keeps lists of K + 64 entries (more waves per workgroup)
    while the exact pass, which merges lists, keeps its 2K; the tie-heavy cases go through both.  From about
    k = 500 the finalize step takes its bound from the second to sixteenth entries of the lists instead of their heads."""
    n, d = 120_000, 96
    rows = cases.make_rows(kind, 63, n, d)
    ids = cases.make_ids("hash", 63, n)
    q = cases.make_queries(kind, 63, 3, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, cases.DTYPES[dtype])
    engine.set_option("gemm_path", 0)
    try:
        _check(engine, table_name, stored, ids, q, k, m)
    finally:
        engine.set_option("gemm_path", 1)


@pytest.mark.parametrize("metric,dtype,kind,d", [("ip", "f32", "synth", 64), ("l2", "bf16", "synth", 64),
                                                 ("cosine", "f16", "synth", 64), ("ip", "f32", "ties", 64),
                                                 ("cosine", "f32", "synth", 768)])
def test_sampled_floor_scan_parity(engine, table_name, metric, dtype, kind, d):
    """Rows of up to 4 KiB, long candidate lists (k = 100) and enough rows: the scan starts from a floor taken
    from a sample of the rows.  Exact results with it, and the same results without it."""
    n, k = 300_000, 100
    rows = cases.make_rows(kind, 57, n, d)
    ids = cases.make_ids("hash", 57, n)
    q = cases.make_queries(kind, 57, 5, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, cases.DTYPES[dtype])
    engine.set_option("gemm_path", 0)           # five queries would otherwise go to the batched path
    try:
        _check(engine, table_name, stored, ids, q, k, m)
        with_floor = engine.search(table_name, q, k)
        engine.set_option("scan_sample", 0)
        without = engine.search(table_name, q, k)
    finally:
        engine.set_option("scan_sample", 1)
        engine.set_option("gemm_path", 1)
    for a, b in zip(with_floor, without):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("split", [2, 1, 0])
@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
@pytest.mark.parametrize("d,nq", [(100, 130), (768, 70), (36, 64), (20, 31)])
def test_batched_fp32_table_all_fast_pass_modes(engine, table_name, metric, d, nq, split):
    """fp32 tables: the passes run on the 16-bit matrix unit with rows and queries rounded to bf16
    (`gemm32_split` 2, default) or presented as bf16 hi + lo with three products (1), or on the fp32
    MFMA (0).  Same exact results in every mode; dims that are not multiples of 8 exercise the zero
    padding of the 16-bit queries, dims of one or two k-steps the stages past the last tile."""
    n = 70_000
    rows = cases.make_rows("synth", 61, n, d)
    ids = cases.make_ids("hash", 61, n)
    q = cases.make_queries("synth", 61, nq, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, _lib.F32)
    engine.set_option("gemm32_split", split)
    try:
        before = engine.stats()
        _check(engine, table_name, stored, ids, q, 10, m)
        after = engine.stats()
        assert after[2] - before[2] == nq           # took the batched path
    finally:
        engine.set_option("gemm32_split", 2)


@pytest.mark.parametrize("split", [2, 1])
def test_batched_fp32_with_large_magnitudes_and_ties(engine, table_name, split):
    """Rows with a wide range of magnitudes and quantised duplicates: the 16-bit fast passes may send
    queries to the exact pass (the error bound scales with the largest row norm), never change a
    result."""
    n, d, nq = 70_000, 64, 80
    rows = cases.make_rows("ties", 63, n, d)
    rows[::7] *= 1000.0
    rows[1::11] *= 1e-4
    ids = cases.make_ids("negdesc", 63, n)
    q = cases.make_queries("ties", 63, nq, d, rows)
    engine.set_option("gemm32_split", split)
    try:
        for metric in (O.IP, O.L2):
            name = table_name + "_%d" % metric
            stored = _load(engine, name, rows, ids, metric, _lib.F32)
            try:
                _check(engine, name, stored, ids, q, 10, metric)
            finally:
                engine.drop_table(name)
    finally:
        engine.set_option("gemm32_split", 2)


@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
@pytest.mark.parametrize("kind,d", [("synth", 768), ("synth", 64), ("ties", 64), ("synth", 100)])
def test_single_queries_over_the_bf16_mirror_of_an_fp32_table(engine, table_name, metric, kind, d):
    """Option scan_mirror: the scan reads the bf16 mirror (half the bytes) and the candidates are re-scored
    from the fp32 rows; the answers are the oracle's bit for bit, through the certificate or (tie-heavy
    rows) the exact pass.  d = 100 has no mirror row of the query's length and scans the fp32 rows."""
    n = 50_000
    rows = cases.make_rows(kind, 71, n, d)
    if kind == "synth":
        rows[::50] *= 30.0
    ids = cases.make_ids("hash", 71, n)
    q = cases.make_queries(kind, 71, 3, d, rows)
    m = cases.METRICS[metric]
    stored = _load(engine, table_name, rows, ids, m, _lib.F32)
    engine.set_option("scan_mirror", 1)
    engine.set_option("use_blas_threshold", 1000)
    try:
        before = engine.stats()
        _check(engine, table_name, stored, ids, q, 10, m)
        _check(engine, table_name, stored, ids, q[:1], 100, m)     # list too short for the mirror's bound: fp32 scan
        after = engine.stats()
        assert after[2] == before[2]                               # no batched path
        if kind == "synth" and d == 768:
            assert after[1] == before[1]                           # certified without the exact pass
    finally:
        engine.set_option("scan_mirror", 0)
        engine.set_option("use_blas_threshold", 2)


def test_mirror_scan_falls_back_to_the_fp32_rows_before_the_exact_pass(engine, table_name):
    """Four hundred rows along the query's direction whose lengths differ by 5e-5 each: the scores around
    rank 10 are closer together than the rounding of the bf16 mirror (2e-3) but far apart for an fp32 scan.
    The mirror scan cannot certify its answer; the second tier re-scans the fp32 rows for that query and
    can, so no exact pass runs (with the tier switched off it does)."""
    n, d = 50_000, 64
    rng = np.random.default_rng(73)
    u = rng.standard_normal(d).astype(np.float32)
    u /= np.linalg.norm(u)
    rows = (0.1 * cases.make_rows("synth", 73, n, d)).astype(np.float32)
    for j in range(400):
        rows[j * 7] = (np.float32(1.0) + np.float32(j) * np.float32(5e-5)) * u
    ids = cases.make_ids("hash", 73, n)
    q = np.stack([3.0 * u, u + 0.01 * rng.standard_normal(d).astype(np.float32)]).astype(np.float32)
    stored = _load(engine, table_name, rows, ids, O.IP, _lib.F32)
    engine.set_option("scan_mirror", 1)
    engine.set_option("use_blas_threshold", 1000)
    try:
        exact = {}
        for tier2 in (1, 0):
            engine.set_option("scan_mirror_tier2", tier2)
            before = engine.stats()
            _check(engine, table_name, stored, ids, q, 10, O.IP)
            exact[tier2] = engine.stats()[1] - before[1]
        assert exact[1] == 0 and exact[0] == 2, exact
    finally:
        engine.set_option("scan_mirror", 0)
        engine.set_option("scan_mirror_tier2", 1)
        engine.set_option("use_blas_threshold", 2)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("nq", [3, 40, 150])
def test_l2_rows_with_heavy_tailed_norms_stay_on_the_fast_path(engine, table_name, nq, dtype):
    """L2 keys carry the error term of their own row, so one row in a hundred that is a hundred times
    longer than the rest does not send queries to the exact pass (with the table-wide norm bound it
    sends every batched query there); same exact results either way.  3 queries: scan; 40: the
    128-query tile kernel; 150: the ping-pong kernel."""
    n, d = 70_000, 96
    rows = cases.make_rows("synth", 67, n, d)
    rows[::100] *= 100.0
    ids = cases.make_ids("hash", 67, n)
    q = cases.make_queries("synth", 67, nq, d, rows)
    dt = {"f32": _lib.F32, "bf16": _lib.BF16, "f16": _lib.F16}[dtype]
    stored = _load(engine, table_name, rows, ids, O.L2, dt)
    try:
        fallbacks = {}
        for rb in (1, 0):
            engine.set_option("l2_row_bounds", rb)
            before = engine.stats()
            _check(engine, table_name, stored, ids, q, 10, O.L2)
            after = engine.stats()
            fallbacks[rb] = after[1] - before[1]
        assert fallbacks[1] == 0, fallbacks
        print("exact passes with / without the per-row term:", fallbacks[1], fallbacks[0])
        if nq == 150 and dtype == "f32":
            assert fallbacks[0] > 0, fallbacks       # the case the per-row term exists for
    finally:
        engine.set_option("l2_row_bounds", 1)


@pytest.mark.parametrize("metric,dtype,n,d,k,nq", [
    ("cosine", "f32", 10_000_000, 768, 10, 4),        # headline / BASELINE config 2 shape at 10M
    ("cosine", "f32", 10_000_000, 768, 10, 1024),     # config 3: batch 1024
    ("ip", "bf16", 12_500_000, 128, 100, 3),          # config 4: one shard of 100M / 8
    ("l2", "f16", 20_000_000, 768, 10, 256),          # config 5
    ("l2", "f16", 125_000_000, 768, 10, 256),         # config 5, one whole per-GPU shard of 1B / 8: 192 GB
])
def test_full_size_properties(engine, table_name, metric, dtype, n, d, k, nq):
    """BASELINE's own sizes, through size-independent properties: scores are the canonical scores of
    the returned rows (regenerated and quantised on the host), order is (score, id), a planted copy
    of a stored row comes first, no row of a 100k-row sample beats the k-th hit, and the per-query
    scan and the batched path agree bit for bit."""
    from oracle import oracle_c as OC
    m, dt = cases.METRICS[metric], cases.DTYPES[dtype]
    engine.create_table(table_name, d, m, dt)
    engine.reserve(table_name, n)
    engine.insert_synthetic(table_name, n, seed=0, first_row=0)
    q = OC.synth_rows(1, 0, nq, d)
    planted = 7_654_321
    q[nq - 1] = OC.synth_rows(0, planted, 1, d)[0]
    ids_b, sc_b, cnt_b = engine.search(table_name, q, k)            # batched path (nq >= 2)
    assert list(cnt_b) == [k] * nq
    check = sorted(set([0, 1, nq // 2, nq - 2, nq - 1]) & set(range(nq)))
    engine.set_option("gemm_path", 0)
    try:
        ids, sc, cnt = engine.search(table_name, q[check], k)       # one scan per checked query
    finally:
        engine.set_option("gemm_path", 1)
    assert np.array_equal(ids, ids_b[check]) and np.array_equal(sc.view(np.uint32), sc_b[check].view(np.uint32))
    if metric != "l2":
        assert ids_b[nq - 1, 0] == planted
    else:
        assert ids_b[nq - 1, 0] == planted and sc_b[nq - 1, 0] <= 1e-2    # fp16 rounding of the stored copy
    for j, qi in enumerate(check):
        qp = O.prepare_query(q[qi], m)
        raw = np.concatenate([OC.synth_rows(0, int(r), 1, d) for r in ids[j]])
        want = OC.scores(O.prepare_store(raw, m, dt), qp, m)
        assert np.array_equal(want.view(np.uint32), sc[j].view(np.uint32))
        key = sc[j] if metric == "l2" else -sc[j]
        assert np.array_equal(np.lexsort((ids[j], key)), np.arange(k))
        s0 = 1_000_000 + j * 2_000_000
        ssc = OC.scores(O.prepare_store(OC.synth_rows(0, s0, 100_000, d), m, dt), qp, m)
        better = np.nonzero(ssc < sc[j, k - 1] if metric == "l2" else ssc > sc[j, k - 1])[0] + s0
        assert set(better.tolist()) <= set(ids[j].tolist())


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_rows_that_are_not_finite_are_refused(engine, table_name, dtype):
    """A NaN, an infinity or a value beyond the storage type's range makes the whole insert fail with
    VDB_EINVAL: nothing of the call becomes visible, the table's norm bound stays finite and later
    searches answer from the fast path as before."""
    from vectordb_amd._lib import VdbError
    d, n, k = 64, 5000, 10
    rows = cases.make_rows("synth", 77, n, d)
    ids = np.arange(n, dtype=np.int64)
    stored = _load(engine, table_name, rows, ids, O.IP, cases.DTYPES[dtype])
    before = engine.stats()
    for bad_value in (np.nan, np.inf, -np.inf) + ((1.0e6,) if dtype == "f16" else ()):
        bad = cases.make_rows("synth", 78, 7, d)
        bad[3, 11] = bad_value
        with pytest.raises(VdbError) as ei:
            engine.insert(table_name, bad, np.arange(n, n + 7, dtype=np.int64))
        assert ei.value.code == _lib.EINVAL and "finite" in ei.value.message
        assert engine.count(table_name) == n
    # a good insert after the refused ones lands where they would have
    more = cases.make_rows("synth", 79, 3, d)
    engine.insert(table_name, more, np.arange(n, n + 3, dtype=np.int64))
    assert engine.count(table_name) == n + 3
    stored2 = np.concatenate([stored, O.prepare_store(more, O.IP, cases.DTYPES[dtype])])
    q = cases.make_queries("synth", 80, 4, d, rows)
    _check(engine, table_name, stored2, np.arange(n + 3, dtype=np.int64), q, k, O.IP)
    after = engine.stats()
    assert after[1] == before[1], "queries fell back to the exact pass: the norm bound was spoiled"


def test_table_grows_in_place_past_half_of_the_hbm(engine, table_name):
    """No reserve call: eight appends grow a bf16 table to 55 % of the GPU's memory.  Growth by
    allocate-and-copy would need old and new arrays side by side (more than the GPU holds); the table
    grows inside its reserved address range instead, and rows stored by the first append stay where
    they are and stay searchable after every later one."""
    import torch
    _, total = torch.cuda.mem_get_info()
    d, k = 1024, 10
    rows_target = int(0.55 * total / (d * 2 + 12))
    chunk = rows_target // 8
    engine.create_table(table_name, d, _lib.METRIC_IP, _lib.BF16)
    probe = None
    for i in range(8):
        engine.insert_synthetic(table_name, chunk, seed=91, first_row=i * chunk)
        info = engine.table_info(table_name)
        assert info["rows"] == (i + 1) * chunk and info["capacity"] >= info["rows"]
        if probe is None:
            probe, pid = engine.fetch_rows(table_name, 4321, 1)
            assert int(pid[0]) == 4321
        again, _ = engine.fetch_rows(table_name, 4321, 1)
        assert np.array_equal(again.view(np.uint32), probe.view(np.uint32))
        ids, scores, cnt = engine.search(table_name, probe * 4.0, k)
        assert int(cnt[0]) == k
        # the row itself scores ||x||^2 * 4; any row that beats it must really score higher
        own = float((probe[0].astype(np.float64) ** 2).sum() * 4.0)
        assert 4321 in ids[0].tolist() or float(scores[0, -1]) >= np.float32(own)
        assert np.all(np.diff(scores[0]) <= 0)
    assert (engine.table_info(table_name)["rows"] * (d * 2)) > 0.5 * total


def test_mirror_makes_room_for_rows_under_cache_capacity(engine, table_name):
    """gpu_resource_config.cache_capacity (/root/reference/server_config.yaml:42) is reached by rows plus
    their bf16 mirror while the rows alone still fit: the table drops its mirror and keeps taking rows
    (no VDB_ENOMEM for rows that fit), and batched searches, now converting the fp32 rows on the fly,
    still return the oracle's answer."""
    d, n, step, k = 256, 130_000, 10_000, 10
    rows = cases.make_rows("synth", 123, n, d)
    ids = cases.make_ids("hash", 123, n)
    engine.set_option("cache_capacity_gb", 0.16)           # 171.8 MB: 133 MB of rows fit, rows + 67 MB of mirror do not
    try:
        engine.create_table(table_name, d, _lib.METRIC_COSINE, _lib.F32)
        had_mirror = False
        for lo in range(0, n, step):
            engine.insert(table_name, rows[lo:lo + step], ids[lo:lo + step])
            had_mirror = had_mirror or engine.mirror_bytes(table_name) > 0
        assert had_mirror and engine.mirror_bytes(table_name) == 0 and engine.count(table_name) == n
        stored = O.prepare_store(rows, O.COSINE, O.F32)
        q = cases.make_queries("synth", 124, 150, d, rows)
        before = engine.stats()
        _check(engine, table_name, stored, ids, q, k, O.COSINE)          # batched path, no mirror
        _check(engine, table_name, stored, ids, q[:2], k, O.COSINE)
        assert engine.stats()[2] - before[2] >= 150
        # what does not fit even without a mirror is still refused
        from vectordb_amd._lib import VdbError
        with pytest.raises(VdbError) as ei:
            engine.insert_synthetic(table_name, 100_000, seed=9, first_row=0)
        assert ei.value.code == _lib.ENOMEM and engine.count(table_name) == n
    finally:
        engine.set_option("cache_capacity_gb", 0.0)


def test_fp32_table_grows_past_two_thirds_of_the_hbm(engine, table_name):
    """An fp32 table with a bf16 mirror needs 1.5 x its rows.  Grown until the rows alone take more than
    two thirds of the GPU's memory, it must have given up the mirror on the way instead of failing, and
    single and batched searches stay exact (properties as in test_full_size_properties)."""
    import torch
    from oracle import oracle_c as OC
    _, total = torch.cuda.mem_get_info()
    d, k = 768, 10
    rows_target = int(0.68 * total / (d * 4 + 12))
    chunk = rows_target // 10
    engine.create_table(table_name, d, _lib.METRIC_COSINE, _lib.F32)
    seen_mirror = False
    for i in range(10):
        engine.insert_synthetic(table_name, chunk, seed=0, first_row=i * chunk)
        seen_mirror = seen_mirror or engine.mirror_bytes(table_name) > 0
    n = 10 * chunk
    assert engine.count(table_name) == n and n * d * 4 > 0.66 * total
    assert seen_mirror and engine.mirror_bytes(table_name) == 0
    nq = 130
    q = OC.synth_rows(1, 0, nq, d)
    planted = n - 12_345
    q[nq - 1] = OC.synth_rows(0, planted, 1, d)[0]
    ids_b, sc_b, cnt_b = engine.search(table_name, q, k)                  # batched: converting kernel over the fp32 rows
    assert list(cnt_b) == [k] * nq and ids_b[nq - 1, 0] == planted
    check = [0, nq // 2, nq - 1]
    engine.set_option("gemm_path", 0)
    try:
        ids, sc, cnt = engine.search(table_name, q[check], k)
    finally:
        engine.set_option("gemm_path", 1)
    assert np.array_equal(ids, ids_b[check]) and np.array_equal(sc.view(np.uint32), sc_b[check].view(np.uint32))
    for j, qi in enumerate(check):
        qp = O.prepare_query(q[qi], O.COSINE)
        raw = np.concatenate([OC.synth_rows(0, int(r), 1, d) for r in ids[j]])
        want = OC.scores(O.prepare_store(raw, O.COSINE, O.F32), qp, O.COSINE)
        assert np.array_equal(want.view(np.uint32), sc[j].view(np.uint32))
        assert np.array_equal(np.lexsort((ids[j], -sc[j])), np.arange(k))
        s0 = n - 3_000_000 + j * 1_000_000
        ssc = OC.scores(O.prepare_store(OC.synth_rows(0, s0, 100_000, d), O.COSINE, O.F32), qp, O.COSINE)
        better = np.nonzero(ssc > sc[j, k - 1])[0] + s0
        assert set(better.tolist()) <= set(ids[j].tolist())


def _rows_with_long_orthogonal_ones(seed, n, d, q, n_long, factor):
    """Synthetic rows; n_long of them are `factor` times longer than the rest and orthogonal to every
    query (to fp32 rounding): their true scores are about zero, their bf16 fast keys are anything."""
    rows = cases.make_rows("synth", seed, n, d)
    rng = np.random.default_rng(seed)
    basis, _ = np.linalg.qr(q.astype(np.float64).T)                  # [d, nq] orthonormal basis of the queries' span
    where = rng.choice(n, size=n_long, replace=False)
    for r in where:
        v = rng.standard_normal(d)
        v -= basis @ (basis.T @ v)
        rows[r] = (v / np.linalg.norm(v) * np.sqrt(d) * factor).astype(np.float32)
    return rows, np.sort(where)


@pytest.mark.parametrize("nq", [3, 40, 150])
def test_ip_rows_far_longer_than_the_rest_stay_off_the_exact_pass(engine, table_name, nq):
    """IP on raw vectors is the reference's metric (/root/reference/app.py:22,141).  A few rows a hundred
    times longer than the rest and orthogonal to the queries put 2^-8 of their norm into the table-wide
    bound of the batched path, which sends every query to the exact pass; kept on the side list (and
    re-scored canonically for every query) they cost nothing.  Same exact answers either way."""
    n, d = 70_000, 256                                   # 65536 rows or more: batches take the batched path
    q = cases.make_queries("synth", 71, 150, d, None)
    rows, where = _rows_with_long_orthogonal_ones(71, n, d, q, 12, 100.0)
    ids = cases.make_ids("hash", 71, n)
    fallbacks = {}
    try:
        for on in (1, 0):
            engine.set_option("ip_outliers", on)
            name = "%s_%d" % (table_name, on)
            if engine.has_table(name):
                engine.drop_table(name)
            # several inserts: the list is extended as rows arrive
            engine.create_table(name, d, _lib.METRIC_IP, _lib.F32)
            for lo in range(0, n, 25_000):
                engine.insert(name, rows[lo:lo + 25_000], ids[lo:lo + 25_000])
            stored = O.prepare_store(rows, O.IP, O.F32)
            before = engine.stats()
            _check(engine, name, stored, ids, q[:nq], 10, O.IP)
            fallbacks[on] = engine.stats()[1] - before[1]
            engine.drop_table(name)
        assert fallbacks[1] == 0, fallbacks
        if nq >= 40:
            assert fallbacks[0] > 0, fallbacks               # the case the list exists for
    finally:
        engine.set_option("ip_outliers", 1)
        for on in (1, 0):
            if engine.has_table("%s_%d" % (table_name, on)):
                engine.drop_table("%s_%d" % (table_name, on))


def test_long_rows_that_are_the_best_hits_are_still_found(engine, table_name, tmp_path):
    """Rows on the side list are answers like any other: long rows aligned with a query come first, also
    after the table went through a snapshot (the list is rebuilt on load), and a row that arrives later
    and is longer still joins the list."""
    n, d, nq = 70_000, 128, 50
    q = cases.make_queries("synth", 81, nq, d, None)
    rows = cases.make_rows("synth", 81, n, d)
    for j, r in enumerate((5, 777, 60_001)):
        rows[r] = q[j] * 40.0                                            # long and parallel to query j
    rows[12_345] *= 60.0                                                 # long, unrelated
    ids = cases.make_ids("hash", 81, n)
    stored = _load(engine, table_name, rows, ids, O.IP, _lib.F32)
    before = engine.stats()
    _check(engine, table_name, stored, ids, q, 10, O.IP)
    got_ids, _, _ = engine.search(table_name, q[:3], 10)
    assert [int(got_ids[j, 0]) for j in range(3)] == [int(ids[5]), int(ids[777]), int(ids[60_001])]
    path = str(tmp_path / "long.vdbsnap")
    engine.save_table(table_name, path)
    engine.drop_table(table_name)
    engine.load_table(table_name, path)
    _check(engine, table_name, stored, ids, q, 10, O.IP)
    late = (q[7:8] * 90.0).astype(np.float32)
    engine.insert(table_name, late, np.array([2 ** 40], dtype=np.int64))
    stored2 = np.concatenate([stored, O.prepare_store(late, O.IP, O.F32)])
    ids2 = np.concatenate([ids, np.array([2 ** 40], dtype=np.int64)])
    _check(engine, table_name, stored2, ids2, q, 10, O.IP)
    assert engine.stats()[1] == before[1]


@pytest.mark.parametrize("d", [7, 384, 1100])
def test_calls_of_a_few_queries_work_on_the_pinned_block_directly(engine, table_name, d):
    """Blocking calls of up to four queries skip both copies (the kernels read the queries from pinned host
    memory, write the answers there, and the caller watches one word per query): same bits as the copying
    path (option host_direct 0) and as the oracle, for query blocks whose length is not a multiple of 16 bytes,
    queries longer than the 1024 elements the preparation keeps in registers, the exact pass asked for
    through the words, and the scan over the mirror, which does not write them."""
    n = 6_000
    rows = cases.make_rows("synth", 91, n, d)
    ids = cases.make_ids("hash", 91, n)
    stored = _load(engine, table_name, rows, ids, O.COSINE, _lib.F32)
    q = cases.make_queries("synth", 92, 5, d, rows)
    try:
        for nq in (1, 3, 4, 5):
            got = {}
            for direct in (1, 0):
                engine.set_option("host_direct", direct)
                got[direct] = engine.search(table_name, q[:nq], 10)
            for a, b in zip(got[1], got[0]):
                assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
        engine.set_option("host_direct", 1)
        _check(engine, table_name, stored, ids, q[:2], 10, O.COSINE)
        engine.set_option("force_exact", 1)
        _, f0, _ = engine.stats()
        _check(engine, table_name, stored, ids, q[:3], 10, O.COSINE)
        assert engine.stats()[1] - f0 == 3
        engine.set_option("force_exact", 0)
        engine.set_option("scan_mirror", 1)
        _check(engine, table_name, stored, ids, q[:2], 10, O.COSINE)
    finally:
        engine.set_option("host_direct", 1)
        engine.set_option("force_exact", 0)
        engine.set_option("scan_mirror", 0)
