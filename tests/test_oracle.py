"""CPU tests of the oracle itself: pinned against the reference's own function where one exists
(hash_vector), against exact rational arithmetic, and against its committed golden outputs."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import cases
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def test_hash_vector_matches_reference_golden():
    golden = json.load(open(os.path.join(HERE, "golden", "hash_vector.json")))
    assert len(golden) >= 5
    for g in golden:
        assert O.hash_vector(g["vector"]) == g["hash"]
    # known answers of app.py:328-334 for three fixed inputs (computed from its definition by hand: md5 of the text form)
    assert O.hash_vector([0.0] * 8) == 8770679782603771991
    assert O.hash_vector([1.0, 2.0, 3.0, 4.0]) == -232953353825745962
    assert O.hash_vector([]) == 338333539836370388


def _exact_dot(x, q):
    return sum(Fraction(float(a)) * Fraction(float(b)) for a, b in zip(x, q))


def _ulp32(v):
    v = np.float32(abs(v))
    return float(np.spacing(v)) if v > 0 else float(np.spacing(np.float32(0)))


@pytest.mark.parametrize("d", [1, 7, 64, 70, 200])
def test_canonical_dot_is_within_one_ulp_of_exact(d):
    rng = np.random.default_rng(d)
    x = rng.standard_normal((40, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    got = O.scores(x, q, O.IP)
    for i in range(x.shape[0]):
        exact = _exact_dot(x[i], q)
        assert abs(Fraction(float(got[i])) - exact) <= Fraction(_ulp32(float(exact)))


def test_canonical_l2_is_within_one_ulp_of_exact():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((30, 50)).astype(np.float32)
    q = rng.standard_normal(50).astype(np.float32)
    got = O.scores(x, q, O.L2)
    for i in range(x.shape[0]):
        exact = sum((Fraction(float(a)) - Fraction(float(b))) ** 2 for a, b in zip(x[i], q))
        assert abs(Fraction(float(got[i])) - exact) <= Fraction(_ulp32(float(exact)))


def test_tree_order_is_the_documented_one():
    # 64 interleaved partials combined by the fixed binary tree, spelled out with scalar loops
    rng = np.random.default_rng(11)
    d = 200
    x = rng.standard_normal(d).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    part = [0.0] * 64
    for j in range(d):
        part[j % 64] = part[j % 64] + float(x[j]) * float(q[j])
    s = 32
    while s >= 1:
        for p in range(s):
            part[p] = part[p] + part[p + s]
        s //= 2
    assert np.float32(part[0]) == O.scores(x[None, :], q, O.IP)[0]


def test_normalize_rows():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((100, 37)).astype(np.float32)
    x[3] = 0.0
    y = O.normalize_rows(x)
    n = np.linalg.norm(y.astype(np.float64), axis=1)
    assert np.all(y[3] == 0.0)
    assert np.allclose(np.delete(n, 3), 1.0, atol=1e-6)


def test_quantize_bf16_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.3895314e38, 1e-40], dtype=np.float32)
    y = O.quantize(x, O.BF16)
    assert y[0] == 1.0
    assert y[1] == 1.0            # exactly half way between 1.0 and 1.0078125: ties to even
    assert y[2] == np.float32(1.015625)   # half way between 1.0078125 and 1.015625: ties to even
    assert y[3] == -2.5
    assert np.all((y.view(np.uint32) & 0xFFFF) == 0)
    h = O.quantize(np.array([0.1], dtype=np.float32), O.F16)
    assert h[0] == np.float32(np.float16(0.1))


def test_topk_order_ties_by_signed_id():
    sc = np.array([1.0, 2.0, 2.0, 2.0, 0.5], dtype=np.float32)
    ids = np.array([5, 9, -3, 4, -100], dtype=np.int64)
    i, s = O.topk(sc, ids, 3, O.IP)
    assert list(i) == [-3, 4, 9] and list(s) == [2.0, 2.0, 2.0]
    i, s = O.topk(sc, ids, 10, O.L2)
    assert list(i) == [-100, 5, -3, 4, 9]
    i, s = O.topk(sc[:0], ids[:0], 3, O.IP)
    assert len(i) == 0


def test_search_golden_regression():
    golden = json.load(open(os.path.join(HERE, "golden", "search_golden.json")))
    assert len(golden) == len(cases.GOLDEN_CASES)
    for g, case in zip(golden, cases.GOLDEN_CASES):
        assert tuple(g["case"]) == case
        _, _, _, res = cases.run_oracle(case)
        for qi, (ids, sc) in enumerate(res):
            assert [int(x) for x in ids] == g["ids"][qi]
            assert [int(x) for x in sc.view(np.uint32)] == g["score_bits"][qi]


@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
def test_fast_search_agrees_with_search(metric):
    rows = cases.make_rows("synth", 21, 5000, 96)
    ids = cases.make_ids("hash", 21, 5000)
    q = cases.make_queries("synth", 21, 4, 96, rows)
    stored = O.prepare_store(rows, cases.METRICS[metric])
    a = O.search(stored, ids, q, 10, cases.METRICS[metric])
    b = O.fast_search(stored, ids, q, 10, cases.METRICS[metric])
    for (ia, sa), (ib, sb) in zip(a, b):
        assert np.array_equal(ia, ib) and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))


def test_synth_rows_is_counter_based():
    a = O.synth_rows(0, 0, 64, 48)
    b = O.synth_rows(0, 32, 32, 48)
    assert np.array_equal(a[32:], b)
    assert not np.array_equal(O.synth_rows(1, 0, 4, 48), a[:4])
    big = O.synth_rows(3, 0, 2000, 256)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 60), st.integers(1, 12), st.integers(1, 5), st.integers(0, 2**31 - 1))
def test_sharded_topk_merge_equals_global(n, k, shards, seed):
    """top-k is a decomposable reduction: merging per-shard top-k lists gives the global list,
    ties and negative ids included (the property the multi-GPU merge relies on)."""
    rng = np.random.default_rng(seed)
    sc = (rng.integers(-3, 4, size=n) / 2.0).astype(np.float32)     # many ties
    ids = rng.integers(-50, 50, size=n).astype(np.int64)
    for metric in (O.IP, O.L2):
        gi, gs = O.topk(sc, ids, k, metric)
        part_i, part_s = [], []
        for sh in range(shards):
            sl = slice(sh * n // shards, (sh + 1) * n // shards)
            i, s = O.topk(sc[sl], ids[sl], k, metric)
            part_i.append(i)
            part_s.append(s)
        mi, ms = O.topk(np.concatenate(part_s), np.concatenate(part_i), k, metric)
        assert np.array_equal(mi, gi) and np.array_equal(ms, gs)


def test_c_restatement_matches_numpy_oracle():
    from oracle import oracle_c as OC
    x = O.synth_rows(3, 10, 800, 130)
    assert np.array_equal(x.view(np.uint32), OC.synth_rows(3, 10, 800, 130).view(np.uint32))
    q = O.synth_rows(4, 0, 1, 130)[0]
    ids = cases.make_ids("hash", 1, 800)
    for m in (O.IP, O.L2, O.COSINE):
        for dt in (O.F32, O.BF16, O.F16):
            st = O.prepare_store(x, m, dt)
            qp = O.prepare_query(q, m)
            a, b = O.scores(st, qp, m), OC.scores(st, qp, m)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            ai, as_ = O.topk(a, ids, 17, m)
            bi, bs = OC.topk(a, ids, 17, m)
            assert np.array_equal(ai, bi) and np.array_equal(as_.view(np.uint32), bs.view(np.uint32))
    assert np.array_equal(O.normalize_rows(x).view(np.uint32), OC.normalize_rows(x).view(np.uint32))
