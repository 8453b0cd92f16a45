"""Merging of concurrent searches (vectordb_amd/coalesce.py) against a stand-in search function:
every caller gets its own rows back, callers that arrive during a call share the next call, errors
reach every caller of the failing batch and nobody else."""
import threading
import time

import numpy as np
import pytest

from vectordb_amd.coalesce import SearchCoalescer


class _FakeEngine:
    """Answers query row i with ids = round(q[i, 0]) + arange(k); records the batch sizes."""

    def __init__(self, delay=0.02):
        self.delay = delay
        self.batches = []
        self.lock = threading.Lock()

    def search(self, name, q, k):
        with self.lock:
            self.batches.append((name, q.shape[0], k))
        time.sleep(self.delay)
        if name == "missing":
            raise KeyError("no such table: missing")
        base = np.rint(q[:, 0]).astype(np.int64)
        ids = base[:, None] + np.arange(k, dtype=np.int64)[None, :]
        return ids, ids.astype(np.float32), np.full((q.shape[0],), k, dtype=np.int32)


def _run(co, name, value, k, out, idx, nq=1, d=4):
    q = np.full((nq, d), float(value), dtype=np.float32)
    try:
        out[idx] = co.search(name, q, k)
    except Exception as exc:                      # noqa: BLE001 - the test inspects it
        out[idx] = exc


def test_a_lone_caller_goes_straight_through():
    eng = _FakeEngine(delay=0.0)
    co = SearchCoalescer(eng.search)
    ids, sc, cnt = co.search("t", np.full((3, 4), 7.0, dtype=np.float32), 5)
    assert ids.shape == (3, 5) and ids[0, 0] == 7 and cnt.tolist() == [5, 5, 5]
    assert eng.batches == [("t", 3, 5)] and co.calls == 1 and co.requests == 1


def test_callers_that_arrive_during_a_call_share_the_next_one():
    eng = _FakeEngine(delay=0.05)
    co = SearchCoalescer(eng.search)
    n = 12
    out = [None] * n
    threads = [threading.Thread(target=_run, args=(co, "t", 100 + i, 3, out, i)) for i in range(n)]
    threads[0].start()
    time.sleep(0.01)                              # the first call is under way
    for t in threads[1:]:
        t.start()
    for t in threads:
        t.join(timeout=10)
    for i in range(n):
        ids, sc, cnt = out[i]
        assert ids.shape == (1, 3) and ids[0].tolist() == [100 + i, 101 + i, 102 + i]
        assert cnt.tolist() == [3]
    sizes = [b[1] for b in eng.batches]
    assert sum(sizes) == n and sizes[0] == 1
    assert len(sizes) <= 3 and max(sizes) >= (n - 1) // 2      # the eleven late callers went in one or two calls
    assert co.requests == n and co.calls == len(sizes)


def test_only_like_requests_are_merged():
    eng = _FakeEngine(delay=0.05)
    co = SearchCoalescer(eng.search)
    specs = [("t", 3, 4), ("t", 3, 4), ("u", 3, 4), ("t", 5, 4), ("t", 3, 8), ("t", 3, 4)]
    out = [None] * (len(specs) + 1)
    first = threading.Thread(target=_run, args=(co, "t", 1, 3, out, len(specs)))
    first.start()
    time.sleep(0.01)
    threads = [threading.Thread(target=_run, args=(co, name, 10 * i, k, out, i, 1, d)) for i, (name, k, d) in enumerate(specs)]
    for t in threads:
        t.start()
    for t in threads + [first]:
        t.join(timeout=10)
    for i, (name, k, d) in enumerate(specs):
        ids, sc, cnt = out[i]
        assert ids.shape == (1, k) and ids[0, 0] == 10 * i
    # after the first call: one merged call for the three ("t", 3, 4) requests, one each for the others
    later = sorted(eng.batches[1:])
    assert later == sorted([("t", 3, 3), ("u", 1, 3), ("t", 1, 5), ("t", 1, 3)])


def test_an_error_reaches_every_caller_of_its_batch_only():
    eng = _FakeEngine(delay=0.05)
    co = SearchCoalescer(eng.search)
    out = [None] * 5
    first = threading.Thread(target=_run, args=(co, "t", 1, 2, out, 0))
    first.start()
    time.sleep(0.01)
    rest = [threading.Thread(target=_run, args=(co, "missing", 2, 2, out, 1)),
            threading.Thread(target=_run, args=(co, "missing", 3, 2, out, 2)),
            threading.Thread(target=_run, args=(co, "t", 4, 2, out, 3)),
            threading.Thread(target=_run, args=(co, "t", 5, 2, out, 4))]
    for t in rest:
        t.start()
    for t in rest + [first]:
        t.join(timeout=10)
    assert isinstance(out[1], KeyError) and isinstance(out[2], KeyError)
    for i, v in ((0, 1), (3, 4), (4, 5)):
        assert not isinstance(out[i], Exception) and out[i][0][0, 0] == v
    # the coalescer is usable afterwards
    assert co.search("t", np.full((1, 4), 9.0, dtype=np.float32), 2)[0][0, 0] == 9


def test_a_batch_is_capped():
    eng = _FakeEngine(delay=0.05)
    co = SearchCoalescer(eng.search, max_queries=8)
    out = [None] * 7
    first = threading.Thread(target=_run, args=(co, "t", 0, 2, out, 0))
    first.start()
    time.sleep(0.01)
    rest = [threading.Thread(target=_run, args=(co, "t", i, 2, out, i, 3)) for i in range(1, 7)]   # 3 queries each
    for t in rest:
        t.start()
    for t in rest + [first]:
        t.join(timeout=10)
    assert all(out[i][0].shape == (3 if i else 1, 2) and out[i][0][0, 0] == i for i in range(7))
    assert max(b[1] for b in eng.batches) <= 9        # the cap plus at most the request that takes the turn
