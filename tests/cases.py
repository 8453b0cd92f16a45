"""Deterministic test datasets shared by the CPU and GPU tests and the golden generator."""
import numpy as np

from oracle import oracle as O


def make_rows(kind: str, seed: int, n: int, d: int) -> np.ndarray:
    """fp32 rows [n, d]."""
    if kind == "synth":          # the device generator's distribution (about N(0,1))
        return O.synth_rows(seed, 0, n, d)
    rng = np.random.default_rng(seed)
    if kind == "uniform":        # what the reference's createVector.py:6 draws, seeded
        return rng.random((n, d)).astype(np.float32)
    if kind == "ties":           # multiples of 1/8: fp32 dot products are exact, many equal scores
        base = rng.integers(-8, 9, size=(max(1, n // 3), d)).astype(np.float32) / 8.0
        return base[rng.integers(0, base.shape[0], size=n)]
    if kind == "dups":           # every row identical
        row = rng.integers(-8, 9, size=(1, d)).astype(np.float32) / 8.0
        return np.repeat(row, n, axis=0)
    raise ValueError(kind)


def make_ids(kind: str, seed: int, n: int) -> np.ndarray:
    if kind == "index":
        return np.arange(n, dtype=np.int64)
    rng = np.random.default_rng(seed + 7919)
    if kind == "hash":           # md5-derived ids are uniform over the signed 64-bit range
        return rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)
    if kind == "negdesc":        # strictly descending, mostly negative
        return (np.arange(n, dtype=np.int64)[::-1] - n // 2 - 5).copy()
    raise ValueError(kind)


def make_queries(kind: str, seed: int, nq: int, d: int, rows: np.ndarray) -> np.ndarray:
    if kind in ("ties", "dups"):
        rng = np.random.default_rng(seed + 104729)
        return rng.integers(-8, 9, size=(nq, d)).astype(np.float32) / 8.0
    if kind == "uniform":
        return np.random.default_rng(seed + 1).random((nq, d)).astype(np.float32)
    return O.synth_rows(seed + 1, 0, nq, d)


METRICS = {"l2": O.L2, "ip": O.IP, "cosine": O.COSINE}
DTYPES = {"f32": O.F32, "bf16": O.BF16, "f16": O.F16}

# (rows kind, ids kind, seed, n, d, metric, dtype, k, nq)
GOLDEN_CASES = [
    ("synth", "index", 0, 1000, 8, "cosine", "f32", 10, 3),
    ("synth", "hash", 1, 1000, 8, "ip", "f32", 10, 3),
    ("synth", "hash", 2, 777, 100, "l2", "f32", 10, 2),
    ("uniform", "hash", 3, 2000, 384, "cosine", "f32", 10, 2),
    ("synth", "index", 4, 1500, 128, "ip", "bf16", 100, 2),
    ("synth", "index", 5, 1200, 768, "l2", "f16", 10, 2),
    ("ties", "hash", 6, 600, 16, "ip", "f32", 10, 3),
    ("ties", "negdesc", 7, 600, 16, "l2", "f32", 25, 2),
    ("dups", "hash", 8, 300, 12, "cosine", "f32", 10, 1),
    ("synth", "index", 9, 5, 33, "cosine", "f32", 10, 2),
]


def run_oracle(case):
    rk, ik, seed, n, d, metric, dtype, k, nq = case
    rows = make_rows(rk, seed, n, d)
    ids = make_ids(ik, seed, n)
    queries = make_queries(rk, seed, nq, d, rows)
    stored = O.prepare_store(rows, METRICS[metric], DTYPES[dtype])
    res = O.search(stored, ids, queries, k, METRICS[metric])
    return rows, ids, queries, res


def oracle_search(stored, ids, queries, k, metric):
    """Oracle answer per query.  Small cases use the numpy oracle directly; large ones score with the
    C restatement (OpenMP), which tests/test_oracle.py pins bit for bit to the numpy oracle, and rank
    with the numpy oracle's (score, id) order."""
    from oracle import oracle as O
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    if queries.ndim == 1:
        queries = queries[None, :]
    if stored.shape[0] * stored.shape[1] * queries.shape[0] < 2e8:
        return O.search(stored, ids, queries, k, metric)
    from oracle import oracle_c as OC
    out = []
    for q in queries:
        qp = O.prepare_query(q, metric)
        out.append(O.topk(OC.scores(stored, qp, metric), ids, k, metric))
    return out
