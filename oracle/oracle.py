"""numpy oracle: exact brute-force top-k, the specification the HIP path must match.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package.

PARITY UNPINNED for the search arithmetic: the reference holds no scoring or
top-k code and no test of it.  /root/reference/app.py:379-409 (lookup_milvus)
forwards the query to an external, un-vendored, unpinned Milvus server
(app.py:391 `milvus.search_vectors`, requirements.txt:3), configured for an
approximate IVFLAT index (app.py:145-149,388).  What the reference does pin is
followed here:

  * result shape and order: best first, at most k hits, one (id, distance)
    pair per hit (app.py:392-406);
  * one caller-supplied signed 64-bit id per stored row (app.py:315-325);
  * dimension and metric fixed per table, metric = inner product (app.py:137-142,
    app.py:22); cosine and L2 are additions of this build;
  * ids derive from `hash_vector` (app.py:328-334), restated below and pinned by
    tests/golden/hash_vector.json (generated from the reference's own function).

The arithmetic below is therefore this build's own specification.

Canonical score
---------------
Stored values and query values are exact binary floating-point numbers (fp32
queries; fp32, bf16 or fp16 stored rows).  A score is accumulated in fp64 in a
fixed order and rounded once to fp32:

  * 64 interleaved partial sums: partial p (0..63) accumulates, for
    j = p, p+64, p+128, ... in increasing j, the term of element j;
  * the partials are combined by a fixed binary tree:
    for s in 32,16,8,4,2,1: partial[p] += partial[p+s] for p < s;
  * partial[0] is rounded to fp32 (round-to-nearest-even).

Terms: IP/COSINE `double(x_j) * double(q_j)` (exact in fp64); L2
`d = double(x_j) - double(q_j); d*d` (each operation rounded separately, no
fused multiply-add).  COSINE rows are normalised when stored and the query when
searched: `n2` = the fp64 tree sum of `x_j*x_j`, `inv = 1.0 / sqrt(n2)` in fp64,
`y_j = fp32(double(x_j) * inv)`; a zero vector stays zero.

Order: IP/COSINE larger score first, L2 smaller distance first; equal scores
are ordered by id ascending as signed 64-bit integers.  `min(k, N)` hits.
Non-finite inputs are outside the specification.
"""
from __future__ import annotations

import hashlib

import numpy as np

L2, IP, COSINE = 1, 2, 3          # metric codes shared with include/vdb.h
F32, BF16, F16 = 0, 1, 2          # storage dtype codes shared with include/vdb.h

LANES = 64

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_COLK = np.uint64(0xD1B54A32D192ED03)
# fp32 nearest to 1/sqrt(4*(65536^2-1)/12): gives the synthetic values unit variance.
SYNTH_SCALE = np.array([0x37DDB3D7], dtype=np.uint32).view(np.float32)[0]   # 0x1.bb67aep-16


# --------------------------------------------------------------------------- ids
def hash_vector(vector) -> int:
    """Restates /root/reference/app.py:328-334: '%.5f' per component joined by ',',
    md5, first 8 digest bytes little-endian as a signed 64-bit integer."""
    vector_str = ",".join("%.5f" % num for num in vector)
    digest = hashlib.md5(vector_str.encode("utf-8")).digest()
    return int.from_bytes(digest[:8], "little", signed=True)


# --------------------------------------------------------------- synthetic data
def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def synth_rows(seed: int, row0: int, nrows: int, d: int) -> np.ndarray:
    """Counter-based synthetic embeddings, bit-identical to the device generator
    (vdb_synth_fill_dev): element (row, col) depends only on (seed, row, col).

    key = mix64(seed + GOLD*(row+1)); h = mix64(key ^ COLK*(col+1));
    value = fp32(sum of the four 16-bit fields of h - 131070) * SYNTH_SCALE
    (approximately N(0,1): an Irwin-Hall sum of four uniforms)."""
    with np.errstate(over="ignore"):
        rows = np.arange(row0, row0 + nrows, dtype=np.uint64)
        cols = np.arange(d, dtype=np.uint64)
        key = _mix64(np.uint64(seed) + _GOLD * (rows + np.uint64(1)))
        h = _mix64(key[:, None] ^ (_COLK * (cols + np.uint64(1)))[None, :])
    m = np.uint64(0xFFFF)
    a = (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))
    v = (a.astype(np.int64) - 131070).astype(np.float32)
    return v * SYNTH_SCALE


# ------------------------------------------------------------------ quantisation
def quantize(x32: np.ndarray, dtype: int) -> np.ndarray:
    """Round fp32 values to the storage dtype (round-to-nearest-even) and return
    them as fp32 (exactly representable)."""
    x32 = np.ascontiguousarray(x32, dtype=np.float32)
    if dtype == F32:
        return x32
    if dtype == F16:
        return x32.astype(np.float16).astype(np.float32)
    if dtype == BF16:
        u = x32.view(np.uint32)
        r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
        return (((u + r) >> np.uint32(16)) << np.uint32(16)).view(np.float32)
    raise ValueError("dtype")


# ----------------------------------------------------------- canonical arithmetic
def _tree(acc: np.ndarray) -> np.ndarray:
    s = LANES // 2
    while s >= 1:
        acc[:, :s] += acc[:, s:2 * s]
        s //= 2
    return acc[:, 0]


def canon_dot64(x: np.ndarray, q: np.ndarray) -> np.ndarray:
    """fp64 interleaved-tree sum of x[i,j]*q[j]; x [N,d], q [d] -> [N] float64."""
    x = np.asarray(x)
    q = np.asarray(q, dtype=np.float64)
    n, d = x.shape
    acc = np.zeros((n, LANES), dtype=np.float64)
    for j0 in range(0, d, LANES):
        j1 = min(j0 + LANES, d)
        acc[:, : j1 - j0] += x[:, j0:j1].astype(np.float64) * q[j0:j1]
    return _tree(acc)


def canon_l2_64(x: np.ndarray, q: np.ndarray) -> np.ndarray:
    x = np.asarray(x)
    q = np.asarray(q, dtype=np.float64)
    n, d = x.shape
    acc = np.zeros((n, LANES), dtype=np.float64)
    for j0 in range(0, d, LANES):
        j1 = min(j0 + LANES, d)
        diff = x[:, j0:j1].astype(np.float64) - q[j0:j1]
        acc[:, : j1 - j0] += diff * diff
    return _tree(acc)


def canon_n2_64(x: np.ndarray) -> np.ndarray:
    """fp64 interleaved-tree sum of squares per row."""
    x = np.asarray(x)
    n, d = x.shape
    acc = np.zeros((n, LANES), dtype=np.float64)
    for j0 in range(0, d, LANES):
        j1 = min(j0 + LANES, d)
        xx = x[:, j0:j1].astype(np.float64)
        acc[:, : j1 - j0] += xx * xx
    return _tree(acc)


def normalize_rows(x32: np.ndarray) -> np.ndarray:
    """Canonical L2 normalisation of fp32 rows -> fp32 rows."""
    x32 = np.ascontiguousarray(x32, dtype=np.float32)
    n2 = canon_n2_64(x32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / np.sqrt(n2)
    inv[n2 == 0.0] = 0.0
    return (x32.astype(np.float64) * inv[:, None]).astype(np.float32)


# ------------------------------------------------------------------- store/query
def prepare_store(x32: np.ndarray, metric: int, dtype: int = F32) -> np.ndarray:
    """What a table holds after insert: normalised for COSINE, then rounded to
    the storage dtype; returned as fp32 values."""
    x32 = np.ascontiguousarray(x32, dtype=np.float32)
    if x32.ndim == 1:
        x32 = x32[None, :]
    if metric == COSINE:
        x32 = normalize_rows(x32)
    return quantize(x32, dtype)


def prepare_query(q32: np.ndarray, metric: int) -> np.ndarray:
    q32 = np.ascontiguousarray(q32, dtype=np.float32)
    if metric == COSINE:
        return normalize_rows(q32[None, :])[0]
    return q32


def scores(stored: np.ndarray, qprep: np.ndarray, metric: int) -> np.ndarray:
    """Canonical fp32 scores of one prepared query against stored rows."""
    if stored.shape[0] == 0:
        return np.zeros((0,), dtype=np.float32)
    if metric == L2:
        return canon_l2_64(stored, qprep).astype(np.float32)
    return canon_dot64(stored, qprep).astype(np.float32)


def topk(sc: np.ndarray, ids: np.ndarray, k: int, metric: int):
    """Order by (score best first, id ascending); return (ids[k'], scores[k'])."""
    ids = np.asarray(ids, dtype=np.int64)
    primary = sc if metric == L2 else -sc
    order = np.lexsort((ids, primary))[: max(0, min(k, sc.shape[0]))]
    return ids[order], sc[order]


def search(stored: np.ndarray, ids: np.ndarray, queries: np.ndarray, k: int, metric: int,
           block: int = 65536):
    """Brute-force search of raw fp32 queries [nq,d] against stored rows [N,d].
    Returns a list of (ids, scores) per query."""
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    if queries.ndim == 1:
        queries = queries[None, :]
    ids = np.asarray(ids, dtype=np.int64)
    out = []
    for q in queries:
        qp = prepare_query(q, metric)
        sc = np.empty((stored.shape[0],), dtype=np.float32)
        for r0 in range(0, stored.shape[0], block):
            sc[r0:r0 + block] = scores(stored[r0:r0 + block], qp, metric)
        out.append(topk(sc, ids, k, metric))
    return out


# ------------------------------------------------ fast CPU path (timing baseline)
def fast_search(stored: np.ndarray, ids: np.ndarray, queries: np.ndarray, k: int, metric: int,
                n2: np.ndarray | None = None, margin: int = 54):
    """BLAS scoring (fp32 `X @ q`), argpartition to k+margin candidates, canonical
    re-scoring of the candidates, (score, id) order.  Agrees with `search` whenever
    the candidate margin exceeds the fp32 scoring error (checked in tests); this is
    the CPU baseline bench.py times."""
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    if queries.ndim == 1:
        queries = queries[None, :]
    ids = np.asarray(ids, dtype=np.int64)
    n = stored.shape[0]
    kk = min(n, k + margin)
    qp = np.stack([prepare_query(q, metric) for q in queries])
    fast = stored @ qp.T                                   # [N, nq] fp32 BLAS
    if metric == L2:
        if n2 is None:
            n2 = canon_n2_64(stored).astype(np.float32)
        fast = 2.0 * fast - n2[:, None]
    out = []
    for qi in range(queries.shape[0]):
        col = fast[:, qi]
        cand = np.argpartition(-col, kk - 1)[:kk] if kk < n else np.arange(n)
        sc = scores(stored[cand], qp[qi], metric)
        out.append(topk(sc, ids[cand], k, metric))
    return out
