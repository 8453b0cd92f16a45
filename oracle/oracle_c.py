"""ctypes wrapper of the C restatement (oracle/oracle_c.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("build the C oracle first: make -C oracle")
        _lib = C.CDLL(_PATH)
        _lib.oc_topk.restype = C.c_int64
    return _lib


def scores(stored: np.ndarray, qprep: np.ndarray, metric: int) -> np.ndarray:
    stored = np.ascontiguousarray(stored, dtype=np.float32)
    qprep = np.ascontiguousarray(qprep, dtype=np.float32)
    out = np.empty((stored.shape[0],), dtype=np.float32)
    load().oc_scores(C.c_void_p(stored.ctypes.data), C.c_int64(stored.shape[0]), C.c_int(stored.shape[1]),
                     C.c_void_p(qprep.ctypes.data), C.c_int(metric), C.c_void_p(out.ctypes.data))
    return out


def normalize_rows(rows: np.ndarray) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.empty_like(rows)
    load().oc_normalize(C.c_void_p(rows.ctypes.data), C.c_int64(rows.shape[0]), C.c_int(rows.shape[1]),
                        C.c_void_p(out.ctypes.data))
    return out


def synth_rows(seed: int, row0: int, n: int, d: int) -> np.ndarray:
    out = np.empty((n, d), dtype=np.float32)
    load().oc_synth_rows(C.c_uint64(seed), C.c_int64(row0), C.c_int64(n), C.c_int(d), C.c_void_p(out.ctypes.data))
    return out


def topk(sc: np.ndarray, ids: np.ndarray, k: int, metric: int):
    sc = np.ascontiguousarray(sc, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    oi = np.empty((k,), dtype=np.int64)
    os_ = np.empty((k,), dtype=np.float32)
    c = load().oc_topk(C.c_void_p(sc.ctypes.data), C.c_void_p(ids.ctypes.data), C.c_int64(sc.shape[0]), C.c_int(k),
                       C.c_int(metric), C.c_void_p(oi.ctypes.data), C.c_void_p(os_.ctypes.data))
    return oi[:c], os_[:c]
