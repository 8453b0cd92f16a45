"""Host-side wrapper over the C ABI: numpy in, numpy out, plus device-pointer entry points for
callers that keep queries and results in HBM (PyTorch tensors supply the pointers).

All arithmetic happens in libvdb.so on the GPU; this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import (F16, F32, BF16, METRIC_COSINE, METRIC_IP, METRIC_L2, VdbError, check)
from .config import EngineConfig

_init_lock = threading.Lock()


_CBuf = C.c_char * 0          # zero-length ctypes array: from_buffer(arr) passes arr's address to a pointer argument


def _name(s: str) -> bytes:
    return s.encode("utf-8")


class Engine:
    """One process-wide engine bound to one GPU."""

    def __init__(self, device: int = 0, config: Optional[EngineConfig] = None):
        self.lib = _lib.load()
        with _init_lock:
            check(self.lib.vdb_init(int(device)))
        self.device = int(device)
        self._dims = {}            # table name -> dimension (see _dim)
        self._dims_at = -1
        self.config = config or EngineConfig()
        check(self.lib.vdb_set_option(b"use_blas_threshold", float(self.config.use_blas_threshold)))
        check(self.lib.vdb_set_option(b"cache_capacity_gb", float(self.config.cache_capacity_gb)))
        check(self.lib.vdb_set_option(b"insert_buffer_size_gb", float(self.config.insert_buffer_size_gb)))
        check(self.lib.vdb_set_option(b"coalesce_host", 1.0 if self.config.coalesce_searches else 0.0))
        check(self.lib.vdb_set_option(b"scan_mirror", 1.0 if self.config.scan_mirror else 0.0))
        check(self.lib.vdb_set_option(b"bf16_mirror", 1.0 if self.config.bf16_mirror else 0.0))

    # ---- lifecycle
    def create_table(self, name: str, dim: int, metric: int = METRIC_IP, dtype: int = F32) -> None:
        check(self.lib.vdb_create_table(_name(name), int(dim), int(metric), int(dtype)))

    def has_table(self, name: str) -> bool:
        out = C.c_int(0)
        check(self.lib.vdb_has_table(_name(name), C.byref(out)))
        return bool(out.value)

    def drop_table(self, name: str) -> None:
        check(self.lib.vdb_drop_table(_name(name)))

    def create_index(self, name: str, index_type: int = _lib.INDEX_IVFLAT, nlist: int = 2048) -> None:
        check(self.lib.vdb_create_index(_name(name), int(index_type), int(nlist)))

    def count(self, name: str) -> int:
        out = C.c_int64(0)
        check(self.lib.vdb_count(_name(name), C.byref(out)))
        return int(out.value)

    def reserve(self, name: str, rows: int) -> None:
        check(self.lib.vdb_reserve(_name(name), int(rows)))

    def table_info(self, name: str) -> dict:
        dim, rows, cap, stride = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        metric, dtype = C.c_int(), C.c_int()
        check(self.lib.vdb_table_info(_name(name), C.byref(dim), C.byref(metric), C.byref(dtype),
                                      C.byref(rows), C.byref(cap), C.byref(stride)))
        return dict(dim=dim.value, metric=metric.value, dtype=dtype.value, rows=rows.value,
                    capacity=cap.value, row_stride_bytes=stride.value)

    def mirror_bytes(self, name: str) -> int:
        """HBM bytes of the table's bf16 mirror (0: none, or dropped to make room for rows)."""
        out = C.c_int64(0)
        check(self.lib.vdb_table_mirror_bytes(_name(name), C.byref(out)))
        return int(out.value)

    # ---- append
    def insert(self, name: str, rows, ids) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim == 1:
            rows = rows[None, :]
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        if rows.shape[0] != ids.shape[0]:
            raise VdbError(_lib.EINVAL, "rows and ids differ in length")
        info = self.table_info(name)
        if rows.shape[0] and rows.shape[1] != info["dim"]:
            raise VdbError(_lib.EINVAL, "row dimension %d does not match the table's %d" % (rows.shape[1], info["dim"]))
        check(self.lib.vdb_insert(_name(name), rows.shape[0], rows.ctypes.data, ids.ctypes.data))

    def insert_dev(self, name: str, n: int, rows_ptr: int, ids_ptr: int, stream: int = 0) -> None:
        check(self.lib.vdb_insert_dev(_name(name), int(n), C.c_void_p(rows_ptr), C.c_void_p(ids_ptr),
                                      C.c_void_p(stream)))

    def insert_synthetic(self, name: str, n: int, seed: int, first_row: int = 0) -> None:
        check(self.lib.vdb_insert_synthetic(_name(name), int(n), int(seed), int(first_row)))

    def synth_fill_dev(self, out_ptr: int, n: int, dim: int, seed: int, first_row: int = 0, stream: int = 0) -> None:
        check(self.lib.vdb_synth_fill_dev(C.c_void_p(out_ptr), int(n), int(dim), int(seed), int(first_row),
                                          C.c_void_p(stream)))

    def fetch_rows(self, name: str, first: int, n: int) -> Tuple[np.ndarray, np.ndarray]:
        info = self.table_info(name)
        rows = np.empty((n, info["dim"]), dtype=np.float32)
        ids = np.empty((n,), dtype=np.int64)
        check(self.lib.vdb_fetch_rows(_name(name), int(first), int(n), rows.ctypes.data, ids.ctypes.data))
        return rows, ids

    # ---- snapshots
    def save_table(self, name: str, path: str) -> int:
        """Writes the table to `path`, or appends the rows that arrived since the last save of the same
        table to that path.  Returns the number of rows written."""
        written = C.c_int64(0)
        check(self.lib.vdb_save_table_rows(_name(name), os.fsencode(path), C.byref(written)))
        return int(written.value)

    def fetch_ids(self, name: str, first: int, n: int) -> np.ndarray:
        ids = np.empty((n,), dtype=np.int64)
        if n:
            check(self.lib.vdb_fetch_rows(_name(name), int(first), int(n), None, ids.ctypes.data))
        return ids

    def load_table(self, name: str, path: str) -> None:
        check(self.lib.vdb_load_table(_name(name), os.fsencode(path)))

    # ---- search
    def _dim(self, name: str) -> int:
        """Dimension of a table, kept beside its name until a table is created anywhere in the process
        (vdb_tables_created): one call without arguments per search instead of a vdb_table_info."""
        created = self.lib.vdb_tables_created()
        if created != self._dims_at:
            self._dims = {}
            self._dims_at = created
        d = self._dims.get(name)
        if d is None:
            d = self._dims[name] = self.table_info(name)["dim"]
        return d

    def search(self, name: str, queries, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """queries [nq, d] -> (ids [nq,k] int64, scores [nq,k] fp32, counts [nq] int32)."""
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        nq = queries.shape[0]
        dim = self._dim(name)
        if nq and queries.shape[1] != dim:
            raise VdbError(_lib.EINVAL, "query dimension %d does not match the table's %d" % (queries.shape[1], dim))
        k = int(k)
        # the library writes every entry of the three arrays (unused places: id -1, worst score, count 0)
        ids = np.empty((nq, k), dtype=np.int64)
        scores = np.empty((nq, k), dtype=np.float32)
        counts = np.empty((nq,), dtype=np.int32)
        if nq == 0 or k < 1:
            ids.fill(-1)
            scores.fill(0)
            counts.fill(0)
        # a ctypes view of an array's buffer costs a quarter of arr.ctypes.data (which builds a helper object per
        # call); read-only query blocks cannot give one
        try:
            qbuf = _CBuf.from_buffer(queries)
        except (TypeError, ValueError):
            qbuf = queries.ctypes.data
        check(self.lib.vdb_search(_name(name), nq, qbuf, k, _CBuf.from_buffer(ids), _CBuf.from_buffer(scores),
                                  _CBuf.from_buffer(counts)))
        return ids, scores, counts

    def search_dev(self, name: str, nq: int, queries_ptr: int, k: int, out_ids_ptr: int,
                   out_scores_ptr: int, out_counts_ptr: int, stream: int = 0) -> None:
        check(self.lib.vdb_search_dev(_name(name), int(nq), C.c_void_p(queries_ptr), int(k),
                                      C.c_void_p(out_ids_ptr), C.c_void_p(out_scores_ptr),
                                      C.c_void_p(out_counts_ptr), C.c_void_p(stream)))

    def search_dev_noalloc(self, name: str, nq: int, queries_ptr: int, k: int, out_ids_ptr: int,
                           out_scores_ptr: int, out_counts_ptr: int, stream: int = 0) -> None:
        """vdb_search_dev without any device allocation (fixed scratch inside the library)."""
        check(self.lib.vdb_search_dev_noalloc(_name(name), int(nq), C.c_void_p(queries_ptr), int(k),
                                              C.c_void_p(out_ids_ptr), C.c_void_p(out_scores_ptr),
                                              C.c_void_p(out_counts_ptr), C.c_void_p(stream)))

    def merge_topk_dev(self, nparts: int, nq: int, k: int, metric: int, in_ids_ptr: int, in_scores_ptr: int,
                       in_counts_ptr: int, out_ids_ptr: int, out_scores_ptr: int, out_counts_ptr: int,
                       stream: int = 0) -> None:
        check(self.lib.vdb_merge_topk_dev(int(nparts), int(nq), int(k), int(metric),
                                          C.c_void_p(in_ids_ptr), C.c_void_p(in_scores_ptr),
                                          C.c_void_p(in_counts_ptr), C.c_void_p(out_ids_ptr),
                                          C.c_void_p(out_scores_ptr), C.c_void_p(out_counts_ptr),
                                          C.c_void_p(stream)))

    def record_bytes(self, nq: int, k: int) -> int:
        return int(self.lib.vdb_record_bytes(int(nq), int(k)))

    def merge_records_dev(self, nparts: int, nq: int, k: int, metric: int, records_ptr: int,
                          out_ids_ptr: int, out_scores_ptr: int, out_counts_ptr: int, stream: int = 0) -> None:
        check(self.lib.vdb_merge_records_dev(int(nparts), int(nq), int(k), int(metric), C.c_void_p(records_ptr),
                                             C.c_void_p(out_ids_ptr), C.c_void_p(out_scores_ptr),
                                             C.c_void_p(out_counts_ptr), C.c_void_p(stream)))

    def set_option(self, key: str, value: float) -> None:
        check(self.lib.vdb_set_option(_name(key), float(value)))

    def bench_read_dev(self, buf_ptr: int, nbytes: int, nt: int, blocks: int, threads: int, stream: int = 0) -> None:
        check(self.lib.vdb_bench_read_dev(C.c_void_p(buf_ptr), int(nbytes), int(nt), int(blocks), int(threads),
                                          C.c_void_p(stream)))

    # ---- measurement
    def prof_enable(self, on) -> None:
        """False/0 off, True/1 every launch of the dominant kernel timed, n > 1 every n-th launch."""
        check(self.lib.vdb_prof_enable(int(on)))

    def prof_read(self) -> Tuple[float, int]:
        ms, n = C.c_double(0.0), C.c_int64(0)
        check(self.lib.vdb_prof_read(C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def stats(self) -> Tuple[int, int, int]:
        """(queries answered, queries that took the exact second pass, queries on the batched path)"""
        q, f, b = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.vdb_stats(C.byref(q), C.byref(f), C.byref(b)))
        return int(q.value), int(f.value), int(b.value)

    def shutdown(self) -> None:
        check(self.lib.vdb_shutdown())
