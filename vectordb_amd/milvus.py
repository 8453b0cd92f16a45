"""Drop-in for the vector-engine client the reference imports (`from milvus import Milvus,
IndexType, MetricType`, /root/reference/app.py:3), backed by libvdb.so in this process.

Mirrors exactly what app.py calls and relies on:

  Milvus()                                   app.py:172
  .connect(host=, port=) -> status.OK()      app.py:176-178
  .has_table(name) -> (status, bool)         app.py:120
  .create_table({'table_name','dimension','index_file_size','metric_type'})   app.py:137-143
  .create_index(name, {'index_type','nlist'})                                 app.py:145-149
  .delete_table(name) -> status.code / .message                               app.py:242-245
  .insert(name, records=[[...]], ids=[...])                                   app.py:325
  .search_vectors(table_name=, query_records=, top_k=, nprobe=) -> (status, results);
      results[i] iterates best first over objects with .id and .distance      app.py:384-405

No method raises: failures come back in the status, as the caller expects.
"""
from __future__ import annotations

import enum
import os
import re
import threading
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import VdbError
from .config import EngineConfig, load_config
from .engine import Engine


class MetricType(enum.IntEnum):
    INVALID = 0
    L2 = 1
    IP = 2
    COSINE = 3        # addition of this engine: rows normalised on insert, query on search


class IndexType(enum.IntEnum):
    INVALID = 0
    FLAT = 1
    IVFLAT = 2
    IVF_SQ8 = 3
    MIX_NSG = 4
    IVF_SQ8H = 5


class Status:
    SUCCESS = 0

    def __init__(self, code: int = 0, message: str = "Success"):
        self.code = code
        self.message = message

    def OK(self) -> bool:
        return self.code == Status.SUCCESS

    def __repr__(self) -> str:
        return "Status(code=%d, message=%r)" % (self.code, self.message)

    def __eq__(self, other) -> bool:
        if isinstance(other, int):
            return self.code == other
        return isinstance(other, Status) and self.code == other.code


class QueryResult:
    __slots__ = ("id", "distance")

    def __init__(self, id: int, distance: float):
        self.id = id
        self.distance = distance

    def __repr__(self) -> str:
        return "QueryResult(id=%d, distance=%r)" % (self.id, self.distance)

    __str__ = __repr__


_DTYPE_BY_NAME = {"f32": _lib.F32, "fp32": _lib.F32, "float32": _lib.F32,
                  "bf16": _lib.BF16, "bfloat16": _lib.BF16,
                  "f16": _lib.F16, "fp16": _lib.F16, "float16": _lib.F16}

_engine_lock = threading.Lock()
_engine: Optional[Engine] = None

SNAPSHOT_SUFFIX = ".vdbsnap"
_TABLE_FILE_RE = re.compile(r"^[A-Za-z0-9_][A-Za-z0-9_.-]{0,254}$")


def snapshot_path(primary_path: str, table_name: str) -> str:
    """File of `table_name` under db_config.primary_path (/root/reference/server_config.yaml:12).
    Names that could leave the directory are refused."""
    if not _TABLE_FILE_RE.match(table_name) or ".." in table_name:
        raise VdbError(_lib.EINVAL, "table name cannot be used as a snapshot file name: %r" % (table_name,))
    return os.path.join(primary_path, table_name + SNAPSHOT_SUFFIX)


def _preload(engine: Engine, cfg: EngineConfig) -> List[str]:
    """db_config.preload_table (/root/reference/server_config.yaml:22): '*' loads every snapshot
    in primary_path, otherwise the listed tables.  Returns the names loaded; a listed table
    without a snapshot is skipped (a fresh deployment has none yet)."""
    if not cfg.preload_table or not os.path.isdir(cfg.primary_path):
        return []
    if "*" in cfg.preload_table:
        names = sorted(f[:-len(SNAPSHOT_SUFFIX)] for f in os.listdir(cfg.primary_path)
                       if f.endswith(SNAPSHOT_SUFFIX))
    else:
        names = list(cfg.preload_table)
    loaded = []
    for name in names:
        path = snapshot_path(cfg.primary_path, name)
        if os.path.exists(path) and not engine.has_table(name):
            engine.load_table(name, path)
            with _snap_lock:
                _snap_rows[name] = engine.count(name)
            loaded.append(name)
    return loaded


def _get_engine(config: Optional[EngineConfig]) -> Engine:
    global _engine
    with _engine_lock:
        if _engine is None:
            cfg = config or load_config()
            if not cfg.gpu_enable:
                raise VdbError(_lib.ENOGPU, "gpu_resource_config.enable is false and there is no CPU engine")
            rank = int(os.environ.get("LOCAL_RANK", "0"))
            device = int(os.environ.get("VDB_DEVICE", cfg.device_for_rank(rank)))
            engine = Engine(device=device, config=cfg)
            _preload(engine, cfg)
            _engine = engine
        return _engine


# rows each table held at its last snapshot (process-wide, like the engine), for snapshot_every_rows
_snap_lock = threading.Lock()
_snap_rows: Dict[str, int] = {}


class Milvus:
    def __init__(self, config: Optional[EngineConfig] = None):
        self._config = config
        self._engine: Optional[Engine] = None
        self.last_snapshot_rows_written = 0

    # -- connection -------------------------------------------------------------------------
    def connect(self, host: Optional[str] = None, port: Optional[str] = None, uri: Optional[str] = None,
                timeout: Optional[int] = None) -> Status:
        # host/port named the external engine's gRPC endpoint; the engine now lives in-process
        try:
            self._engine = _get_engine(self._config)
        except (VdbError, ImportError, OSError, ValueError) as exc:
            return Status(getattr(exc, "code", _lib.ENOGPU), str(exc))
        return Status()

    def connected(self) -> bool:
        return self._engine is not None

    def disconnect(self) -> Status:
        self._engine = None
        return Status()

    def server_version(self) -> Tuple[Status, str]:
        return Status(), "vectordb_amd (MI355X brute-force engine)"

    def _eng(self) -> Engine:
        if self._engine is None:
            raise VdbError(_lib.ESTATE, "not connected")
        return self._engine

    @staticmethod
    def _status(exc: Exception) -> Status:
        if isinstance(exc, VdbError):
            return Status(exc.code, exc.message)
        return Status(_lib.EINVAL, "%s: %s" % (type(exc).__name__, exc))

    # -- table lifecycle ---------------------------------------------------------------------
    def has_table(self, table_name: str) -> Tuple[Status, bool]:
        try:
            return Status(), self._eng().has_table(table_name)
        except Exception as exc:
            return self._status(exc), False

    def create_table(self, param: Dict[str, Any]) -> Status:
        try:
            name = param["table_name"]
            dim = int(param["dimension"])
            metric = int(param.get("metric_type", MetricType.L2))
            dtype = param.get("dtype", _lib.F32)            # addition: storage dtype of the rows
            if isinstance(dtype, str):
                dtype = _DTYPE_BY_NAME[dtype.lower()]
            # index_file_size sized the external engine's data files; one HBM matrix has no files
            self._eng().create_table(name, dim, metric, int(dtype))
            return Status()
        except Exception as exc:
            return self._status(exc)

    def create_index(self, table_name: str, index: Optional[Dict[str, Any]] = None, timeout: int = -1) -> Status:
        try:
            index = index or {}
            self._eng().create_index(table_name, int(index.get("index_type", IndexType.FLAT)),
                                     int(index.get("nlist", 0)))
            return Status()
        except Exception as exc:
            return self._status(exc)

    def delete_table(self, table_name: str) -> Status:
        try:
            eng = self._eng()
            eng.drop_table(table_name)
            with _snap_lock:
                _snap_rows.pop(table_name, None)
            # a dropped table must not come back at the next start
            try:
                os.remove(snapshot_path(eng.config.primary_path, table_name))
            except (OSError, VdbError):
                pass
            return Status()
        except Exception as exc:
            return self._status(exc)

    drop_table = delete_table

    def get_table_row_count(self, table_name: str) -> Tuple[Status, int]:
        try:
            return Status(), self._eng().count(table_name)
        except Exception as exc:
            return self._status(exc), 0

    def describe_table(self, table_name: str) -> Tuple[Status, Optional[Dict[str, Any]]]:
        try:
            info = self._eng().table_info(table_name)
            return Status(), {"table_name": table_name, "dimension": info["dim"],
                              "metric_type": MetricType(info["metric"]), "dtype": info["dtype"],
                              "row_count": info["rows"]}
        except Exception as exc:
            return self._status(exc), None

    # -- snapshots -------------------------------------------------------------------------
    def snapshot(self, table_name: str) -> Status:
        """Write `table_name` to db_config.primary_path so that a restarted process can preload it
        (the external engine persisted every insert to that directory by itself).  A table that already
        has a snapshot there gets only its new rows appended; `last_snapshot_rows_written` says how many."""
        try:
            eng = self._eng()
            os.makedirs(eng.config.primary_path, exist_ok=True)
            rows = eng.count(table_name)
            self.last_snapshot_rows_written = eng.save_table(table_name, snapshot_path(eng.config.primary_path, table_name))
            with _snap_lock:
                _snap_rows[table_name] = max(rows, _snap_rows.get(table_name, 0))
            return Status()
        except Exception as exc:
            return self._status(exc)

    def _maybe_snapshot(self, table_name: str) -> None:
        """engine_config.snapshot_every_rows: called after an insert went through."""
        eng = self._eng()
        every = int(getattr(eng.config, "snapshot_every_rows", 0) or 0)
        if every <= 0:
            return
        rows = eng.count(table_name)
        with _snap_lock:
            due = rows - _snap_rows.get(table_name, 0) >= every
            if due:
                _snap_rows[table_name] = rows          # one caller writes; the others go on
        if due:
            st = self.snapshot(table_name)
            if not st.OK():
                print("snapshot of %s failed: %s" % (table_name, st.message))

    def table_ids(self, table_name: str) -> Tuple[Status, np.ndarray]:
        """ids of every stored row (what /checkintegrity compares with the relational side)."""
        try:
            eng = self._eng()
            n = eng.count(table_name)
            parts = [eng.fetch_ids(table_name, i, min(1 << 22, n - i)) for i in range(0, n, 1 << 22)]
            return Status(), (np.concatenate(parts) if parts else np.empty((0,), dtype=np.int64))
        except Exception as exc:
            return self._status(exc), np.empty((0,), dtype=np.int64)

    def preload_table(self, table_name: str, timeout: Optional[int] = None) -> Status:
        """Bring `table_name` into HBM from its snapshot unless it is already resident."""
        try:
            eng = self._eng()
            if not eng.has_table(table_name):
                eng.load_table(table_name, snapshot_path(eng.config.primary_path, table_name))
                with _snap_lock:
                    _snap_rows[table_name] = eng.count(table_name)
            return Status()
        except Exception as exc:
            return self._status(exc)

    # -- append ----------------------------------------------------------------------------
    def insert(self, table_name: str, records: Sequence[Sequence[float]], ids: Optional[Sequence[int]] = None,
               timeout: int = -1) -> Tuple[Status, List[int]]:
        try:
            eng = self._eng()
            # JSON floats arrive as float64; the engine boundary is where they become fp32
            rows = np.asarray(records, dtype=np.float32)
            if rows.ndim != 2:
                raise VdbError(_lib.EINVAL, "records must be a list of equal-length vectors")
            if ids is None:
                base = eng.count(table_name)
                ids = list(range(base, base + rows.shape[0]))
            eng.insert(table_name, rows, np.asarray(ids, dtype=np.int64))
            self._maybe_snapshot(table_name)
            return Status(), [int(i) for i in ids]
        except Exception as exc:
            return self._status(exc), []

    add_vectors = insert

    # -- search ----------------------------------------------------------------------------
    def search_vectors(self, table_name: str, top_k: int, nprobe: int = 0,
                       query_records: Optional[Sequence[Sequence[float]]] = None,
                       query_ranges=None) -> Tuple[Status, List[List[QueryResult]]]:
        # nprobe selected clusters of the approximate index; every row is scanned here
        try:
            q = np.asarray(query_records, dtype=np.float32)
            if q.ndim != 2:
                raise VdbError(_lib.EINVAL, "query_records must be a list of equal-length vectors")
            # concurrent callers (one thread per HTTP request) are merged inside vdb_search
            ids, scores, counts = self._eng().search(table_name, q, int(top_k))
            results = [[QueryResult(int(ids[i, j]), float(scores[i, j])) for j in range(int(counts[i]))]
                       for i in range(q.shape[0])]
            return Status(), results
        except Exception as exc:
            return self._status(exc), []

    search = search_vectors
