"""One engine client over several GPUs: the devices of gpu_resource_config.search_resources
(/root/reference/server_config.yaml:43-44), which the engine the reference configures uses from one
server process.

Here every GPU gets its own worker process (the library binds a process to one device), each
holding one row shard of every table behind the ordinary in-process client.  `ClusterMilvus` has
the surface of that client (the seven calls of /root/reference/app.py plus the additions) and

  * sends table lifecycle calls to every worker,
  * appends new rows to the least-filled shard (bulk inserts are split evenly),
  * sends a search to all workers at once; every worker searches its shard on its GPU, the workers
    exchange their k hits per query in ONE all-gather between the GPUs (torch.distributed: RCCL over
    xGMI; the record layout and the merge kernel of vectordb_amd/sharded.py) and merge them on the
    device by (score best first, id ascending) -- the order each shard already uses, so the merged
    top-k equals the answer of one table holding all rows.  Worker 0 hands the merged answer back.

engine_config.cluster_backend chooses how the workers exchange their hits: "nccl" (RCCL; the default
when every worker has a GPU of its own), "gloo" (the same collective through host memory: the default
when workers share a GPU, which RCCL does not allow, i.e. the rehearsal on one device) or "host" (no
process group: the parent merges the per-shard answers it receives over the pipes).
"""
from __future__ import annotations

import dataclasses
import multiprocessing as mp
import os
import threading
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .coalesce import SearchCoalescer
from .config import EngineConfig, load_config
from .milvus import MetricType, QueryResult, Status


class _CollectiveSearch:
    """Worker side of a search over all shards: local search into a record, all-gather of the records,
    merge on the device (ShardedSearcher).  A worker whose local search fails still takes part in the
    collective, with an empty record, and reports the failure afterwards: the other workers are never
    left waiting."""

    def __init__(self, client, device: int, rank: int, world: int):
        import torch
        from .sharded import ShardedSearcher
        self._torch, self._make = torch, ShardedSearcher
        self.client, self.rank, self.world = client, rank, world
        self.dev = torch.device("cuda", device)
        self.searchers: Dict[Tuple[int, int], Any] = {}
        self.table, self.metric, self.err = "", 0, None

    def _searcher(self, nq: int, k: int):
        s = self.searchers.get((nq, k))
        if s is None:
            if len(self.searchers) >= 32:           # callers merged in front of the workers come in many sizes
                self.searchers.clear()
            eng = self.client._eng()

            def local_search(queries, rec) -> None:
                st = self._torch.cuda.current_stream().cuda_stream
                try:
                    eng.search_dev(self.table, nq, queries.data_ptr(), k, rec.ids.data_ptr(), rec.scores.data_ptr(),
                                   rec.counts.data_ptr(), st)
                except Exception as exc:
                    self.err = exc
                    rec.counts.zero_()

            def merge(gathered, world, out) -> None:
                st = self._torch.cuda.current_stream().cuda_stream
                eng.merge_records_dev(world, nq, k, self.metric, gathered.data_ptr(), out.ids.data_ptr(),
                                      out.scores.data_ptr(), out.counts.data_ptr(), st)

            s = self.searchers[(nq, k)] = self._make(nq, k, self.dev, local_search, merge)
        return s

    def search(self, name: str, q: np.ndarray, k: int):
        """-> (code, message, ids, scores, counts); the arrays only on worker 0."""
        torch = self._torch
        eng = self.client._eng()
        nq = int(q.shape[0])
        self.err = None
        try:
            info = eng.table_info(name)
            self.table, self.metric = name, int(info["metric"])
            if nq and q.shape[1] != info["dim"]:
                raise _lib.VdbError(_lib.EINVAL, "query dimension %d does not match the table's %d" % (q.shape[1], info["dim"]))
        except Exception as exc:
            self.err = exc
        with torch.cuda.device(self.dev):
            qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(self.dev)
            s = self._searcher(nq, k)
            if self.err is not None:
                # no table to search here: an empty record, so that the collective below still has every worker in it
                s.rec_views.counts.zero_()
                saved, s.local_search = s.local_search, (lambda queries, rec: None)
                try:
                    out = s.search(qd)
                finally:
                    s.local_search = saved
            else:
                out = s.search(qd)
            if self.err is not None:
                torch.cuda.current_stream().synchronize()
                st = self.client._status(self.err)
                return (st.code, st.message, None, None, None)
            if self.rank != 0:
                torch.cuda.current_stream().synchronize()
                return (0, "Success", None, None, None)
            return (0, "Success", out.ids.cpu().numpy(), out.scores.cpu().numpy(), out.counts.cpu().numpy())


def _worker_main(conn, device: int, cfg: EngineConfig, rank: int = 0, world: int = 1, backend: str = "host",
                 port: int = 0) -> None:
    """Runs in the worker process: one in-process client on one GPU, requests over a pipe."""
    os.environ["VDB_DEVICE"] = str(device)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # RCCL's IPC handles need the dmabuf path on this driver
    if backend != "host" and world > 1:
        # before the library is loaded: torch ships a HIP runtime of its own, and the one a process loads
        # first serves both; torch finds no device behind the other one
        import torch  # noqa: F401
    from .milvus import Milvus                      # imported here: the parent never loads the library
    client = Milvus(cfg)
    status = client.connect()
    collective = None
    if status.OK() and backend != "host" and world > 1:
        try:
            import datetime
            import torch.distributed as dist
            torch.cuda.set_device(device)
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(port)
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
            collective = _CollectiveSearch(client, device, rank, world)
        except Exception as exc:
            status = Status(_lib.ESTATE, "process group (%s) of the engine workers: %s: %s" % (backend, type(exc).__name__, exc))
    conn.send((status.code, status.message))
    if not status.OK():
        return
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            break
        if msg is None:
            break
        method, args, kwargs = msg
        try:
            if method == "search_collective":
                name, q, k = args
                try:
                    conn.send(collective.search(name, q, int(k)))
                except Exception as exc:
                    conn.send((_lib.EHIP, "%s: %s" % (type(exc).__name__, exc), None, None, None))
                continue
            if method == "search_arrays":
                # search without building result objects: (code, message, ids, scores, counts)
                name, q, k = args
                try:
                    ids, sc, cnt = client._eng().search(name, q, k)
                    conn.send((0, "Success", ids, sc, cnt))
                except Exception as exc:            # reported, never raised across the pipe
                    st = client._status(exc)
                    conn.send((st.code, st.message, None, None, None))
                continue
            out = getattr(client, method)(*args, **kwargs)
            if isinstance(out, tuple):
                conn.send((out[0].code, out[0].message) + tuple(out[1:]))
            else:
                conn.send((out.code, out.message))
        except Exception as exc:
            conn.send((_lib.EINVAL, "%s: %s" % (type(exc).__name__, exc)))
    if collective is not None:
        try:
            import torch.distributed as dist
            dist.destroy_process_group()
        except Exception:
            pass


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


class ClusterMilvus:
    def __init__(self, config: Optional[EngineConfig] = None, devices: Optional[Sequence[int]] = None):
        self._config = config or load_config()
        self._devices = list(devices) if devices is not None else list(self._config.search_resources)
        self._workers: List[Tuple[Any, Any]] = []            # (process, pipe)
        self._lock = threading.Lock()                       # one request at a time per pipe set
        self._metric: Dict[str, int] = {}
        self._rows: Dict[str, List[int]] = {}                # rows per shard, for least-filled placement
        self._coalescer = SearchCoalescer(self._search_arrays) if self._config.coalesce_searches else None
        # how the workers exchange their hits (module docstring); RCCL wants one GPU per rank
        backend = os.environ.get("VDB_CLUSTER_BACKEND") or self._config.cluster_backend
        if not backend:
            backend = "nccl" if len(set(self._devices)) == len(self._devices) else "gloo"
        if backend not in ("nccl", "gloo", "host"):
            raise ValueError("engine_config.cluster_backend must be nccl, gloo or host: %r" % (backend,))
        self._backend = backend if len(self._devices) > 1 else "host"

    # -- plumbing ---------------------------------------------------------------------------
    @staticmethod
    def _recv(pipe) -> tuple:
        try:
            return pipe.recv()
        except (EOFError, OSError) as exc:          # the worker process is gone
            return (_lib.ESTATE, "engine worker not reachable: %s" % type(exc).__name__, None, None, None)

    def _call_all(self, method: str, *args, **kwargs) -> List[tuple]:
        for _, pipe in self._workers:
            try:
                pipe.send((method, args, kwargs))
            except (OSError, ValueError):
                pass                                # reported by the failing receive below
        return [self._recv(pipe) for _, pipe in self._workers]

    @staticmethod
    def _first_failure(replies: List[tuple]) -> Status:
        for r in replies:
            if r[0] != 0:
                return Status(r[0], r[1])
        return Status()

    # -- connection -------------------------------------------------------------------------
    def connect(self, host: Optional[str] = None, port: Optional[str] = None, uri: Optional[str] = None,
                timeout: Optional[int] = None) -> Status:
        with self._lock:
            if self._workers:
                return Status()
            if not self._devices:
                return Status(_lib.EINVAL, "no devices in gpu_resource_config.search_resources")
            ctx = mp.get_context("spawn")                    # never fork a process that may hold a GPU context
            port = _free_port() if self._backend != "host" else 0
            for shard, dev in enumerate(self._devices):
                parent, child = ctx.Pipe()
                # every shard keeps its snapshots in a directory of its own under primary_path
                cfg = dataclasses.replace(self._config, primary_path=os.path.join(self._config.primary_path, "shard%d" % shard))
                proc = ctx.Process(target=_worker_main, args=(child, dev, cfg, shard, len(self._devices), self._backend, port),
                                   daemon=True)
                proc.start()
                child.close()
                self._workers.append((proc, parent))
            replies = [self._recv(pipe) for _, pipe in self._workers]
            st = self._first_failure(replies)
            if not st.OK():
                self._stop()
            return st

    def connected(self) -> bool:
        return bool(self._workers)

    def _stop(self) -> None:
        for proc, pipe in self._workers:
            try:
                pipe.send(None)
            except (OSError, ValueError):
                pass
        for proc, pipe in self._workers:
            proc.join(timeout=10)
            if proc.is_alive():
                proc.terminate()
            pipe.close()
        self._workers = []

    def disconnect(self) -> Status:
        with self._lock:
            self._stop()
        return Status()

    def server_version(self) -> Tuple[Status, str]:
        return Status(), "vectordb_amd cluster client (%d GPUs)" % len(self._devices)

    def _need(self) -> Optional[Status]:
        return None if self._workers else Status(_lib.ESTATE, "not connected")

    # -- table lifecycle ---------------------------------------------------------------------
    def has_table(self, table_name: str) -> Tuple[Status, bool]:
        with self._lock:
            bad = self._need()
            if bad:
                return bad, False
            replies = self._call_all("has_table", table_name)
            st = self._first_failure(replies)
            return st, st.OK() and all(r[2] for r in replies)

    def create_table(self, param: Dict[str, Any]) -> Status:
        with self._lock:
            bad = self._need()
            if bad:
                return bad
            st = self._first_failure(self._call_all("create_table", param))
            if st.OK():
                self._metric[param["table_name"]] = int(param.get("metric_type", MetricType.L2))
                self._rows[param["table_name"]] = [0] * len(self._workers)
            return st

    def create_index(self, table_name: str, index: Optional[Dict[str, Any]] = None, timeout: int = -1) -> Status:
        with self._lock:
            return self._need() or self._first_failure(self._call_all("create_index", table_name, index))

    def delete_table(self, table_name: str) -> Status:
        with self._lock:
            bad = self._need()
            if bad:
                return bad
            st = self._first_failure(self._call_all("delete_table", table_name))
            self._metric.pop(table_name, None)
            self._rows.pop(table_name, None)
            return st

    drop_table = delete_table

    def get_table_row_count(self, table_name: str) -> Tuple[Status, int]:
        with self._lock:
            bad = self._need()
            if bad:
                return bad, 0
            replies = self._call_all("get_table_row_count", table_name)
            st = self._first_failure(replies)
            return st, sum(int(r[2]) for r in replies) if st.OK() else 0

    def _table_state(self, table_name: str) -> Optional[Status]:
        """Metric and per-shard row counts of a table this client did not create (preloaded)."""
        if table_name in self._metric:
            return None
        replies = self._call_all("describe_table", table_name)
        st = self._first_failure(replies)
        if not st.OK():
            return st
        self._metric[table_name] = int(replies[0][2]["metric_type"])
        self._rows[table_name] = [int(r[2]["row_count"]) for r in replies]
        return None

    # -- append ----------------------------------------------------------------------------
    def insert(self, table_name: str, records: Sequence[Sequence[float]], ids: Optional[Sequence[int]] = None,
               timeout: int = -1) -> Tuple[Status, List[int]]:
        with self._lock:
            bad = self._need() or self._table_state(table_name)
            if bad:
                return bad, []
            rows = np.asarray(records, dtype=np.float32)
            if rows.ndim != 2:
                return Status(_lib.EINVAL, "records must be a list of equal-length vectors"), []
            counts = self._rows[table_name]
            if ids is None:
                base = sum(counts)
                ids = list(range(base, base + rows.shape[0]))
            ids = np.asarray(ids, dtype=np.int64)
            if ids.shape[0] != rows.shape[0]:
                return Status(_lib.EINVAL, "records and ids differ in length"), []
            # fill the least-filled shards first, level after that
            n, w = rows.shape[0], len(self._workers)
            target = (sum(counts) + n + w - 1) // w
            take = [0] * w
            left = n
            for s in sorted(range(w), key=lambda i: counts[i]):
                t = max(0, min(left, target - counts[s]))
                take[s] = t
                left -= t
            take[min(range(w), key=lambda i: counts[i] + take[i])] += left
            pos, sent = 0, []
            for s in range(w):
                if take[s]:
                    self._workers[s][1].send(("insert", (table_name, rows[pos:pos + take[s]], ids[pos:pos + take[s]]), {}))
                    sent.append(s)
                    pos += take[s]
            st = Status()
            for s in sent:
                r = self._recv(self._workers[s][1])
                if r[0] != 0:
                    st = Status(r[0], r[1])
                else:
                    counts[s] += take[s]
            return st, [int(i) for i in ids] if st.OK() else []

    add_vectors = insert

    # -- search ----------------------------------------------------------------------------
    def _search_arrays(self, table_name: str, q: np.ndarray, k: int):
        """All shards at once: (ids [nq, k], scores [nq, k], counts [nq]), merged by the workers on their GPUs
        after one all-gather, or (cluster_backend "host") here from the per-shard answers."""
        with self._lock:
            bad = self._need() or self._table_state(table_name)
            if bad:
                raise _lib.VdbError(bad.code, bad.message)
            if self._backend != "host" and q.shape[0] > 0:
                replies = self._call_all("search_collective", table_name, q, k)
                st = self._first_failure(replies)
                if not st.OK():
                    raise _lib.VdbError(st.code, st.message)
                return replies[0][2], replies[0][3], replies[0][4]
            replies = self._call_all("search_arrays", table_name, q, k)
            st = self._first_failure(replies)
            if not st.OK():
                raise _lib.VdbError(st.code, st.message)
            l2 = self._metric[table_name] == int(MetricType.L2)
        nq = q.shape[0]
        out_ids = np.full((nq, k), -1, dtype=np.int64)
        out_sc = np.full((nq, k), np.inf if l2 else -np.inf, dtype=np.float32)
        out_cnt = np.zeros((nq,), dtype=np.int32)
        for qi in range(nq):
            ids = np.concatenate([r[2][qi, :int(r[4][qi])] for r in replies])
            sc = np.concatenate([r[3][qi, :int(r[4][qi])] for r in replies])
            order = np.lexsort((ids, sc if l2 else -sc))[:k]
            out_ids[qi, :len(order)] = ids[order]
            out_sc[qi, :len(order)] = sc[order]
            out_cnt[qi] = len(order)
        return out_ids, out_sc, out_cnt

    def search_vectors(self, table_name: str, top_k: int, nprobe: int = 0,
                       query_records: Optional[Sequence[Sequence[float]]] = None,
                       query_ranges=None) -> Tuple[Status, List[List[QueryResult]]]:
        q = np.asarray(query_records, dtype=np.float32)
        if q.ndim != 2:
            return Status(_lib.EINVAL, "query_records must be a list of equal-length vectors"), []
        try:
            # concurrent callers are merged into one call to the workers (vectordb_amd/coalesce.py)
            fn = self._coalescer.search if self._coalescer is not None else self._search_arrays
            ids, sc, cnt = fn(table_name, q, int(top_k))
        except _lib.VdbError as exc:
            return Status(exc.code, exc.message), []
        return Status(), [[QueryResult(int(ids[i, j]), float(sc[i, j])) for j in range(int(cnt[i]))]
                          for i in range(q.shape[0])]

    search = search_vectors

    # -- snapshots -------------------------------------------------------------------------
    def snapshot(self, table_name: str) -> Status:
        """Every worker writes its shard under <primary_path>/shard<i>/ (and preloads it from there)."""
        with self._lock:
            return self._need() or self._first_failure(self._call_all("snapshot", table_name))

    def preload_table(self, table_name: str, timeout: Optional[int] = None) -> Status:
        with self._lock:
            return self._need() or self._first_failure(self._call_all("preload_table", table_name))

    def table_ids(self, table_name: str) -> Tuple[Status, np.ndarray]:
        with self._lock:
            bad = self._need()
            if bad:
                return bad, np.empty((0,), dtype=np.int64)
            replies = self._call_all("table_ids", table_name)
            st = self._first_failure(replies)
            if not st.OK():
                return st, np.empty((0,), dtype=np.int64)
            return st, np.concatenate([np.asarray(r[2], dtype=np.int64) for r in replies])
