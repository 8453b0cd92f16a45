"""Row-sharded search across the GPUs of one node: one process per GPU, each holding a contiguous
range of rows in its own HBM table.  A query is answered in three steps:

  1. every rank runs the local scan + top-k on its shard (vdb_search_dev) and writes its k hits
     into a small per-rank record (ids | scores | counts, vdb_record_bytes);
  2. ONE all-gather (RCCL over xGMI via torch.distributed) exchanges the records: 12*k*nq + 4*nq
     bytes per rank, i.e. latency-bound, so the three arrays travel in one collective;
  3. every rank merges the world_size records with the same (score, id) order
     (vdb_merge_records_dev) and holds the global answer.

Top-k is a decomposable reduction (the global top-k is contained in the union of the local
top-k lists), so the result is identical to a single-table search.

The two compute steps are injected as callables so that the partitioning, record layout and
collective can be exercised with the `gloo` backend on CPU tensors in tests; `gpu_backend`
below supplies the real ones and is what bench.py and applications use.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range [first, last) of `rank`."""
    if world < 1 or not (0 <= rank < world) or n_rows < 0:
        raise ValueError("bad shard request")
    return rank * n_rows // world, (rank + 1) * n_rows // world


def record_bytes(nq: int, k: int) -> int:
    """Same layout rule as vdb_record_bytes (include/vdb.h)."""
    return (nq * k * 12 + nq * 4 + 15) // 16 * 16


@dataclass
class RecordViews:
    ids: torch.Tensor       # [nq, k] int64
    scores: torch.Tensor    # [nq, k] float32
    counts: torch.Tensor    # [nq] int32


def record_views(rec: torch.Tensor, nq: int, k: int) -> RecordViews:
    """Typed views into one uint8 record."""
    assert rec.dtype == torch.uint8 and rec.numel() >= record_bytes(nq, k)
    a, b = nq * k * 8, nq * k * 12
    return RecordViews(ids=rec[:a].view(torch.int64).view(nq, k),
                       scores=rec[a:b].view(torch.float32).view(nq, k),
                       counts=rec[b:b + nq * 4].view(torch.int32))


_comm_streams = {}


def comm_stream(device: torch.device) -> "torch.cuda.Stream":
    """The one stream of this process on which collectives over device tensors are issued.  Queries are
    pipelined over several streams, and a communicator wants its operations in one order on every rank:
    issued from one stream, in program order, they are, whatever stream the caller was on."""
    key = (device.type, device.index)
    s = _comm_streams.get(key)
    if s is None:
        s = _comm_streams[key] = torch.cuda.Stream(device=device)
    return s


LocalSearch = Callable[[torch.Tensor, RecordViews], None]      # (queries [nq,d], record views)
Merge = Callable[[torch.Tensor, int, RecordViews], None]       # (gathered records, world, out views)


class ShardedSearcher:
    def __init__(self, nq: int, k: int, device: torch.device, local_search: LocalSearch, merge: Merge,
                 group: Optional[dist.ProcessGroup] = None):
        self.nq, self.k = nq, k
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_search, self.merge = local_search, merge
        nb = record_bytes(nq, k)
        # allocated once: the query loop allocates nothing
        self.rec = torch.zeros(nb, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(nb * self.world, dtype=torch.uint8, device=device)
        self.out = torch.zeros(nb, dtype=torch.uint8, device=device)
        self.rec_views = record_views(self.rec, nq, k)
        self.out_views = record_views(self.out, nq, k)
        # when a list: one (local search, collective, merge) triple of host microseconds per call, the time
        # each of the three steps took to ENQUEUE (with `trace_sync` set, to complete on the device)
        self.trace: Optional[list] = None
        self.trace_sync = False
        self.force_collective = False      # a world of one rank goes through the collective as well (check tools)

    def _mark(self) -> int:
        import time
        if self.trace_sync and self.rec.is_cuda:
            torch.cuda.current_stream().synchronize()
        return time.perf_counter_ns()

    def search(self, queries: torch.Tensor) -> RecordViews:
        """queries [nq, d] fp32 on this rank's device, identical on every rank."""
        t0 = self._mark() if self.trace is not None else 0
        self.local_search(queries, self.rec_views)
        t1 = self._mark() if self.trace is not None else 0
        if self.world == 1 and not self.force_collective:
            if self.trace is not None:
                self.trace.append(((t1 - t0) / 1e3, 0.0, 0.0))
            return self.rec_views
        if self.rec.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal of the N > 1 path without RCCL (ranks sharing one GPU): gloo moves host
            # memory, so the record is staged through the host around the same collective
            host = torch.empty(self.gathered.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(host, self.rec.cpu(), group=self.group)
            self.gathered.copy_(host)
        elif self.rec.is_cuda:
            # on the process's collective stream, between the caller's local search and its merge
            cur = torch.cuda.current_stream(self.rec.device)
            cs = comm_stream(self.rec.device)
            cs.wait_stream(cur)
            with torch.cuda.stream(cs):
                dist.all_gather_into_tensor(self.gathered, self.rec, group=self.group)
            cur.wait_stream(cs)
        else:
            dist.all_gather_into_tensor(self.gathered, self.rec, group=self.group)
        t2 = self._mark() if self.trace is not None else 0
        self.merge(self.gathered, self.world, self.out_views)
        if self.trace is not None:
            t3 = self._mark()
            self.trace.append(((t1 - t0) / 1e3, (t2 - t1) / 1e3, (t3 - t2) / 1e3))
        return self.out_views


def gpu_backend(engine, table: str, nq: int, k: int, metric: int) -> Tuple[LocalSearch, Merge]:
    """The product path: both steps run in libvdb.so on the current HIP stream."""

    def local_search(queries: torch.Tensor, rec: RecordViews) -> None:
        st = torch.cuda.current_stream().cuda_stream
        engine.search_dev(table, nq, queries.data_ptr(), k, rec.ids.data_ptr(), rec.scores.data_ptr(),
                          rec.counts.data_ptr(), st)

    def merge(gathered: torch.Tensor, world: int, out: RecordViews) -> None:
        st = torch.cuda.current_stream().cuda_stream
        engine.merge_records_dev(world, nq, k, metric, gathered.data_ptr(), out.ids.data_ptr(),
                                 out.scores.data_ptr(), out.counts.data_ptr(), st)

    return local_search, merge
