"""Multi-process test of the row-sharded search on CPU (gloo, world sizes 2 and 3): partitioning, the
per-rank record layout, the single all-gather and the merge call pattern of
vectordb_amd/sharded.py.  The two compute steps are the injection points of ShardedSearcher; here
they are stood in by the oracle (the product's `gpu_backend` needs a GPU and is covered by the
GPU tests), so what is under test is the host-side exchange logic, against a global oracle search."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, metric_name, ret):
    import cases
    from oracle import oracle as O
    from vectordb_amd.sharded import RecordViews, ShardedSearcher, record_bytes, record_views, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, d, k, nq = 1003, 24, 10, 3
        metric = cases.METRICS[metric_name]
        rows = cases.make_rows("ties", 42, n, d)
        ids = cases.make_ids("hash", 42, n)
        q = cases.make_queries("ties", 42, nq, d, rows)
        stored = O.prepare_store(rows, metric)
        lo, hi = shard_range(n, world, rank)

        def local_search(queries: torch.Tensor, rec: RecordViews) -> None:
            res = O.search(stored[lo:hi], ids[lo:hi], queries.numpy(), k, metric)
            rec.ids.fill_(-1)
            for qi, (ri, rs) in enumerate(res):
                rec.ids[qi, :len(ri)] = torch.from_numpy(ri)
                rec.scores[qi, :len(rs)] = torch.from_numpy(rs)
                rec.counts[qi] = len(ri)

        def merge(gathered: torch.Tensor, w: int, out: RecordViews) -> None:
            nb = record_bytes(nq, k)
            parts = [record_views(gathered[i * nb:(i + 1) * nb], nq, k) for i in range(w)]
            for qi in range(nq):
                pi = np.concatenate([p.ids[qi, :int(p.counts[qi])].numpy() for p in parts])
                ps = np.concatenate([p.scores[qi, :int(p.counts[qi])].numpy() for p in parts])
                mi, ms = O.topk(ps, pi, k, metric)
                out.ids[qi, :len(mi)] = torch.from_numpy(mi)
                out.scores[qi, :len(ms)] = torch.from_numpy(ms)
                out.counts[qi] = len(mi)

        s = ShardedSearcher(nq, k, torch.device("cpu"), local_search, merge)
        assert s.world == world and s.rank == rank
        out = s.search(torch.from_numpy(q))
        want = O.search(stored, ids, q, k, metric)
        ok = True
        for qi, (wi, ws) in enumerate(want):
            ok &= int(out.counts[qi]) == len(wi)
            ok &= np.array_equal(out.ids[qi, :len(wi)].numpy(), wi)
            ok &= np.array_equal(out.scores[qi, :len(wi)].numpy().view(np.uint32), ws.view(np.uint32))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,world", [("ip", 2), ("l2", 2), ("ip", 3)])
def test_sharded_search_gloo(metric, world):
    # (three ranks: shards of unequal length, 334 + 334 + 335 rows)
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, metric, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
        for p in procs:
            assert p.exitcode == 0, "rank exited with %s" % p.exitcode
        assert dict(ret) == {r: True for r in range(world)}


def test_shard_ranges_cover_rows_exactly():
    from vectordb_amd.sharded import record_bytes, shard_range
    for n in (0, 1, 7, 8, 1000, 10_000_000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert record_bytes(1, 10) == 128 and record_bytes(3, 10) % 16 == 0
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
