"""GPU test of the whole row-sharded product path with two processes (both on cuda:0, gloo as
the collective because two RCCL ranks cannot share a GPU): per-rank tables over shard_range,
vdb_search_dev into the record, one all-gather, vdb_merge_records_dev; result against the oracle
on the unsharded data."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import cases
    from oracle import oracle as O
    from vectordb_amd import _lib
    from vectordb_amd.engine import Engine
    from vectordb_amd.sharded import ShardedSearcher, gpu_backend, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        eng = Engine(0)
        n, d, k, nq = 20_000, 96, 10, 3
        rows = cases.make_rows("synth", 7, n, d)
        ids = cases.make_ids("hash", 7, n)
        q = cases.make_queries("synth", 7, nq, d, rows)
        lo, hi = shard_range(n, world, rank)
        name = "shard_r%d" % rank
        eng.create_table(name, d, _lib.METRIC_COSINE, _lib.F32)
        eng.insert(name, rows[lo:hi], ids[lo:hi])
        ls, mg = gpu_backend(eng, name, nq, k, _lib.METRIC_COSINE)
        s = ShardedSearcher(nq, k, torch.device("cuda", 0), ls, mg)
        out = s.search(torch.from_numpy(q).cuda())
        torch.cuda.synchronize()
        want = O.search(O.prepare_store(rows, O.COSINE), ids, q, k, O.COSINE)
        ok = True
        for qi, (wi, ws) in enumerate(want):
            ok &= int(out.counts[qi]) == k
            ok &= np.array_equal(out.ids[qi].cpu().numpy(), wi)
            ok &= np.array_equal(out.scores[qi].cpu().numpy().view(np.uint32), ws.view(np.uint32))
        ret[rank] = bool(ok)
        eng.drop_table(name)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_search_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
        for p in procs:
            assert p.exitcode == 0, "rank exited with %s" % p.exitcode
        assert dict(ret) == {0: True, 1: True}


def _run_bench(extra):
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "400000", "--steps", "8", "--warmup", "2",
           "--cpu-seconds", "0", "--no-secondary"] + extra
    env = dict(os.environ)
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_with_two_ranks_answers_like_one():
    """`bench.py --gpus 2` as the multi-GPU driver runs it, rehearsed here with two ranks that share the
    one GPU (gloo collective): bench.py starts its own rank processes, shards the rows, exchanges the
    records, merges, and prints one line whose answers are those of the one-rank run (the shard set is the
    reference's search_resources list, /root/reference/server_config.yaml:43-44)."""
    one = _run_bench([])
    two = _run_bench(["--gpus", "2", "--backend", "gloo", "--trace-steps", "4"])
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and two["config"]["rows_per_gpu"] == 200000
    assert two["config"]["backend"] == "gloo" and two["scaling"] == "strong"
    assert two["roofline"]["bytes_per_launch"] == 200000 * 768 * 4
    assert one["n_gpus"] == 1 and one["roofline"]["bytes_per_launch"] == 400000 * 768 * 4
    assert two["sample_result"] == one["sample_result"]
    assert two["config"]["exact_fallbacks"] == 0 and two["steps"] == 8
    tr = two["trace"]
    for part in ("host_enqueue_us", "device_complete_us"):
        assert set(tr[part]) >= {"local_search", "collective", "merge"} and tr[part]["steps"] >= 4
    assert tr["device_complete_us"]["local_search"] > 0 and tr["device_complete_us"]["collective"] > 0


def test_rccl_path_with_a_world_of_one_rank():
    """The calls an N-rank run makes, through the real RCCL backend with the one rank a one-GPU box allows:
    process group with a device id, ShardedSearcher.search on three streams (local search, the all-gather
    issued from the process's collective stream, merge), teardown; the answers are those of a plain
    search (tools/rccl_single_rank_check.py, run as a child process with a port of its own)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(var, None)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank_check.py")], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, (res.stdout[-1000:], res.stderr[-2000:])
    assert "RCCL single-rank path: ok" in res.stdout
