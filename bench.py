#!/usr/bin/env python3
"""Headline benchmark: QPS of single-query cosine top-10 over 10M x 768 fp32 rows held in HBM.

  python bench.py --gpus 1 --steps 100 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one query scored against every stored row, exact top-10 returned (ids + scores) in
device memory.  With N > 1 the 10M rows are row-sharded over the N ranks (strong scaling): each
rank scans its shard, one RCCL all-gather exchanges the per-rank top-10 records, every rank
merges them.  Inputs (table and queries) are resident in HBM before the timed region.
Single-query steps are issued round-robin on --streams HIP streams (default 3), as the requests
of concurrent clients are: the library runs the table scans strictly one after the other, and the
short kernels and the collective around one scan overlap with the next query's scan.

Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel (the scan:
algorithmic bytes = rows_per_gpu * dim * 4 per launch over its HIP-event duration inside the
timed region, against 8 TB/s HBM), and a CPU baseline (the numpy/BLAS port of the same search,
timed on this host on a bounded sample of the same workload and scaled by rows).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vectordb_amd import _lib                      # noqa: E402
from vectordb_amd.engine import Engine             # noqa: E402
from vectordb_amd.sharded import ShardedSearcher, gpu_backend, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3          # dense fp32-input MFMA peak of one MI355X
DTYPES = {"f32": (_lib.F32, 4), "bf16": (_lib.BF16, 2), "f16": (_lib.F16, 2)}
METRICS = {"cosine": _lib.METRIC_COSINE, "ip": _lib.METRIC_IP, "l2": _lib.METRIC_L2}


def cpu_baseline(rows_total: int, dim: int, k: int, metric_name: str, budget_s: float):
    """Times the oracle's BLAS path (X @ q, argpartition, canonical re-score) on a sample of the
    workload's shape and scales QPS by sample_rows / rows_total."""
    from oracle import oracle as O                  # checker/baseline only, never the product path
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    metric = {"cosine": O.COSINE, "ip": O.IP, "l2": O.L2}[metric_name]
    sample_rows = min(rows_total, 400_000)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((sample_rows, dim), dtype=np.float32)
    if metric == O.COSINE:
        x /= np.linalg.norm(x, axis=1, keepdims=True)       # timing data only
    ids = np.arange(sample_rows, dtype=np.int64)
    n2 = (x * x).sum(axis=1) if metric == O.L2 else None
    q = rng.standard_normal((256, dim), dtype=np.float32)
    O.fast_search(x, ids, q[:1], k, metric, n2=n2)           # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        O.fast_search(x, ids, q[done % 256:done % 256 + 1], k, metric, n2=n2)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 2000:
            break
    qps_sample = done / el
    return {
        "value": qps_sample * sample_rows / rows_total,
        "unit": "QPS",
        "cores": int(threads),
        "kind": "port",
        "sample": "%d queries in %.1f s on %d x %d fp32 rows (numpy %s sgemv + argpartition + fp64 re-score of "
                  "k+54 candidates), QPS scaled by %d/%d rows" % (done, el, sample_rows, dim, np.__version__,
                                                                  sample_rows, rows_total),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nq", type=int, default=1, help="queries per step")
    ap.add_argument("--metric", default="cosine", choices=sorted(METRICS))
    ap.add_argument("--dtype", default="f32", choices=sorted(DTYPES))
    ap.add_argument("--streams", type=int, default=3,
                    help="queries in flight: step i runs on HIP stream i %% streams, as the requests of "
                         "concurrent clients do; the library runs the table scans one after the other and "
                         "overlaps the short kernels and the collective around them")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="library option set through vdb_set_option before the run (A/B runs), e.g. scan_serialize=0")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU baseline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo rehearses the path on fewer GPUs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path exists)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # ranks share a GPU only in the --backend gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            if world > ndev:
                raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank" % (world, ndev))
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    eng = Engine(device=dev_index)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        eng.set_option(key, float(val))
    dt, esz = DTYPES[args.dtype]
    metric = METRICS[args.metric]
    first, last = shard_range(args.rows, world, rank)
    n_local = last - first
    table = "bench"
    eng.create_table(table, args.dim, metric, dt)
    eng.reserve(table, n_local)
    t_fill = time.perf_counter()
    eng.insert_synthetic(table, n_local, seed=0, first_row=first)
    t_fill = time.perf_counter() - t_fill

    total = args.steps + args.warmup
    queries = torch.empty((total * args.nq, args.dim), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    eng.synth_fill_dev(queries.data_ptr(), total * args.nq, args.dim, seed=1, first_row=0, stream=stream.cuda_stream)
    local_search, merge = gpu_backend(eng, table, args.nq, args.k, metric)
    # batches keep the GPU busy on their own; single queries are pipelined over several streams
    n_streams = max(1, args.streams) if args.nq == 1 else 1
    searchers = [ShardedSearcher(args.nq, args.k, dev, local_search, merge) for _ in range(n_streams)]
    streams = [stream] if n_streams == 1 else [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    torch.cuda.synchronize()

    def step(i: int):
        slot = i % n_streams
        with torch.cuda.stream(streams[slot]):
            return searchers[slot].search(queries[i * args.nq:(i + 1) * args.nq])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    eng.prof_read()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        out = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    eng.prof_enable(False)
    scan_ms, scan_launches = eng.prof_read()
    last_ids = out.ids.cpu().numpy().copy()
    last_scores = out.scores.cpu().numpy().copy()

    # request latency through the host entry point (vdb_search: H2D of the query, the same kernels,
    # D2H of ids and scores), one blocking call at a time -- outside the timed region
    host_lat = None
    if world == 1 and args.nq == 1:
        q_host = queries[:64].cpu().numpy()
        lat = []
        for i in range(64):
            t1 = time.perf_counter()
            eng.search(table, q_host[i:i + 1], args.k)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = np.sort(np.array(lat[4:]))
        host_lat = {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                    "calls": int(lat.size), "what": "blocking vdb_search from host memory, one query per call"}

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    queries_done, fallbacks, batched = eng.stats()

    if rank == 0:
        qps = args.steps * args.nq / elapsed
        bytes_per_launch = n_local * args.dim * esz + (n_local * 4 if args.metric == "l2" else 0)
        avg_scan_s = (scan_ms / max(1, scan_launches)) / 1e3
        achieved = bytes_per_launch / avg_scan_s / 1e9 if avg_scan_s > 0 else 0.0
        line = {
            "metric": "QPS top-%d %s, %dx%d %s, single query" % (args.k, args.metric, args.rows, args.dim, args.dtype)
                      if args.nq == 1 else
                      "QPS top-%d %s, %dx%d %s, batch %d" % (args.k, args.metric, args.rows, args.dim, args.dtype, args.nq),
            "value": qps,
            "unit": "QPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else args.dtype + " rows, f32 accumulate",
            "data": "synthetic (counter-hash ~N(0,1) embeddings generated on device, seed 0; queries seed 1)",
            "config": {
                "workload": "%dx%d %s %s top-%d, nq=%d per step, rows sharded over %d GPU(s), exact ids+scores"
                            % (args.rows, args.dim, args.dtype, args.metric, args.k, args.nq, world),
                "rows_per_gpu": n_local,
                "parallelism": "row-shard x%d + RCCL all-gather of top-k records" % world if world > 1 else "single GPU",
                "streams": n_streams,
                "options": args.opt,
                "table_fill_s": round(t_fill, 2),
                "exact_fallbacks": fallbacks,
                "queries_counted": queries_done,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_scan_s * 1e3,
                "launches_timed": scan_launches,
            },
            "host_latency_ms": host_lat,
            "sample_result": {"ids": [int(x) for x in last_ids[0][:3]], "scores": [float(x) for x in last_scores[0][:3]]},
        }
        # HBM bytes per scan launch from the PMC pass (rocprofv3 --pmc FETCH_SIZE, a run of its own,
        # gfx950 half-count correction applied); only known for configurations that were profiled
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                pmc = json.load(fh).get("scan_kernel|rows_per_gpu=%d|dim=%d|dtype=%s" % (n_local, args.dim, args.dtype))
            if pmc and batched == 0:
                line["roofline"]["traffic"] = pmc["hbm_read_bytes_per_launch"]
                line["roofline"]["traffic_source"] = pmc["source"]
        except (OSError, ValueError):
            pass
        if batched > 0:
            # batched path: the dominant kernel is the MFMA GEMM of the last threshold pass over all rows
            # (one launch per 1024-query chunk); algorithmic flops = 2 * rows * dim * queries per launch
            nq_launch = min(args.nq, 1024)
            flops = 2.0 * n_local * args.dim * nq_launch
            tf = flops / avg_scan_s / 1e12 if avg_scan_s > 0 else 0.0
            opts = dict(o.split("=", 1) for o in args.opt)
            if args.dtype == "f32" and opts.get("gemm32_split") == "0":
                line["roofline"] = {
                    "bound": "mfma", "kernel": "gemm_thresh_kernel", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF, "traffic": None,
                    "flops_per_launch": flops, "avg_launch_ms": avg_scan_s * 1e3, "launches_timed": scan_launches,
                }
            else:
                # every table type runs its batched passes on the 16-bit matrix unit: fp32 rows rounded
                # to bf16 on the way to LDS (one product per term; gemm32_split=1: bf16 hi + lo, three
                # products), 16-bit rows as stored (query rounded once, or hi + lo: two products).  The
                # ceiling for algorithmic flops is the dense 16-bit peak / products per term.
                if args.dtype == "f32":
                    products = 3 if opts.get("gemm32_split") == "1" else 1
                    line["dtype"] = ("f32 rows; fast pass bf16 hi+lo x3 on MFMA" if products == 3 else
                                     "f32 rows; fast pass rounded to bf16 on MFMA") + ", f32 accumulate; fp64 re-score"
                else:
                    products = 2 if opts.get("gemm_split_" + args.dtype) == "1" else 1
                    line["dtype"] = args.dtype + " rows on MFMA, f32 accumulate; fp64 re-score"
                peak = 2500.0 / products
                line["roofline"] = {
                    "bound": "mfma", "kernel": "gemm32s_kernel", "achieved": tf, "peak": peak,
                    "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                    "flops_per_launch": flops, "mfma_flops_per_launch": products * flops,
                    "avg_launch_ms": avg_scan_s * 1e3, "launches_timed": scan_launches,
                    "row_bytes_per_launch": bytes_per_launch, "row_read_gbs_if_read_once": achieved,
                }
                if args.nq < 128:
                    # less than one 128-query tile: the pass is bound by reading the rows once
                    line["roofline"].update({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": achieved / HBM_PEAK_GBS, "bytes_per_launch": bytes_per_launch,
                                             "algorithmic_tflops": tf})
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                    pmc = json.load(fh).get("%s|rows_per_gpu=%d|dim=%d|dtype=%s|nq=%d" % (
                        line["roofline"]["kernel"], n_local, args.dim, args.dtype, nq_launch))
                if pmc and not args.opt:
                    # HBM bytes per launch of the full pass (own --pmc FETCH_SIZE run): the query tiles of
                    # one row range share the rows through L2, so this stays near one read of the table
                    line["roofline"]["traffic"] = pmc["hbm_read_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = pmc["source"]
            except (OSError, ValueError):
                pass
        if world == 1 and args.cpu_seconds > 0 and args.nq == 1:
            line["cpu_baseline"] = cpu_baseline(args.rows, args.dim, args.k, args.metric, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
