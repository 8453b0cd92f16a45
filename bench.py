#!/usr/bin/env python3
"""Headline benchmark: QPS of single-query cosine top-10 over 10M x 768 fp32 rows held in HBM.

  python bench.py --gpus 1 --steps 100 --warmup 10
  python bench.py --gpus N ...                      (starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one query (or one batch of --nq queries) scored against every stored row, exact top-k
returned (ids + scores) in device memory.  With N > 1 the rows are row-sharded over the N ranks
(strong scaling): each rank scans its shard, one RCCL all-gather exchanges the per-rank top-k
records, every rank merges them.  Inputs (table and queries) are resident in HBM before the timed
region.  Single-query steps are issued round-robin on --streams HIP streams (default: 3; 2 over a
single-GPU table under 8 GB), as the requests of concurrent clients are: the library runs the table scans strictly one after the other,
and the short kernels and the collective around one scan overlap with the next query's scan.

Prints ONE JSON line (rank 0): the throughput, the roofline of the dominant kernel (single queries:
the scan, algorithmic bytes = rows_per_gpu * dim * element size per launch over its HIP-event
duration inside the timed region, against 8 TB/s of HBM; batches: the full pass of the MFMA kernel,
2 * rows * dim * queries flops per launch against the dense 16-bit MFMA peak), a CPU baseline (the
numpy/BLAS port of the same search, timed on this host on a bounded sample of the same workload and
scaled by rows), and, on one GPU with the default workload, a `secondary` list: the other BASELINE
configurations that fit one GPU, measured in the same run, each with its own roofline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3          # dense fp32-input MFMA peak of one MI355X
MFMA_16_PEAK_TF = 2500.0          # dense bf16 / fp16 MFMA peak


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
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": qps_sample * sample_rows / rows_total,
        "unit": "QPS",
        "cores": int(threads),
        "os_cpu_count": os.cpu_count(),
        "cpu_model": cpu_model,
        "openblas_num_threads": os.environ.get("OPENBLAS_NUM_THREADS"),
        "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
        "kind": "port",
        "sample": "%d queries in %.1f s on %d x %d fp32 rows (numpy %s sgemv + argpartition + fp64 re-score of "
                  "k+54 candidates), QPS scaled by %d/%d rows" % (done, el, sample_rows, dim, np.__version__,
                                                                  sample_rows, rows_total),
    }


def spawn_ranks(n: int, argv) -> int:
    """`bench.py --gpus N` from a bare shell: this process touches no GPU; it starts N fresh rank
    processes (one per GPU, rendezvous on 127.0.0.1), relays rank 0's JSON line and returns non-zero
    if any rank failed."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    # rank 0's JSON line only (the gloo backend prints a connection notice on stdout)
    for ln in (out or "").splitlines():
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nq", type=int, default=1, help="queries per step")
    ap.add_argument("--metric", default="cosine", choices=["cosine", "ip", "l2"])
    ap.add_argument("--dtype", default="f32", choices=["bf16", "f16", "f32"])
    ap.add_argument("--streams", type=int, default=0,
                    help="queries in flight: step i runs on HIP stream i %% streams, as the requests of "
                         "concurrent clients do; the library runs the table scans one after the other and "
                         "overlaps the short kernels and the collective around them")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="library option set through vdb_set_option before the run (A/B runs), e.g. scan_serialize=0")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU baseline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo rehearses the path on fewer GPUs)")
    ap.add_argument("--trace-steps", type=int, default=0,
                    help="after the timed region, this many extra steps are broken down per phase (local search, "
                         "collective, merge): host microseconds to enqueue each, and microseconds until each has "
                         "completed on the device")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other BASELINE configurations measured after the headline on one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))      # before anything touches a GPU
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank_main(args, world)


def rank_main(args, world: int) -> None:
    import torch
    import torch.distributed as dist

    from vectordb_amd import _lib
    from vectordb_amd.engine import Engine
    from vectordb_amd.sharded import ShardedSearcher, gpu_backend, shard_range

    DTYPES = {"f32": (_lib.F32, 4), "bf16": (_lib.BF16, 2), "f16": (_lib.F16, 2)}
    METRICS = {"cosine": _lib.METRIC_COSINE, "ip": _lib.METRIC_IP, "l2": _lib.METRIC_L2}

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path exists)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # ranks share a GPU only in the --backend gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ranks_seen = 1
    if world > 1:
        if args.backend == "nccl":
            if world > ndev:
                raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank" % (world, ndev))
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        # how many ranks the collective backend really joins: a sum of ones over the group
        ones = torch.ones(1, dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    eng = Engine(device=dev_index)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        eng.set_option(key, float(val))
    opts = dict(o.split("=", 1) for o in args.opt)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stream_pool = []

    def run_workload(table, rows_total, n_local, dim, dtype, metric_name, k, nq, steps, warmup, n_streams, latency, mirror=False):
        """Times `steps` steps on an existing table; returns the measured entry (rank 0's view)."""
        esz = DTYPES[dtype][1]
        metric = METRICS[metric_name]
        total = steps + warmup
        queries = torch.empty((total * nq, dim), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream()
        eng.synth_fill_dev(queries.data_ptr(), total * nq, dim, seed=1, first_row=0, stream=stream.cuda_stream)
        local_search, merge = gpu_backend(eng, table, nq, k, metric)
        # batches keep the GPU busy on their own; single queries are pipelined over several streams
        if n_streams <= 0:
            # queries in flight: two; over a scan of a few hundred microseconds a third queue's small kernels cost more
            # between scans than they hide (measured: 12 us per query)
            # (two also over a long scan on one GPU: the same QPS as three, and an answer after two scans instead of
            # three; the collective of a sharded run is what the third one hides)
            n_streams = 3 if world > 1 else 2
        n_streams = max(1, n_streams) if nq == 1 else 1
        searchers = [ShardedSearcher(nq, k, dev, local_search, merge) for _ in range(n_streams)]
        # the same stream objects for every workload of the process: the device has four hardware queues,
        # and streams beyond them share one (their work is then serialised and timed with its queueing)
        while len(stream_pool) < n_streams:
            stream_pool.append(torch.cuda.Stream(device=dev))
        streams = [stream] if n_streams == 1 else stream_pool[:n_streams]
        torch.cuda.synchronize()

        def step(i: int):
            slot = i % n_streams
            with torch.cuda.stream(streams[slot]):
                return searchers[slot].search(queries[i * nq:(i + 1) * nq])

        q0, f0, b0 = eng.stats()
        for i in range(warmup):
            step(i)
        fence()
        eng.prof_read()
        # every launch of the dominant kernel is timed with HIP events on its stream; in runs of 64 or more
        # scans over less than 8 GB every eighth one (a pair of timing events costs the queue about 10 us
        # between two kernels: 2 % of a scan of half a millisecond)
        timed_every = 8 if (steps >= 64 and nq == 1 and n_local * dim * esz < 8e9) else 1
        eng.prof_enable(timed_every)
        t0 = time.perf_counter()
        out = None
        for i in range(warmup, total):
            out = step(i)
        fence()
        elapsed = time.perf_counter() - t0
        eng.prof_enable(False)
        kern_ms, launches = eng.prof_read()
        last_ids = out.ids.cpu().numpy().copy()
        last_scores = out.scores.cpu().numpy().copy()

        # When is a query's answer complete while queries are pipelined over several streams?  Outside the timed
        # region, with the same issue pattern: a pair of events around every step on its stream (the first fires
        # when the stream is free for the step, i.e. when the step two or three before it has completed).
        pipe_lat = None
        if nq == 1 and n_streams > 1 and latency:
            nlat = min(steps, 60)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlat)]
            for j in range(nlat):
                i = warmup + (j % steps)
                with torch.cuda.stream(streams[i % n_streams]):
                    evs[j][0].record()
                step(i)
                with torch.cuda.stream(streams[i % n_streams]):
                    evs[j][1].record()
            fence()
            ms = np.sort(np.array([a.elapsed_time(b) for a, b in evs[n_streams:]]))
            pipe_lat = {"p50": float(np.percentile(ms, 50)), "p99": float(np.percentile(ms, 99)), "steps": int(ms.size),
                        "what": "first kernel of a query started on its stream -> its answer complete, %d streams in flight" % n_streams}

        # per-phase breakdown of a step, outside the timed region: what the host spends enqueueing each
        # phase while steps are pipelined, and how long each phase takes on the device when nothing overlaps
        trace = None
        if args.trace_steps > 0:
            trace = {}
            for label, sync in (("host_enqueue_us", False), ("device_complete_us", True)):
                for s in searchers:
                    s.trace, s.trace_sync = [], sync
                for j in range(args.trace_steps):
                    step(warmup + (j % max(1, steps)))
                    if sync:
                        torch.cuda.synchronize()
                fence()
                rows_ = np.array([t for s in searchers for t in s.trace], dtype=np.float64).reshape(-1, 3)
                trace[label] = {"local_search": float(np.median(rows_[:, 0])), "collective": float(np.median(rows_[:, 1])),
                                "merge": float(np.median(rows_[:, 2])), "steps": int(rows_.shape[0]), "stat": "median"}
                for s in searchers:
                    s.trace, s.trace_sync = None, False

        # request latency through the host entry point (vdb_search: H2D of the query, the same kernels,
        # D2H of ids and scores), one blocking call at a time -- outside the timed region, on queries of
        # its own
        host_lat = None
        if latency and world == 1 and nq == 1:
            ncalls = 64
            q_lat = torch.empty((ncalls, dim), dtype=torch.float32, device=dev)
            eng.synth_fill_dev(q_lat.data_ptr(), ncalls, dim, seed=2, first_row=0, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            q_host = q_lat.cpu().numpy()
            lat = []
            for i in range(ncalls):
                one = q_host[i:i + 1]
                assert one.shape[0] == 1
                t1 = time.perf_counter()
                eng.search(table, one, k)
                lat.append((time.perf_counter() - t1) * 1e3)
            lat = np.sort(np.array(lat[4:]))
            host_lat = {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                        "calls": int(lat.size), "what": "blocking vdb_search from host memory, one query per call"}

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if world > 1 and args.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elif world > 1:
            tc = t.cpu()
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            t = tc
        elapsed = float(t.item())
        q1, f1, b1 = eng.stats()
        batched = b1 - b0

        qps = steps * nq / elapsed
        bytes_per_launch = n_local * dim * esz + (n_local * 4 if metric_name == "l2" else 0)
        mirror_bytes = n_local * ((dim + 31) // 32 * 32) * 2 + (n_local * 4 if metric_name == "l2" else 0)
        mirror = mirror or (opts.get("scan_mirror") == "1" and dtype == "f32" and dim % 32 == 0 and k <= 10)
        if mirror and nq == 1:
            # option scan_mirror: the scan reads the bf16 mirror of the fp32 rows (rows padded to 32 elements)
            bytes_per_launch = mirror_bytes
        avg_s = (kern_ms / max(1, launches)) / 1e3
        gbs = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        entry = {
            "metric": "QPS top-%d %s, %dx%d %s, %s" % (k, metric_name, rows_total, dim, dtype,
                                                      "single query" if nq == 1 else "batch %d" % nq),
            "value": qps,
            "unit": "QPS",
            "ms_per_step": elapsed / steps * 1e3,
            "steps": steps,
            "warmup": warmup,
            "dtype": "f32" if dtype == "f32" else dtype + " rows, f32 accumulate",
            "workload": "%dx%d %s %s top-%d, nq=%d per step, rows sharded over %d GPU(s), exact ids+scores"
                        % (rows_total, dim, dtype, metric_name, k, nq, world),
            "rows_per_gpu": n_local,
            "streams": n_streams,
            "exact_fallbacks": f1 - f0,
            "queries_counted": q1 - q0,
            "roofline": {
                "bound": "hbm", "kernel": "scan_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_s * 1e3, "launches_timed": launches, "timed_every": timed_every,
            },
            "host_latency_ms": host_lat,
            "pipelined_completion_ms": pipe_lat,
            "sample_result": {"ids": [int(x) for x in last_ids[0][:3]], "scores": [float(x) for x in last_scores[0][:3]]},
        }
        if trace is not None:
            entry["trace"] = trace
        if mirror and nq == 1:
            entry["dtype"] = "f32 rows; scan over their bf16 mirror, f32 accumulate; fp64 re-score of the candidates from the f32 rows"
            entry["roofline"]["kernel"] = "scan_kernel (bf16 mirror)"
            entry["roofline"]["note"] = ("bytes_per_launch are the mirror's: the fp32 rows (%d bytes) are not read by the scan"
                                         % (n_local * dim * esz))
        if batched > 0:
            # batched path: the dominant kernel is the full pass of the MFMA kernel over all rows (one
            # launch per 1024-query chunk); algorithmic flops = 2 * rows * dim * queries per launch
            nq_launch = min(nq, 1024)
            flops = 2.0 * n_local * dim * nq_launch
            tf = flops / avg_s / 1e12 if avg_s > 0 else 0.0
            if dtype == "f32" and opts.get("gemm32_split") == "0":
                entry["roofline"] = {
                    "bound": "mfma", "kernel": "gemm_thresh_kernel", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF, "traffic": None,
                    "flops_per_launch": flops, "avg_launch_ms": avg_s * 1e3, "launches_timed": launches,
                }
            else:
                # every table type runs its batched passes on the 16-bit matrix unit: fp32 tables multiply
                # a bf16 image of their rows (one product per term; gemm32_split=1: bf16 hi + lo, three
                # products), 16-bit rows go in as stored (query rounded once, or hi + lo: two products).
                # The ceiling for algorithmic flops is the dense 16-bit peak / products per term.
                if dtype == "f32":
                    products = 3 if opts.get("gemm32_split") == "1" else 1
                    entry["dtype"] = ("f32 rows; fast pass bf16 hi+lo x3 on MFMA" if products == 3 else
                                      "f32 rows; fast pass on their bf16 image on MFMA") + ", f32 accumulate; fp64 re-score"
                else:
                    products = 2 if opts.get("gemm_split_" + dtype) == "1" else 1
                    entry["dtype"] = dtype + " rows on MFMA, f32 accumulate; fp64 re-score"
                pp = nq_launch > 128 and products == 1 and opts.get("gemm_pp") != "0" and opts.get("bf16_mirror") != "0"
                peak = MFMA_16_PEAK_TF / products
                entry["roofline"] = {
                    "bound": "mfma", "kernel": ("gemm_pp384_kernel" if (opts.get("gemm_pp_tile") == "384" or (nq_launch <= 256 and opts.get("gemm_pp_tile") != "256"))
                                                        else "gemm_pp_kernel") if pp else "gemm32s_kernel",
                    "achieved": tf, "peak": peak,
                    "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                    "flops_per_launch": flops, "mfma_flops_per_launch": products * flops,
                    "avg_launch_ms": avg_s * 1e3, "launches_timed": launches,
                    "row_bytes_per_launch": bytes_per_launch, "row_read_gbs_if_read_once": gbs,
                }
                # small batches sit near the ridge: the same launch against the HBM peak (rows read once per pass)
                entry["roofline"]["hbm_frac_if_rows_read_once"] = gbs / HBM_PEAK_GBS
                if dtype == "f32" and products == 1 and opts.get("bf16_mirror") != "0":
                    # these passes read the bf16 mirror of the fp32 rows
                    entry["roofline"]["row_bytes_per_launch"] = mirror_bytes
                    entry["roofline"]["row_read_gbs_if_read_once"] = mirror_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
                    entry["roofline"]["hbm_frac_if_rows_read_once"] = entry["roofline"]["row_read_gbs_if_read_once"] / HBM_PEAK_GBS
                if nq < 128:
                    # less than one 128-query tile: the pass is bound by reading the rows once
                    rb = entry["roofline"]["row_bytes_per_launch"]
                    g1 = entry["roofline"]["row_read_gbs_if_read_once"]
                    entry["roofline"].update({"bound": "hbm", "achieved": g1, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": g1 / HBM_PEAK_GBS, "bytes_per_launch": rb,
                                              "algorithmic_tflops": tf})
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE, runs of their own, gfx950
        # half-count correction applied); only known for configurations that were profiled with the
        # current kernels (the file names the commit)
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                tj = json.load(fh)
            key = "%s|rows_per_gpu=%d|dim=%d|dtype=%s|nq=%d" % (entry["roofline"]["kernel"], n_local, dim, dtype, min(nq, 1024))
            pmc = tj.get(key)
            if pmc and not args.opt and pmc.get("hbm_read_bytes_per_launch"):
                entry["roofline"]["traffic"] = pmc["hbm_read_bytes_per_launch"]
                entry["roofline"]["traffic_source"] = pmc["source"]
        except (OSError, ValueError):
            pass
        del queries
        return entry

    # ------------------------------------------------------------------ headline workload
    dt, _ = DTYPES[args.dtype]
    metric = METRICS[args.metric]
    first, last = shard_range(args.rows, world, rank)
    n_local = last - first
    table = "bench"
    eng.create_table(table, args.dim, metric, dt)
    eng.reserve(table, n_local)
    t_fill = time.perf_counter()
    eng.insert_synthetic(table, n_local, seed=0, first_row=first)
    t_fill = time.perf_counter() - t_fill
    main_e = run_workload(table, args.rows, n_local, args.dim, args.dtype, args.metric, args.k, args.nq,
                          args.steps, args.warmup, args.streams, latency=True)

    # ------------------------------------------------------------------ other BASELINE configurations
    secondary = []
    default_run = (args.rows == 10_000_000 and args.dim == 768 and args.dtype == "f32" and args.metric == "cosine"
                   and args.nq == 1 and args.k == 10 and not args.opt)
    if world == 1 and default_run and not args.no_secondary:
        def drop_and_settle(name, nbytes):
            # memory handed back is cleared by the driver in the background, at the expense of whatever
            # scans next (measured: 4 % for about half a second after 46 GB were dropped)
            eng.drop_table(name)
            time.sleep(min(3.0, 0.2 + nbytes / 2.5e10))

        def other(rows, dim, dtype, mname, k, nq, steps, warmup, note, n_streams=None):
            name = "bench_%s_%d" % (dtype, rows)
            eng.create_table(name, dim, METRICS[mname], DTYPES[dtype][0])
            eng.reserve(name, rows)
            eng.insert_synthetic(name, rows, seed=0, first_row=0)
            e = run_workload(name, rows, rows, dim, dtype, mname, k, nq, steps, warmup,
                             args.streams if n_streams is None else n_streams, latency=False)
            e["note"] = note
            secondary.append(e)
            drop_and_settle(name, rows * dim * DTYPES[dtype][1])

        # BASELINE's first configuration as the reference runs it: 10 000 x 384 U[0,1) vectors, one PUT /insert each,
        # then single-vector PUT /lookup calls, through the HTTP routes in this process (JSON, md5 ids, SQLite and the
        # engine call included); the engine part of a lookup is the blocking vdb_search measured by
        # tools/small_table_latency.py
        try:
            import contextlib
            from tools.bench_http import run as http_run
            with contextlib.redirect_stdout(sys.stderr):            # the routes print like the reference's do
                h = http_run(10_000, 384, 100)
            # the engine part of such a lookup: blocking vdb_search calls of one query from host memory
            eng.create_table("bench_small", 384, METRICS["cosine"], DTYPES["f32"][0])
            eng.insert_synthetic("bench_small", 10_000, seed=0, first_row=0)
            sq = np.random.default_rng(1).random((450, 384)).astype(np.float32)
            call_us = []
            for i in range(450):
                t0 = time.perf_counter()
                eng.search("bench_small", sq[i:i + 1], 10)
                if i >= 50:
                    call_us.append((time.perf_counter() - t0) * 1e6)
            eng.drop_table("bench_small")
            # the same engine call on the host cores: the oracle's BLAS path over 10 000 x 384 U[0,1) rows (checker
            # code, timed as the CPU baseline of this configuration; it is not part of any product path)
            cpu_small = None
            if args.cpu_seconds > 0:
                from oracle import oracle as O
                cx = O.prepare_store(np.random.default_rng(0).random((10_000, 384)), O.COSINE, O.F32)
                cids = np.arange(10_000, dtype=np.int64)
                O.fast_search(cx, cids, sq[:1], 10, O.COSINE)
                cpu_us = []
                for i in range(200):
                    t0 = time.perf_counter()
                    O.fast_search(cx, cids, sq[i:i + 1], 10, O.COSINE)
                    cpu_us.append((time.perf_counter() - t0) * 1e6)
                cpu_small = {"p50": float(np.percentile(cpu_us, 50)), "p99": float(np.percentile(cpu_us, 99)), "calls": len(cpu_us),
                             "kind": "port", "what": "oracle.fast_search (numpy sgemv + argpartition + fp64 re-score) of one query over 10000x384 on the host cores"}
            secondary.append({"metric": "PUT /lookup per second, 10000x384 f32 cosine top-10 through the HTTP routes",
                              "value": h["lookup_qps"], "unit": "QPS", "lookup_ms": h["lookup_ms"],
                              "engine_call_us": {"p50": float(np.percentile(call_us, 50)), "p99": float(np.percentile(call_us, 99)),
                                                 "calls": len(call_us), "what": "blocking vdb_search of one query over the same 10000x384 table, through the Python binding"},
                              "cpu_engine_call_us": cpu_small,
                              "insert_rows_per_s": h["insert_rows_per_s"], "workload": h["workload"], "dtype": "f32",
                              "note": "BASELINE configuration 1 (the reference's own workload shape) end to end; no roofline: a call is all fixed cost"})
        except Exception as exc:        # the HTTP layer needs flask; the kernels' numbers do not depend on it
            secondary.append({"metric": "HTTP routes, 10000x384", "value": None, "note": "not measured: %s: %s" % (type(exc).__name__, exc)})
        # the single-query shapes first, the matrix-unit ones after them: a batched pass leaves the chip at a
        # lower clock for a while, which costs a scan of half a millisecond several per cent
        other(1_000_000, 768, "f32", "cosine", 10, 1, 1000, 100, "1M x 768 fp32 single query")
        other(12_500_000, 128, "bf16", "ip", 100, 1, 1000, 100, "one shard of 8 of the 100M x 128 bf16 top-100 configuration")
        # what one GPU of eight does per query of the headline workload (--gpus 8 shards the 10M rows): without the
        # all-gather and the merge, on three streams as the sharded path runs them
        other(1_250_000, 768, "f32", "cosine", 10, 1, 1000, 100,
              "one shard of 8 of the headline table (local part of a --gpus 8 step: no collective, no merge)", n_streams=3)
        # the headline workload with option scan_mirror: the scan reads the bf16 mirror of the rows (half the
        # bytes), the certificate's bound widens by the rounding, results are the same exact ones
        eng.set_option("scan_mirror", 1)
        try:
            e = run_workload(table, args.rows, n_local, 768, "f32", "cosine", 10, 1, 200, 20, args.streams, latency=False, mirror=True)
            e["note"] = "option scan_mirror=1 (not the default)"
            secondary.append(e)
        finally:
            eng.set_option("scan_mirror", 0)
        # batch 1024 on the same table (MFMA GEMM + fused top-k)
        # (the matrix-unit passes of a process get faster over their first eight or so launches, 10.8 -> 10.05 ms for
        # the seven-eighths launch in the per-dispatch trace of this command, hence the longer warm-up)
        secondary.append(run_workload(table, args.rows, n_local, 768, "f32", "cosine", 10, 1024, 10, 6, 1, latency=False))
        drop_and_settle(table, n_local * 768 * 6)
        other(20_000_000, 768, "f16", "l2", 10, 256, 10, 6, "20M of the 125M rows one GPU holds in the 1B x 768 fp16 L2 batch-256 configuration")

    if rank == 0:
        line = {
            "metric": main_e["metric"],
            "value": main_e["value"],
            "unit": "QPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_e["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": main_e["dtype"],
            "data": "synthetic (counter-hash ~N(0,1) embeddings generated on device, seed 0; queries seed 1)",
            "config": {
                "workload": main_e["workload"],
                "rows_per_gpu": n_local,
                "parallelism": "row-shard x%d + all-gather of top-k records" % world if world > 1 else "single GPU",
                "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if world > 1 else "none",
                "rccl_ranks": ranks_seen,
                "streams": main_e["streams"],
                "options": args.opt,
                "table_fill_s": round(t_fill, 2),
                "exact_fallbacks": main_e["exact_fallbacks"],
                "queries_counted": main_e["queries_counted"],
            },
            "roofline": main_e["roofline"],
            "host_latency_ms": main_e["host_latency_ms"],
            "pipelined_completion_ms": main_e.get("pipelined_completion_ms"),
            "sample_result": main_e["sample_result"],
        }
        if "trace" in main_e:
            line["trace"] = main_e["trace"]
        if secondary:
            line["secondary"] = secondary
        if world == 1 and args.cpu_seconds > 0 and args.nq == 1:
            line["cpu_baseline"] = cpu_baseline(args.rows, args.dim, args.k, args.metric, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
