"""Tables whose row norms are heavy-tailed: how many queries leave the fast path, and what the exact
pass costs.

A fraction of the rows (every --every-th) is scaled by --scale before the insert.  For each metric
the script reports single-query and batch-1024 throughput with the number of queries that took the
exact second pass, and then the time of a query when every query is sent to the exact pass
(option force_exact).  Run on the GPU box:  python tools/bench_heavy_tail.py --rows 10000000

--orthogonal N adds N rows that are --scale times longer than the rest and orthogonal to the queries of
the batch (512 queries then, so that a 768-dimensional space has room for such rows): their true scores
are zero, their 16-bit fast keys are noise of the size of the best scores.  --ip-outliers 1 0 runs the
IP table with and without the side list for such rows.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib  # noqa: E402
from vectordb_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--scale", type=float, nargs="+", default=[1.0, 100.0])
    ap.add_argument("--every", type=int, default=1000)
    ap.add_argument("--metrics", nargs="+", default=["ip", "l2"])
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--orthogonal", type=int, default=0)
    ap.add_argument("--ip-outliers", type=int, nargs="+", default=[1])
    ap.add_argument("--row-bounds", type=int, nargs="+", default=[1],
                    help="L2 tables: 1 = keys carry their own row's error term, 0 = table-wide norm bound")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = Engine(0)
    if len(args.ip_outliers) == 1:
        eng.set_option("ip_outliers", args.ip_outliers[0])
    stream = torch.cuda.current_stream().cuda_stream
    metrics = {"ip": _lib.METRIC_IP, "l2": _lib.METRIC_L2, "cosine": _lib.METRIC_COSINE}
    nq_b = 512 if args.orthogonal else 1024
    queries = torch.empty((args.queries + nq_b, args.dim), dtype=torch.float32, device=dev)
    eng.synth_fill_dev(queries.data_ptr(), queries.shape[0], args.dim, seed=1, first_row=0, stream=stream)
    ids1 = torch.empty((1, args.k), dtype=torch.int64, device=dev)
    sc1 = torch.empty((1, args.k), dtype=torch.float32, device=dev)
    cn1 = torch.empty((1,), dtype=torch.int32, device=dev)
    idsb = torch.empty((nq_b, args.k), dtype=torch.int64, device=dev)
    scb = torch.empty((nq_b, args.k), dtype=torch.float32, device=dev)
    cnb = torch.empty((nq_b,), dtype=torch.int32, device=dev)
    chunk = 500_000
    buf = torch.empty((chunk, args.dim), dtype=torch.float32, device=dev)
    for mname in args.metrics:
        for scale in args.scale:
            name = "ht"
            eng.create_table(name, args.dim, metrics[mname], _lib.F32)
            eng.reserve(name, args.rows)
            for c0 in range(0, args.rows, chunk):
                n = min(chunk, args.rows - c0)
                eng.synth_fill_dev(buf.data_ptr(), n, args.dim, seed=0, first_row=c0, stream=stream)
                rid = torch.arange(c0, c0 + n, dtype=torch.int64, device=dev)
                if scale != 1.0:
                    buf[:n][rid % args.every == 0] *= scale
                torch.cuda.synchronize()
                eng.insert_dev(name, n, buf.data_ptr(), rid.data_ptr(), stream)
            torch.cuda.synchronize()
            if args.orthogonal and scale != 1.0:
                qb = queries[args.queries:].double()
                basis, _ = torch.linalg.qr(qb.T)                      # [dim, nq_b]
                v = torch.randn((args.orthogonal, args.dim), dtype=torch.float64, device=dev)
                v -= (v @ basis) @ basis.T
                v = (v / v.norm(dim=1, keepdim=True) * (args.dim ** 0.5) * scale).float().contiguous()
                rid = torch.arange(args.rows, args.rows + args.orthogonal, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                eng.insert_dev(name, args.orthogonal, v.data_ptr(), rid.data_ptr(), stream)
                torch.cuda.synchronize()

            def single(nrun, first=0):
                q0, f0, _ = eng.stats()
                t0 = time.perf_counter()
                for i in range(nrun):
                    eng.search_dev(name, 1, queries[first + i].data_ptr(), args.k, ids1.data_ptr(), sc1.data_ptr(),
                                   cn1.data_ptr(), stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                q1, f1, _ = eng.stats()
                return nrun / dt, f1 - f0

            def batch(nrun):
                q0, f0, _ = eng.stats()
                t0 = time.perf_counter()
                for i in range(nrun):
                    eng.search_dev(name, nq_b, queries[args.queries:].data_ptr(), args.k, idsb.data_ptr(),
                                   scb.data_ptr(), cnb.data_ptr(), stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                q1, f1, _ = eng.stats()
                return nrun * nq_b / dt, f1 - f0

            settings = [(r, 1) for r in args.row_bounds] if mname == "l2" else [(1, o) for o in args.ip_outliers]
            for rbnd, outl in settings:
                eng.set_option("l2_row_bounds", rbnd)
                if mname == "ip" and outl != 1 and len(args.ip_outliers) > 1:
                    # the list is built while rows arrive: a table filled without it
                    print("(ip_outliers=%d needs a refill; run the script with --ip-outliers %d alone)" % (outl, outl), flush=True)
                single(5)
                qps, fb = single(args.queries)
                batch(1)
                bqps, bfb = batch(3)
                eng.set_option("force_exact", 1)
                single(1)
                eqps, efb = single(5)
                eng.set_option("force_exact", 0)
                print("%s %dx%d, every %d-th row x%g, %d orthogonal long rows, row bounds %d, side list %d: single %.1f QPS (one stream), %d of %d queries to the exact pass; "
                      "batch 1024 %.0f QPS, %d of %d to the exact pass; forced exact pass %.1f ms per query"
                      % (mname, args.rows, args.dim, args.every, scale, args.orthogonal if scale != 1.0 else 0, rbnd, outl, qps, fb, args.queries, bqps, bfb, 3 * nq_b,
                         1e3 / eqps), flush=True)
            eng.set_option("l2_row_bounds", 1)
            eng.drop_table(name)


if __name__ == "__main__":
    main()
