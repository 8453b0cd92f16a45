#!/usr/bin/env python3
"""Concurrent single-query callers through the client shim (one thread per caller, the serving
pattern of /root/reference/app.py), with and without merging of concurrent searches (GPU only).
Prints queries per second and the engine calls made, per thread count."""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib  # noqa: E402
from vectordb_amd.milvus import Milvus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--threads", default="1,4,16,64")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--inserters", type=int, default=0,
                    help="threads that insert one row per call (as PUT /insert does) while the lookups run")
    args = ap.parse_args()
    m = Milvus()
    assert m.connect().OK()
    eng = m._eng()
    name = "conc"
    eng.create_table(name, args.dim, _lib.METRIC_COSINE, _lib.F32)
    eng.reserve(name, args.rows)
    eng.insert_synthetic(name, args.rows, 0, 0)
    qd = torch.empty((256, args.dim), dtype=torch.float32, device="cuda")
    eng.synth_fill_dev(qd.data_ptr(), 256, args.dim, 1, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    queries = [row.tolist() for row in qd.cpu().numpy()]
    for merged in (False, True):
        eng.set_option("coalesce_host", 1 if merged else 0)
        for nt in [int(x) for x in args.threads.split(",")]:
            stop = time.perf_counter() + args.seconds
            counts = [0] * nt
            inserted = [0] * max(1, args.inserters)
            rng = np.random.default_rng(5)
            new_rows = rng.standard_normal((4096, args.dim)).astype(np.float32).tolist()

            def inserter(t):
                i = 0
                base = 1 << 40
                while time.perf_counter() < stop:
                    st, _ = m.insert(name, records=[new_rows[i % 4096]], ids=[base + t * (1 << 30) + inserted[t]])
                    assert st.OK(), st.message
                    inserted[t] += 1
                    i += 1

            def worker(t):
                i = t
                while time.perf_counter() < stop:
                    st, res = m.search_vectors(table_name=name, query_records=[queries[i % 256]], top_k=args.k, nprobe=10)
                    assert st.OK() and len(res[0]) == args.k
                    counts[t] += 1
                    i += nt

            t0 = time.perf_counter()
            threads = [threading.Thread(target=worker, args=(t,)) for t in range(nt)]
            threads += [threading.Thread(target=inserter, args=(t,)) for t in range(args.inserters)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            dt = time.perf_counter() - t0
            print("merged=%d threads=%3d inserters=%d: %8.1f QPS  (%d queries, %.2f s)%s" % (
                merged, nt, args.inserters, sum(counts) / dt, sum(counts), dt,
                "  %.1f inserts/s" % (sum(inserted) / dt) if args.inserters else ""), flush=True)
    eng.set_option("coalesce_host", 1)
    eng.drop_table(name)


if __name__ == "__main__":
    main()
