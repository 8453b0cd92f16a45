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

            def worker(t):
                i = t
                while time.perf_counter() < stop:
                    st, res = m.search_vectors(table_name=name, query_records=[queries[i % 256]], top_k=args.k, nprobe=10)
                    assert st.OK() and len(res[0]) == args.k
                    counts[t] += 1
                    i += nt

            t0 = time.perf_counter()
            threads = [threading.Thread(target=worker, args=(t,)) for t in range(nt)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            dt = time.perf_counter() - t0
            print("merged=%d threads=%3d: %8.1f QPS  (%d queries, %.2f s)" % (merged, nt, sum(counts) / dt, sum(counts), dt), flush=True)
    eng.set_option("coalesce_host", 1)
    eng.drop_table(name)


if __name__ == "__main__":
    main()
