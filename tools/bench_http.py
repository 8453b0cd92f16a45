#!/usr/bin/env python3
"""End-to-end timing of the reference's own workload shape through the HTTP routes (GPU only):
N vectors of U[0,1) values (createVector.py:6-8, seeded) inserted one per PUT /insert
(curl-com.txt:1), then Q single-vector PUT /lookup calls (cosine top-10), all through Flask's test
client in this process: JSON parsing, md5 ids, SQLite bookkeeping and the engine call included.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from server.app import create_app  # noqa: E402
from server.metastore import MetaStore  # noqa: E402


def run(rows=10_000, dim=384, queries=100):
    class A:
        pass
    args = A()
    args.rows, args.dim, args.queries = rows, dim, queries
    http = create_app(store=MetaStore()).test_client()
    name = "bench_http"
    http.get("/deletedb?dbname=" + name)
    http.get("/createdb?dbname=%s&dimensions=%d&metric=cosine" % (name, args.dim))
    rng = np.random.default_rng(0)
    vecs = rng.random((args.rows, args.dim))
    t0 = time.perf_counter()
    for i in range(args.rows):
        r = http.put("/insert?dbname=" + name, json={"name": "v%d" % i, "vector": vecs[i].tolist()})
        assert r.get_data(as_text=True) == "hej"
    t_ins = time.perf_counter() - t0
    qrng = np.random.default_rng(1)
    qs = qrng.random((args.queries, args.dim))
    lat = []
    for i in range(args.queries):
        t1 = time.perf_counter()
        r = http.put("/lookup?dbname=" + name, json={"vector": qs[i].tolist()})
        lat.append((time.perf_counter() - t1) * 1e3)
        assert r.status_code == 200 and len(json.loads(r.get_data(as_text=True))) >= 10
    lat = np.sort(np.array(lat[2:]))
    http.get("/deletedb?dbname=" + name)
    return {
        "workload": "%d x %d U[0,1) rows, one PUT /insert each; %d PUT /lookup (cosine top-10)" % (args.rows, args.dim, args.queries),
        "insert_rows_per_s": args.rows / t_ins,
        "lookup_qps": 1e3 / float(lat.mean()),
        "lookup_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99))},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=100)
    args = ap.parse_args()
    print(json.dumps(run(args.rows, args.dim, args.queries)))


if __name__ == "__main__":
    main()
