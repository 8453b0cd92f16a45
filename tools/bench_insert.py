#!/usr/bin/env python3
"""Throughput of bulk host inserts (vdb_insert with n rows per call) and of snapshots (GPU only)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib  # noqa: E402
from vectordb_amd.engine import Engine  # noqa: E402

n, d = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("DIM", 768))
eng = Engine(0)
rng = np.random.default_rng(0)
rows = rng.standard_normal((n, d), dtype=np.float32)
ids = np.arange(n, dtype=np.int64)
for metric, mname in ((_lib.METRIC_IP, "ip"), (_lib.METRIC_COSINE, "cosine")):
    for call in (n, 65536):
        eng.create_table("ins", d, metric, _lib.F32)
        eng.reserve("ins", n)
        t0 = time.perf_counter()
        for i0 in range(0, n, call):
            eng.insert("ins", rows[i0:i0 + call], ids[i0:i0 + call])
        dt = time.perf_counter() - t0
        print("%s insert %d x %d fp32, %d rows per call: %.2f s  %.2f GB/s  %.0f rows/s" % (mname, n, d, call, dt, rows.nbytes / dt / 1e9, n / dt))
        if call == n and metric == _lib.METRIC_IP:
            path = "/tmp/ins.vdbsnap"
            t0 = time.perf_counter(); eng.save_table("ins", path); ts = time.perf_counter() - t0
            eng.drop_table("ins")
            t0 = time.perf_counter(); eng.load_table("ins", path); tl = time.perf_counter() - t0
            print("snapshot save %.2f s (%.2f GB/s), load %.2f s (%.2f GB/s)" % (ts, rows.nbytes / ts / 1e9, tl, rows.nbytes / tl / 1e9))
            os.remove(path)
        eng.drop_table("ins")
