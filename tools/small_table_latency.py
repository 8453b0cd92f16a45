"""Blocking vdb_search latency over small tables (the reference's own workload shape: 10 000 x 384), one query per call
from host memory: where the fixed cost of a call goes once the scan itself is a few microseconds.
  python tools/small_table_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
eng = Engine(0)
for rows, dim in [(10_000, 384), (100_000, 384), (1_000, 8)]:
    name = "lat_%d_%d" % (rows, dim)
    eng.create_table(name, dim, _lib.METRIC_IP, _lib.F32)
    eng.insert_synthetic(name, rows, seed=0, first_row=0)
    q = np.random.default_rng(1).random((512, dim)).astype(np.float32)
    for i in range(50):
        eng.search(name, q[i:i + 1], 10)
    lat = []
    for i in range(50, 512):
        t = time.perf_counter()
        eng.search(name, q[i:i + 1], 10)
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.sort(np.array(lat))
    print("%7d x %3d: p50 %.1f us  p10 %.1f  p90 %.1f  p99 %.1f" % (rows, dim, np.percentile(lat, 50), np.percentile(lat, 10), np.percentile(lat, 90), np.percentile(lat, 99)), flush=True)
    eng.drop_table(name)
