"""Latency of vdb_insert with one row per call (what PUT /insert does, /root/reference/app.py:315-325) into a table
of 10 000 x 384 rows, through the Python binding.
  python tools/single_insert_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
eng = Engine(0)
for metric, mname in ((_lib.METRIC_IP, "ip"), (_lib.METRIC_COSINE, "cosine")):
    name = "ins_" + mname
    eng.create_table(name, 384, metric, _lib.F32)
    eng.insert_synthetic(name, 10_000, seed=0, first_row=0)
    rows = np.random.default_rng(2).random((600, 384)).astype(np.float32)
    ids = np.arange(1 << 40, (1 << 40) + 600, dtype=np.int64)
    lat = []
    for i in range(600):
        t = time.perf_counter()
        eng.insert(name, rows[i:i + 1], ids[i:i + 1])
        if i >= 100:
            lat.append((time.perf_counter() - t) * 1e6)
    lat = np.sort(np.array(lat))
    print("%-6s one row per call: p50 %.1f us  p10 %.1f  p90 %.1f  p99 %.1f" % (mname, np.percentile(lat, 50), np.percentile(lat, 10), np.percentile(lat, 90), np.percentile(lat, 99)), flush=True)
    eng.drop_table(name)
