"""Blocking-call latency over 10 000 x 384 under the scan's launch options (A/B in one process).
  python tools/small_table_options_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
eng = Engine(0)
eng.create_table("ab", 384, _lib.METRIC_IP, _lib.F32)
eng.insert_synthetic("ab", 10_000, seed=0, first_row=0)
q = np.random.default_rng(1).random((700, 384)).astype(np.float32)
def run(label):
    for i in range(100):
        eng.search("ab", q[i:i + 1], 10)
    lat = []
    for i in range(100, 700):
        t = time.perf_counter()
        eng.search("ab", q[i:i + 1], 10)
        lat.append((time.perf_counter() - t) * 1e6)
    print("%-28s p50 %.1f us  p10 %.1f  p90 %.1f" % (label, np.percentile(lat, 50), np.percentile(lat, 10), np.percentile(lat, 90)), flush=True)
run("defaults")
for key, vals, default in (("fin_threads", (1024, 256), 512), ("scan_waves", (4, 16), 8), ("scan_dynamic", (0,), 1), ("scan_nt", (0,), 1), ("scan_contiguous", (0,), 1), ("scan_blocks_per_cu", (1, 2), 4)):
    for v in vals:
        eng.set_option(key, v)
        run("%s=%s" % (key, v))
    eng.set_option(key, default)
run("defaults again")
eng.drop_table("ab")
