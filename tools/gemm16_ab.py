#!/usr/bin/env python3
"""A/B of the two 16-bit batched GEMM kernels (gemm16_variant 0 = LDS tile, 1 = streaming) for
several batch sizes and metrics, interleaved in one process (GPU only)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib
from vectordb_amd.engine import Engine

rows, dim = int(os.environ.get("ROWS", 20_000_000)), 768
eng = Engine(0)
st = torch.cuda.current_stream().cuda_stream
for metric, mname in ((_lib.METRIC_L2, "l2"), (_lib.METRIC_COSINE, "cosine")):
    name = "ab_" + mname
    eng.create_table(name, dim, metric, _lib.F16)
    eng.reserve(name, rows)
    eng.insert_synthetic(name, rows, 0, 0)
    for nq in (128, 256, 1024):
        q = torch.empty((nq, dim), dtype=torch.float32, device="cuda")
        eng.synth_fill_dev(q.data_ptr(), nq, dim, 1, 0, st)
        oi = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
        os_ = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
        oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
        res = {0: [], 1: []}
        for rnd in range(3):
            for v in (0, 1):
                eng.set_option("gemm16_variant", v)
                eng.search_dev(name, nq, q.data_ptr(), 10, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
                torch.cuda.synchronize(); eng.prof_read(); eng.prof_enable(True)
                for _ in range(2):
                    eng.search_dev(name, nq, q.data_ptr(), 10, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
                torch.cuda.synchronize(); eng.prof_enable(False)
                ms, n = eng.prof_read()
                res[v].append(ms / n)
        gb = rows * dim * 2 / 1e9
        for v in (0, 1):
            m = statistics.median(res[v])
            print("%s nq=%4d variant %d: gemm %.2f ms  %.0f GB/s per pass-equivalent  %.0f TFLOP/s" %
                  (mname, nq, v, m, gb / (m / 1e3), 2.0 * rows * dim * nq / (m / 1e3) / 1e12))
    eng.drop_table(name)
print("fallbacks", eng.stats()[1])
