#!/usr/bin/env python3
"""A/B of one library option on the batched path for several batch sizes, the values interleaved
in one process (GPU only); answers must agree between the values.
Environment: ROWS, DIM, DTYPE (f32|f16|bf16), METRICS (comma list), OPT (option name, default
gemm32s_threads), VARIANTS (comma list of its values), RESET (value restored at the end), NQS."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib  # noqa: E402
from vectordb_amd.engine import Engine  # noqa: E402

rows, dim = int(os.environ.get("ROWS", 20_000_000)), int(os.environ.get("DIM", 768))
dtype = {"f32": _lib.F32, "f16": _lib.F16, "bf16": _lib.BF16}[os.environ.get("DTYPE", "f16")]
opt = os.environ.get("OPT", "gemm32s_threads")
metrics = os.environ.get("METRICS", "l2").split(",")
variants = [int(v) for v in os.environ.get("VARIANTS", "512,256").split(",")]
nqs = [int(v) for v in os.environ.get("NQS", "128,256,1024").split(",")]
codes = {"l2": _lib.METRIC_L2, "cosine": _lib.METRIC_COSINE, "ip": _lib.METRIC_IP}
eng = Engine(0)
st = torch.cuda.current_stream().cuda_stream
for mname in metrics:
    name = "ab_" + mname
    eng.create_table(name, dim, codes[mname], dtype)
    eng.reserve(name, rows)
    eng.insert_synthetic(name, rows, 0, 0)
    for nq in nqs:
        q = torch.empty((nq, dim), dtype=torch.float32, device="cuda")
        eng.synth_fill_dev(q.data_ptr(), nq, dim, 1, 0, st)
        oi = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
        os_ = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
        oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
        res = {v: [] for v in variants}
        ref = None
        for rnd in range(3):
            for v in variants:
                eng.set_option(opt, v)
                eng.search_dev(name, nq, q.data_ptr(), 10, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    eng.search_dev(name, nq, q.data_ptr(), 10, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
                torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t0) / 2 * 1e3)
                got = (oi.cpu(), os_.cpu())
                if ref is None:
                    ref = got
                elif not (torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])):
                    print("MISMATCH between variants at nq=%d variant %d" % (nq, v))
        gb = rows * dim * (4 if dtype == _lib.F32 else 2) / 1e9
        for v in variants:
            m = statistics.median(res[v])
            print("%s nq=%4d variant %2d: batch %.2f ms  %.0f GB/s row traffic  %.0f TFLOP/s  %.0f QPS" %
                  (mname, nq, v, m, gb / (m / 1e3), 2.0 * rows * dim * nq / (m / 1e3) / 1e12, nq / (m / 1e3)))
    eng.drop_table(name)
eng.set_option(opt, int(os.environ.get("RESET", variants[0])))
print("fallbacks", eng.stats()[1])
