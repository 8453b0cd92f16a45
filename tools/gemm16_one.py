#!/usr/bin/env python3
"""One 16-bit batched search configuration, for profiling runs (GPU only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
rows, dim, nq = int(os.environ.get("ROWS", 10_000_000)), 768, int(os.environ.get("NQ", 128))
variant = int(os.environ.get("VARIANT", 1))
eng = Engine(0)
st = torch.cuda.current_stream().cuda_stream
eng.create_table("p", dim, _lib.METRIC_COSINE, _lib.F16)
eng.reserve("p", rows); eng.insert_synthetic("p", rows, 0, 0)
eng.set_option("gemm16_variant", variant)
q = torch.empty((nq, dim), dtype=torch.float32, device="cuda"); eng.synth_fill_dev(q.data_ptr(), nq, dim, 1, 0, st)
oi = torch.empty((nq, 10), dtype=torch.int64, device="cuda"); os_ = torch.empty((nq, 10), dtype=torch.float32, device="cuda"); oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
for _ in range(3):
    eng.search_dev("p", nq, q.data_ptr(), 10, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
torch.cuda.synchronize()
