"""Time of a query that is forced through the exact pass (option force_exact) on 10 M x 768 fp32 rows:
fast pass + exact pass per query, from device-resident queries.
  python tools/exact_pass_time.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k = 768, 10
eng = Engine(0)
eng.create_table("ex", dim, _lib.METRIC_COSINE, _lib.F32)
eng.insert_synthetic("ex", rows, seed=0, first_row=0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
q = torch.empty((1, dim), dtype=torch.float32, device=dev)
eng.synth_fill_dev(q.data_ptr(), 1, dim, seed=9, first_row=0, stream=st)
ids = torch.zeros((1, k), dtype=torch.int64, device=dev)
sc = torch.zeros((1, k), dtype=torch.float32, device=dev)
cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
for force in (0, 1):
    eng.set_option("force_exact", force)
    for _ in range(3):
        eng.search_dev("ex", 1, q.data_ptr(), k, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        eng.search_dev("ex", 1, q.data_ptr(), k, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    print("force_exact=%d: %.3f ms per query (%d x %d fp32)" % (force, (time.perf_counter() - t) / n * 1e3, rows, dim), flush=True)
eng.set_option("force_exact", 0)
eng.drop_table("ex")
