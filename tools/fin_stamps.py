import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
eng = Engine(0)
eng.create_table("t", 768, _lib.METRIC_COSINE, _lib.F32)
eng.insert_synthetic("t", 10_000_000, 0, 0)
st = torch.cuda.current_stream().cuda_stream
q = torch.empty((16, 768), dtype=torch.float32, device="cuda"); eng.synth_fill_dev(q.data_ptr(), 16, 768, 1, 0, st)
oi = torch.empty((1, 10), dtype=torch.int64, device="cuda"); os_ = torch.empty((1, 10), dtype=torch.float32, device="cuda"); oc = torch.empty((1,), dtype=torch.int32, device="cuda")
buf = (C.c_longlong * 32)()
for r in range(6):
    eng.search_dev("t", 1, q[r].data_ptr(), 10, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
    torch.cuda.synchronize()
    eng.lib.vdb_debug_read(buf)
    s = [buf[8 + i] for i in range(6)]
    print("S=%d  stage us:" % buf[20], ["%.1f" % ((s[i + 1] - s[i]) / 100.0) for i in range(5)], "total %.1f" % ((s[5] - s[0]) / 100.0))
