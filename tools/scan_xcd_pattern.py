"""Diagnostic: is the order in which the XCDs get through equal shares of a scan the same from launch to launch,
and does workgroup b run on XCD b % 8?  Needs the library built with -DVDB_SCAN_STAMPS (see tools/scan_waves.py)."""
import ctypes as C, os, sys, collections, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
eng = Engine(0); lib = eng.lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
rows, dim = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), (int(sys.argv[2]) if len(sys.argv) > 2 else 768)
DT = _lib.BF16 if (len(sys.argv) > 3 and sys.argv[3] == 'bf16') else _lib.F32
eng.create_table("t", dim, _lib.METRIC_IP, DT); eng.reserve("t", rows); eng.insert_synthetic("t", rows, seed=0, first_row=0)
q = torch.empty((16, dim), dtype=torch.float32, device=dev); eng.synth_fill_dev(q.data_ptr(), 16, dim, seed=1, first_row=0, stream=s)
ids = torch.empty((1, 10), dtype=torch.int64, device=dev); sc = torch.empty((1, 10), dtype=torch.float32, device=dev); cn = torch.empty((1,), dtype=torch.int32, device=dev)
n = 4096
for rnd in range(6):
    for i in range(1 + rnd % 3):
        eng.search_dev("t", 1, q[i].data_ptr(), 10, ids.data_ptr(), sc.data_ptr(), cn.data_ptr(), s)
    torch.cuda.synchronize()
    ends = (C.c_uint * n)(); xcc = (C.c_uint * n)()
    lib.vdb_debug_scan_waves(ends, xcc, n)
    e = [ends[i] / 100.0 for i in range(n)]
    x = [xcc[i] & 0xf for i in range(n)]
    by = collections.defaultdict(list)
    for i in range(n): by[x[i]].append(e[i])
    mism = sum(1 for i in range(n) if x[i] != (i // 8) % 8)
    be = [max(e[b * 8:(b + 1) * 8]) for b in range(n // 8)]
    bx = [x[b * 8] for b in range(n // 8)]
    bs = sorted(be)
    print("   block ends: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; within-block spread median %.0f; first blocks' xcc: %s" % (
        bs[0], bs[len(bs) // 10], bs[len(bs) // 2], bs[len(bs) * 9 // 10], bs[-1],
        st.median([max(e[b * 8:(b + 1) * 8]) - min(e[b * 8:(b + 1) * 8]) for b in range(n // 8)]), bx[:16]))
    print("launch %d: xcc medians %s  max %s | waves whose xcc != block %% 8: %d" % (
        rnd, " ".join("%3.0f" % st.median(by[k]) for k in sorted(by)), " ".join("%3.0f" % max(by[k]) for k in sorted(by)), mism), flush=True)
