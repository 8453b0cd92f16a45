"""Diagnostic: how long the waves of a scan take, by XCC and CU.  Needs a library built with -DVDB_SCAN_STAMPS:
  cd vectordb_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVDB_SCAN_STAMPS -shared \
      -o ../libvdb_stamps.so vdb_api.hip kernels_scan.hip kernels_store.hip kernels_gemm.hip
  VDB_LIBRARY=$PWD/vectordb_amd/libvdb_stamps.so python tools/scan_waves.py"""
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
for i in range(8):
    eng.search_dev("t", 1, q[i].data_ptr(), 10, ids.data_ptr(), sc.data_ptr(), cn.data_ptr(), s)
torch.cuda.synchronize()
n = 4096
ends = (C.c_uint * n)(); xcc = (C.c_uint * n)()
lib.vdb_debug_scan_waves(ends, xcc, n)
e = [ends[i] / 100.0 for i in range(n)]
x = [xcc[i] & 0xf for i in range(n)]
hw = [xcc[i] >> 4 for i in range(n)]
print("all waves: min %.1f median %.1f max %.1f us" % (min(e), st.median(e), max(e)))
by = collections.defaultdict(list)
for i in range(n): by[x[i]].append(e[i])
for k in sorted(by): print("xcc %d: %4d waves, min %.0f median %.0f max %.0f" % (k, len(by[k]), min(by[k]), st.median(by[k]), max(by[k])))
# by block (8 waves each)
bm = [st.mean(e[b * 8:(b + 1) * 8]) for b in range(n // 8)]
bs = [max(e[b * 8:(b + 1) * 8]) - min(e[b * 8:(b + 1) * 8]) for b in range(n // 8)]
print("per block: mean of means %.0f, spread of block means %.0f..%.0f, median within-block spread %.0f us" % (st.mean(bm), min(bm), max(bm), st.median(bs)))
# by CU id within xcc (HW_ID bits: cu_id [11:8], sh_id [12], se_id [15:13] on gfx9)
cu = collections.defaultdict(list)
for i in range(n): cu[(x[i], (hw[i] >> 13) & 7, (hw[i] >> 12) & 1, (hw[i] >> 8) & 15)].append(e[i])
cm = sorted((st.mean(v), k, len(v)) for k, v in cu.items())
print("CUs seen:", len(cm), " fastest:", ["%.0f" % c[0] for c in cm[:5]], " slowest:", ["%.0f" % c[0] for c in cm[-5:]])
se = collections.defaultdict(list)
for i in range(n): se[(hw[i] >> 13) & 7].append(e[i])
for k in sorted(se): print("se %d: %d waves median %.0f" % (k, len(se[k]), st.median(se[k])))
