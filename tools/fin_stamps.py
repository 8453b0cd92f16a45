"""Stage times of finalize_kernel on a small table (diagnostic library built with -DVDB_FIN_STAMPS):
  hipcc ... -DVDB_FIN_STAMPS -o vectordb_amd/libvdb_stamps.so; VDB_LIBRARY=vectordb_amd/libvdb_stamps.so python tools/fin_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectordb_amd import _lib
from vectordb_amd.engine import Engine
eng = Engine(0)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.vdb_debug_read.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
for rows, dim, k in [(10_000, 384, 10), (100_000, 384, 10), (10_000_000 // 8, 768, 10)]:
    name = "st_%d_%d" % (rows, dim)
    eng.create_table(name, dim, _lib.METRIC_IP, _lib.F32)
    eng.insert_synthetic(name, rows, seed=0, first_row=0)
    q = np.random.default_rng(1).random((64, dim)).astype(np.float32)
    acc = []
    for i in range(64):
        eng.search(name, q[i:i + 1], k)
        buf = (ctypes.c_longlong * 32)()
        assert lib.vdb_debug_read(buf) == 0
        s = [buf[8 + j] for j in range(6)]
        c = [buf[16 + j] for j in range(6)]
        if i >= 8:
            acc.append([(s[j + 1] - s[j]) * 10 for j in range(5)] + [(c[5] - c[0]) / max(1, (s[5] - s[0]) * 10)])
    a = np.median(np.array(acc), axis=0)
    print("%8d x %d: heads+bound %.0f ns  walk %.0f  select %.0f  re-score %.0f  sort+write+certify %.0f  (sum %.0f)  shader clock %.2f GHz" % (rows, dim, a[0], a[1], a[2], a[3], a[4], a[:5].sum(), a[5]), flush=True)
    eng.drop_table(name)
