"""Read-bandwidth ceiling at the sizes of the short scans: the plain-read kernel (vdb_bench_read_dev: the scan's
access shape, no scoring, no candidate lists) over buffers of 0.4 to 30.7 GB, timed with events.
  python tools/read_ceiling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd.engine import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
for gb, blocks, threads in [(0.384, 512, 512), (1.536, 512, 512), (3.072, 512, 512), (3.072, 1024, 512), (3.072, 256, 1024),
                            (3.84, 512, 512), (7.68, 512, 512), (30.72, 512, 512), (30.72, 1024, 512)]:
    n = int(gb * 1e9) // 16 * 16
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    buf.random_(0, 255)
    torch.cuda.synchronize()
    for _ in range(3):
        eng.bench_read_dev(buf.data_ptr(), n, 1, blocks, threads, st.cuda_stream)
    reps = 20
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(st)
        eng.bench_read_dev(buf.data_ptr(), n, 1, blocks, threads, st.cuda_stream)
        b.record(st)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    print("%6.3f GB  %4d blocks x %4d threads: median %.4f ms = %.2f TB/s (min %.4f ms)" % (gb, blocks, threads, med, n / med / 1e9, ts[0]), flush=True)
    del buf
