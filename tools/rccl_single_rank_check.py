#!/usr/bin/env python3
"""Runs the sharded search path through the real RCCL backend with a world of one rank (all a
one-GPU box allows): process-group creation with a device id, all_gather_into_tensor on the uint8
record under side streams, the merge, and teardown -- the calls bench.py makes with N ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib  # noqa: E402
from vectordb_amd.engine import Engine  # noqa: E402
from vectordb_amd.sharded import ShardedSearcher, gpu_backend  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
eng = Engine(0)
eng.create_table("r", 96, _lib.METRIC_COSINE, _lib.F32)
eng.insert_synthetic("r", 200_000, 0, 0)
q = torch.empty((8, 96), dtype=torch.float32, device=dev)
eng.synth_fill_dev(q.data_ptr(), 8, 96, 1, 0, torch.cuda.current_stream().cuda_stream)
ls, mg = gpu_backend(eng, "r", 1, 10, _lib.METRIC_COSINE)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
searchers = [ShardedSearcher(1, 10, dev, ls, mg) for _ in range(3)]
for s in searchers:
    s.force_collective = True            # the collective and the merge although the world has one rank
torch.cuda.synchronize()
outs = []
for i in range(8):
    with torch.cuda.stream(streams[i % 3]):
        out = searchers[i % 3].search(q[i:i + 1])                   # what bench.py calls per step for N > 1
        outs.append((out.ids.clone(), out.scores.clone()))
torch.cuda.synchronize()
# host cost of one step through the RCCL path (enqueue only, three streams in rotation): what the Python + c10d
# layer adds per query next to a shard scan of half a millisecond
import time
for s_ in searchers:
    s_.trace = []
t0 = time.perf_counter()
nrep = 300
for i in range(nrep):
    with torch.cuda.stream(streams[i % 3]):
        searchers[i % 3].search(q[i % 8:i % 8 + 1])
host_us = (time.perf_counter() - t0) / nrep * 1e6
torch.cuda.synchronize()
tr = np.array([t for s_ in searchers for t in s_.trace])
print("host time per step: %.1f us (local search %.1f, collective %.1f, merge %.1f: medians of %d steps)"
      % (host_us, np.median(tr[:, 0]), np.median(tr[:, 1]), np.median(tr[:, 2]), tr.shape[0]))
for s_ in searchers:
    s_.trace = None
dist.barrier()
ids, sc, cnt = eng.search("r", q.cpu().numpy(), 10)
ok = all(np.array_equal(o[0].cpu().numpy()[0], ids[i]) and np.array_equal(o[1].cpu().numpy()[0], sc[i]) for i, o in enumerate(outs))
dist.destroy_process_group()
print("RCCL single-rank path:", "ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
