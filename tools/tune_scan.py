#!/usr/bin/env python3
"""A/B of the scan kernel's launch tunables, interleaved rounds in one process (GPU only).
Prints achieved GB/s per variant (median and best over rounds) next to the plain-read ceiling."""
import argparse
import itertools
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _lib  # noqa: E402
from vectordb_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-ceiling", action="store_true")
    args = ap.parse_args()
    dt, esz = {"f32": (_lib.F32, 4), "bf16": (_lib.BF16, 2), "f16": (_lib.F16, 2)}[args.dtype]
    eng = Engine(0)
    eng.create_table("tune", args.dim, _lib.METRIC_COSINE, dt)
    eng.reserve("tune", args.rows)
    eng.insert_synthetic("tune", args.rows, 0, 0)
    nbytes = args.rows * ((args.dim * esz + 15) // 16 * 16)
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((64, args.dim), dtype=torch.float32, device="cuda")
    eng.synth_fill_dev(q.data_ptr(), 64, args.dim, 1, 0, st)
    oi = torch.empty((1, args.k), dtype=torch.int64, device="cuda")
    os_ = torch.empty((1, args.k), dtype=torch.float32, device="cuda")
    oc = torch.empty((1,), dtype=torch.int32, device="cuda")

    variants = []
    for nt, (w, bpc), contig in itertools.product((1, 0), ((16, 1), (8, 2), (4, 4), (16, 2), (8, 4), (8, 3), (8, 6), (8, 8), (4, 8)), (0, 1)):
        variants.append(dict(scan_nt=nt, scan_waves=w, scan_blocks_per_cu=bpc, scan_contiguous=contig))
    results = {i: [] for i in range(len(variants))}
    for rnd in range(args.rounds):
        for i, v in enumerate(variants):
            for key, val in v.items():
                eng.set_option(key, val)
            eng.search_dev("tune", 1, q.data_ptr(), args.k, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
            torch.cuda.synchronize()
            eng.prof_read()
            eng.prof_enable(True)
            for r in range(args.reps):
                eng.search_dev("tune", 1, q[r % 64].data_ptr(), args.k, oi.data_ptr(), os_.data_ptr(), oc.data_ptr(), st)
            torch.cuda.synchronize()
            eng.prof_enable(False)
            ms, n = eng.prof_read()
            results[i].append(nbytes / (ms / n / 1e3) / 1e9)
    print("scan variants, %d x %d %s (%.2f GB per scan): GB/s median / best" % (args.rows, args.dim, args.dtype, nbytes / 1e9))
    order = sorted(results, key=lambda i: -statistics.median(results[i]))
    for i in order:
        print("  %-78s %7.0f / %7.0f" % (variants[i], statistics.median(results[i]), max(results[i])))

    if args.no_ceiling:
        return
    # plain-read ceiling on a separate 8 GiB buffer
    buf = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
    buf.zero_()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print("plain 16-byte-per-lane read of 8 GiB: GB/s median / best")
    for nt, (blocks, threads) in itertools.product((1, 0), ((256, 1024), (512, 512), (1024, 256), (2048, 256), (4096, 256))):
        vals = []
        for rnd in range(args.rounds):
            eng.bench_read_dev(buf.data_ptr(), buf.numel(), nt, blocks, threads, st)
            ev0.record()
            for r in range(5):
                eng.bench_read_dev(buf.data_ptr(), buf.numel(), nt, blocks, threads, st)
            ev1.record()
            torch.cuda.synchronize()
            vals.append(buf.numel() * 5 / (ev0.elapsed_time(ev1) / 1e3) / 1e9)
        print("  nt=%d blocks=%d threads=%d %38s %7.0f / %7.0f" % (nt, blocks, threads, "", statistics.median(vals), max(vals)))


if __name__ == "__main__":
    main()
