"""Prints the device timeline of the last few blocking calls from a rocprofv3 kernel trace (+ memory-copy trace):
start offset, duration and the gap to the previous operation, in microseconds.
  python tools/trace_timeline.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [calls] [ops per call]"""
import csv, glob, os, sys
d = sys.argv[1]
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ops.sort()
# a call starts at an operation that follows a gap of more than 12 us ... simpler: split at finalize kernels
idx = [i for i, o in enumerate(ops) if "finalize_kernel" in o[2]]
sel = idx[-(ncalls + 40):-40] if len(idx) > ncalls + 40 else idx[-ncalls:]
for i in sel:
    lo = i
    while lo > 0 and ops[lo][0] - ops[lo - 1][1] < 15000 and i - lo < 6: lo -= 1
    hi = i
    while hi + 1 < len(ops) and ops[hi + 1][0] - ops[hi][1] < 15000 and hi - i < 3: hi += 1
    t0 = ops[lo][0]
    print("call ending at op %d: %.1f us from first start to last end" % (i, (ops[hi][1] - t0) / 1e3))
    for j in range(lo, hi + 1):
        s, e, n = ops[j]
        print("   +%7.1f  dur %6.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - ops[j - 1][1]) / 1e3 if j > lo else 0.0, n))
