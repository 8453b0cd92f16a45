"""Turn a `rocprofv3 --pmc FETCH_SIZE` counter CSV into an entry of profiles/traffic.json.

usage: python tools/pmc_to_traffic.py <counter_collection.csv> <kernel substring> <traffic key> <source text> [min KiB] [group]

FETCH_SIZE is reported in KiB; gfx950 counts a 128-byte request of a wide coalesced read as 64 bytes,
so bytes = value * 1024 * 2.  Launches below `min KiB` (sampled passes, warm-up fragments) are left out.
`group` > 1: a pass consists of that many consecutive launches (the batched path's last pass runs as two),
whose counters are added.
"""
import collections
import csv
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path, kern, key, source = sys.argv[1:5]
    floor = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    group = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = [v for _, v in sorted(per.items(), key=lambda kv: int(kv[0])) if v >= floor]
    if group > 1:
        vals = [sum(vals[i:i + group]) for i in range(0, len(vals) - group + 1, group)]
    med = statistics.median(vals)
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tj_path))
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    tj[key] = {
        "counter": "FETCH_SIZE (KiB, rocprofv3 --pmc, own pass)",
        "median_counter_value": med,
        "min_counter_value": min(vals),
        "max_counter_value": max(vals),
        "launches": len(vals),
        "correction": "gfx950 counts a 128-byte request of a wide coalesced read as 64 bytes: bytes = FETCH_SIZE * 1024 * 2",
        "hbm_read_bytes_per_launch": med * 1024 * 2,
        "kernels_at_commit": commit,
        "source": source,
    }
    json.dump(tj, open(tj_path, "w"), indent=1)
    print(key, len(vals), med * 2048 / 1e9, "GB")


main()
