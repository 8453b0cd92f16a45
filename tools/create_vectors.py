#!/usr/bin/env python3
"""Seeded, parameterised counterpart of /root/reference/createVector.py (which prints one
unseeded 8-dimensional U[0,1) vector as {"name": "v1", "vector": [...]}).

  python tools/create_vectors.py --dims 384 --count 3            # JSON lines on stdout
  python tools/create_vectors.py --dims 384 --count 10000 --post http://127.0.0.1:5001 --dbname test

Vectors are drawn exactly as the reference draws them (numpy `rand`, float64 U[0,1)), from
`default_rng(seed)`; names are v0, v1, ...  With --post each vector is PUT to /insert as
curl-com.txt:1 does."""
import argparse
import json
import sys
import urllib.request

import numpy as np


def vectors(dims: int, count: int, seed: int, first: int = 0):
    rng = np.random.default_rng(seed)
    for i in range(count):
        yield {"name": "v%d" % (first + i), "vector": rng.random(dims).tolist()}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, default=8)
    ap.add_argument("--count", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--post", help="base URL of the service; PUT each vector to /insert")
    ap.add_argument("--dbname", default="test")
    args = ap.parse_args()
    for obj in vectors(args.dims, args.count, args.seed):
        body = json.dumps(obj)
        if args.post:
            req = urllib.request.Request("%s/insert?dbname=%s" % (args.post.rstrip("/"), args.dbname),
                                         data=body.encode("utf-8"), method="PUT",
                                         headers={"Content-Type": "application/json"})
            with urllib.request.urlopen(req) as resp:
                resp.read()
        else:
            print(body)
    return 0


if __name__ == "__main__":
    sys.exit(main())
