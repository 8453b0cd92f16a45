"""Generates tests/golden/hash_vector.json by running the reference's own `hash_vector`.

Run in the build container only (it reads /root/reference/app.py).  app.py cannot be imported
as a module: its first import is the absent `milvus` client and importing it starts the server
(app.py:441-443).  The function is pure (hashlib only), so it is compiled from the function's
own syntax-tree node and called on the inputs below; only inputs and outputs are stored.
"""
import ast
import hashlib
import json
import os

import numpy as np

REF = "/root/reference/app.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_vector.json")


def reference_hash_vector():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "hash_vector")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"hashlib": hashlib}
    exec(compile(mod, REF, "exec"), ns)
    return ns["hash_vector"]


def main():
    hv = reference_hash_vector()
    cases = [
        # the one request the reference documents (curl-com.txt:1)
        [0.00793861557915232, 0.9997302132320715, 0.37542373014507036, 0.6712801110144234,
         0.35517993228837497, 0.4425925168063203, 0.48268047072393094, 0.15620825516058134],
        [0.0] * 8,
        [1.0, 2.0, 3.0, 4.0],
        [0.123456, -0.000004],
        [],
        [-1.5, 2.25, 1e-6, 123456.789012],
        [0.999995, 0.999994, -0.999995],
    ]
    rng = np.random.default_rng(0)
    for d in (8, 384, 768):
        cases.append(rng.random(d).tolist())          # createVector.py:6-8 draws U[0,1) float64
    golden = [{"vector": v, "hash": hv(v)} for v in cases]
    with open(OUT, "w") as fh:
        json.dump(golden, fh)
    print("wrote", OUT, len(golden))


if __name__ == "__main__":
    main()
