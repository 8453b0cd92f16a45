"""Writes tests/golden/search_golden.json: expected ids and score bit patterns of the numpy
oracle for the small cases of tests/cases.py (regression pins of this build's own
specification -- the reference holds no expected search results)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import cases  # noqa: E402


def main():
    out = []
    for case in cases.GOLDEN_CASES:
        _, _, _, res = cases.run_oracle(case)
        out.append({
            "case": list(case),
            "ids": [[int(x) for x in r[0]] for r in res],
            "score_bits": [[int(x) for x in r[1].view(np.uint32)] for r in res],
        })
    with open(os.path.join(HERE, "search_golden.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
