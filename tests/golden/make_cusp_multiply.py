#!/usr/bin/env python3
"""Generates tests/golden/cusp_multiply.json: the known-answer cases of the reference's vendored CUSP unit test
cusp/testing/multiply.cu:39-128 (TestSparseMatrixMatrixMultiply) and :384-461 (SpMV literal case).

  * the literal matrices A(3x2) ... F(2x3) are restated from multiply.cu:43-83 (data, typed in below);
  * G = poisson5pt(4,6), H = poisson5pt(8,3) restate cusp/gallery/detail/poisson.inl:29-47 + stencil.inl (pybmsp.gen.poisson);
  * I = random(24,24,150), J = random(24,24,50), K = random(24,12,20) restate cusp/gallery/detail/random.inl:30-60 with the
    C library's srand/rand (glibc here -- the sequence is pinned by committing the result);
  * for every compatible ordered pair the expected product is the DENSE product, computed here by a plain triple loop --
    the reference test's own method (multiply.cu:20-28: `cusp::multiply(A,B,C)` on array2d, compared with `==`).
Nothing here calls the oracle or the HIP path.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bmsparse-spgemm-spmv_amd"))
from pybmsp import gen  # noqa: E402

LITERAL = {  # multiply.cu:43-83
    "A": (3, 2, [[1.0, 2.0], [3.0, 0.0], [5.0, 6.0]]),
    "B": (2, 4, [[0.0, 2.0, 3.0, 4.0], [5.0, 0.0, 0.0, 8.0]]),
    "C": (2, 2, [[0.0, 0.0], [3.0, 5.0]]),
    "D": (2, 1, [[2.0], [3.0]]),
    "E": (2, 2, [[0.0, 0.0], [0.0, 0.0]]),
    "F": (2, 3, [[0.0, 1.5, 3.0], [0.5, 0.0, 0.0]]),
}


def dense_of(m, n, r, c, v):
    d = [[0.0] * n for _ in range(m)]
    for i, j, x in zip(r.tolist(), c.tolist(), v.tolist()):
        d[i][j] = x
    return d


def main():
    mats = {}
    for k, (m, n, d) in LITERAL.items():
        mats[k] = {"rows": m, "cols": n, "dense": d}
    for k, (kind, a, b) in {"G": ("5pt", 4, 6), "H": ("5pt", 8, 3)}.items():
        n, _, r, c, v = gen.poisson(kind, a, b)
        mats[k] = {"rows": n, "cols": n, "dense": dense_of(n, n, r, c, v), "generator": "poisson5pt(%d,%d)" % (a, b)}
    for k, (m, n, s) in {"I": (24, 24, 150), "J": (24, 24, 50), "K": (24, 12, 20)}.items():
        _, _, r, c, v = gen.cusp_random(m, n, s)
        mats[k] = {"rows": m, "cols": n, "dense": dense_of(m, n, r, c, v), "generator": "random(%d,%d,%d)" % (m, n, s)}
    products = []
    names = sorted(mats)
    for ln in names:
        for rn in names:
            L, R = mats[ln], mats[rn]
            if L["cols"] != R["rows"]:
                continue
            P = [[sum(L["dense"][i][k] * R["dense"][k][j] for k in range(L["cols"])) for j in range(R["cols"])] for i in range(L["rows"])]
            products.append({"left": ln, "right": rn, "dense": P})
    # SpMV literal case (multiply.cu:441-461): 5x4 matrix, x[i] = i % 10, y pre-filled with 10 (overwritten)
    spmv_dense = [[13, 80, 0, 0], [0, 27, 0, 0], [55, 0, 24, 42], [0, 69, 0, 83], [0, 0, 27, 0]]
    x = [float(i % 10) for i in range(4)]
    y = [sum(a * b for a, b in zip(row, x)) for row in spmv_dense]
    out = {"source": "cusp/testing/multiply.cu:39-128,384-461; gallery generators restated (see make_cusp_multiply.py)",
           "matrices": mats, "products": products, "spmv": {"dense": spmv_dense, "x": x, "y": y}}
    with open(os.path.join(HERE, "cusp_multiply.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("matrices", len(mats), "products", len(products))


if __name__ == "__main__":
    main()
