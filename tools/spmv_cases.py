#!/usr/bin/env python3
"""experiment helper: SpMV kernel time (HIP events, HBM-resident rotation) on the generator structures.
usage: spmv_cases.py [case-substring[+case-substring...]] [variant]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bmsparse-spgemm-spmv_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pybmsp as B
from pybmsp import gen
import bench
cases = [("rmat(20,2)", lambda: gen.rmat(20, 2.0)), ("banded(1000000,8)", lambda: gen.banded(1000000, 8)), ("banded(500000,32)", lambda: gen.banded(500000, 32)),
         ("fem_like(47,27pt)", lambda: gen.fem_like(47, "27pt")), ("cage_like(1000000)", lambda: gen.cage_like(1000000)), ("rmat(20,16)", lambda: gen.rmat(20, 16.0))]
args = [a for a in sys.argv[1:]]
if args: cases = [c for c in cases if any(a in c[0] for a in args[0].split("+"))]
variant = int(args[1]) if len(args) > 1 else 0
for name, mk in cases:
    n, _, r, c, v = mk()
    first = B.BmSpMatrix.from_coo(n, n, r, c, v)
    del r, c, v
    S = bench.SpmvSet(B, np, first, x_kind="cusp")
    S.timed(10, variant)
    ms = min(S.timed(50, variant) for _ in range(3))
    warm = S.timed(50, variant, rotate=False)
    print("%-22s variant %d  blocks %8d  v/tile %5.1f  %7.2f us  (warm %7.2f)  %6.1f GB/s alg  frac %.3f" % (
        name, variant, S.info["block_num"], S.info["nnz"] / S.info["block_num"], ms * 1e3, warm * 1e3, S.alg_bytes / ms / 1e6, S.alg_bytes / ms / 1e6 / 8000), flush=True)
    del S, first
    B.check(B.lib().bmsp_trim_pool())
