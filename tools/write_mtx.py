#!/usr/bin/env python3
"""experiment helper: write a generator case as MatrixMarket.  usage: write_mtx.py fem|cage|rmat16 out.mtx"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bmsparse-spgemm-spmv_amd"))
import numpy as np, pandas as pd
from pybmsp import gen
n, _, r, c, v = {"fem": lambda: gen.fem_like(47, "27pt"), "cage": lambda: gen.cage_like(130228, 15.6), "rmat16": lambda: gen.rmat(16, 8)}[sys.argv[1]]()
with open(sys.argv[2], "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, r.size))
pd.DataFrame({"r": np.asarray(r, dtype=np.int64) + 1, "c": np.asarray(c, dtype=np.int64) + 1, "v": np.asarray(v, dtype=np.float64)}).to_csv(
    sys.argv[2], sep=" ", header=False, index=False, mode="a", float_format="%.9g")
