import sys, os
sys.path.insert(0, "bmsparse-spgemm-spmv_amd"); sys.path.insert(0, "oracle")
import numpy as np
import pybmsp as B
import oracle as O
from pybmsp import gen
O.lib()
n, _, r, c, v = gen.fem_like(10, "27pt")
for scale in (1.0, 1e-20, 3e-23):
    vv = (v * scale).astype(np.float32).astype(np.float64)
    A = B.BmSpMatrix.from_coo(n, n, r, c, vv, dtype=0); Bt = B.BmSpMatrix.from_coo(n, n, r, c, vv, transposed=True, dtype=0)
    res = {}
    for env in ("1", "0"):
        os.environ["BMSP_SPGEMM_ROWMERGE"] = env
        C, st = B.spgemm(A, Bt, tc_version=5)
        res[env] = (C.host_arrays()[3].copy(), st["sort_path"], st["mac_variant"])
    refA = O.bmsp_from_coo(O.Coo(n, n, r, c, vv), 0, False); refB = O.bmsp_from_coo(O.Coo(n, n, r, c, vv), 0, True)
    refC, _ = O.spgemm(refA, refB)
    rv = np.asarray(refC.values, dtype=np.float32)
    a, b = res["1"][0], res["0"][0]
    print("scale", scale, "paths", res["1"][1:], res["0"][1:], "strip==valu", np.array_equal(a.view(np.uint32), b.view(np.uint32)),
          "strip==oracle", np.array_equal(a.view(np.uint32), rv.view(np.uint32)), "valu==oracle", np.array_equal(b.view(np.uint32), rv.view(np.uint32)),
          "denormal results", int(((np.abs(rv) < 1.1754944e-38) & (rv != 0)).sum()), "zeros strip/oracle", int((a == 0).sum()), int((rv == 0).sum()))
