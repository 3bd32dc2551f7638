import sys, os, time
sys.path.insert(0, "/root/repo/bmsparse-spgemm-spmv_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, pybmsp as B
from pybmsp import gen
n, _, r, c, v = gen.rmat(20, 2.0)
first = B.BmSpMatrix.from_coo(n, n, r, c, v)
for env in ("cache", "nocache"):
    if env == "nocache": os.environ["BMSP_SPMV_NO_POSCACHE"] = "1"
    ts = []
    for i in range(5):
        m = first.clone(); B.synchronize()
        t0 = time.perf_counter(); m.prepare(1); B.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(env, "prepare(1) ms:", ["%.3f" % t for t in ts])
