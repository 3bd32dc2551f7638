import sys, os
sys.path.insert(0, "bmsparse-spgemm-spmv_amd")
import numpy as np
import pybmsp as B
os.environ["BMSP_SPGEMM_ROWMERGE"] = "1"
def run(A, Bm, name):
    a = B.BmSpMatrix.from_coo(*A, dtype=0)
    b = B.BmSpMatrix.from_coo(*Bm, transposed=True, dtype=0)
    C, st = B.spgemm(a, b, tc_version=5)
    k, bm, o, v = C.host_arrays()
    print(name, "path", st["sort_path"], "variant", st["mac_variant"])
    import scipy.sparse as sp
    As = sp.csr_matrix((A[4], (A[2], A[3])), shape=(A[0], A[1])); Bs = sp.csr_matrix((Bm[4], (Bm[2], Bm[3])), shape=(Bm[0], Bm[1]))
    ref = (As @ Bs).toarray()
    got = np.zeros_like(ref)
    for t in range(len(k)):
        br, bc = k[t] >> 32, k[t] & 0xffffffff
        pos = [p for p in range(64) if (int(bm[t]) >> (63 - p)) & 1]
        for r, p in enumerate(pos):
            got[8 * br + p // 8, 8 * bc + p % 8] = v[o[t] + r]
    print("max err", np.abs(got - ref).max())
    if np.abs(got - ref).max() > 0:
        np.set_printoptions(linewidth=250, precision=1, suppress=True)
        print("ref\n", ref[:16, :16]); print("got\n", got[:16, :16])
n = 16
r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
full = (n, n, r.ravel(), c.ravel(), (r * 16 + c + 1).ravel().astype(np.float64))
eye = (n, n, np.arange(n), np.arange(n), np.ones(n))
run(full, eye, "A=full B=I")
run(eye, full, "A=I B=full")
run(full, full, "full^2")
