#!/usr/bin/env python3
"""MatrixMarket ingestion and the binary cache, timed on the webbase-1M-like matrix written as a .mtx file.
Compares the library's parallel parser + device builder with the oracle's restatement of the reference's stream reader (tests only)."""
import sys, os, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bmsparse-spgemm-spmv_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np, pybmsp as B
from pybmsp import gen
n, _, r, c, v = gen.rmat(20, 2.0, seed=1)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
path = os.path.join(d, "webbase_like.mtx")
t0 = time.time()
with open(path, "w") as f:
    f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, r.size))
    np.savetxt(f, np.column_stack([r + 1, c + 1, v]), fmt="%d %d %.17g")
print("wrote %s: %.1f MB in %.1f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0))
for it in range(2):
    t0 = time.time(); A = B.BmSpMatrix.from_mtx(path); B.synchronize(); t1 = time.time()
    print("bmsp_matrix_from_mtx (parallel parse + device build): %.0f ms  (%d values, %d tiles)" % ((t1 - t0) * 1e3, A.nnz, A.block_num))
cache = os.path.join(d, "webbase_like.bmsp")
t0 = time.time(); A.save(cache); t1 = time.time()
print("bmsp_matrix_save: %.0f ms (%.1f MB)" % ((t1 - t0) * 1e3, os.path.getsize(cache) / 1e6))
t0 = time.time(); A2 = B.BmSpMatrix.load(cache); B.synchronize(); t1 = time.time()
print("bmsp_matrix_load: %.0f ms" % ((t1 - t0) * 1e3))
for x, y in zip(A.host_arrays(), A2.host_arrays()):
    assert np.array_equal(x, y)
# device-resident conversions
def timed(f, reps=5):
    f(); B.synchronize()
    t0 = time.time()
    for _ in range(reps): out = f()
    B.synchronize()
    return (time.time() - t0) * 1e3 / reps, out
ms, csr = timed(lambda: A.to_csr_device())
print("bmsp_matrix_to_csr_device: %.2f ms" % ms)
ms, A3 = timed(lambda: B.BmSpMatrix.from_csr_device(n, n, *csr))
print("bmsp_matrix_from_csr_device: %.2f ms" % ms)
for x, y in zip(A.host_arrays(), A3.host_arrays()):
    assert np.array_equal(x, y)
ms, _ = timed(lambda: B.BmSpMatrix.from_coo(n, n, r, c, v))
print("bmsp_matrix_from_coo (host triples -> device build): %.2f ms" % ms)
try:
    import oracle as O
    t0 = time.time(); coo = O.mtx_read(path); t1 = time.time()
    print("oracle restatement of the reference's stream reader (host, 1 thread): %.0f ms" % ((t1 - t0) * 1e3))
except Exception as e:
    print("oracle reader skipped:", e)
