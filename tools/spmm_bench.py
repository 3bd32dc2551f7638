#!/usr/bin/env python3
"""experiment helper: SpMM throughput against k single-vector sweeps"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bmsparse-spgemm-spmv_amd"))
import numpy as np, pybmsp as B
from pybmsp import gen
n, _, r, c, v = gen.rmat(20, 2)
r = np.concatenate([r, np.arange(n, dtype=r.dtype)]); c = np.concatenate([c, np.arange(n, dtype=c.dtype)]); v = np.concatenate([v, np.ones(n)])
A = B.BmSpMatrix.from_coo(n, n, r, c, v)
nnz = A.nnz
x1 = B.DeviceArray.from_host(np.ones(n, np.float32))
y1 = B.spmv(A, x1)
def timeit(f, reps=20):
    f(); B.synchronize()
    e0, e1 = B.Event(), B.Event()
    e0.record()
    for _ in range(reps): f()
    e1.record(); B.synchronize()
    return e0.elapsed_ms(e1) / reps
t1 = timeit(lambda: B.spmv(A, x1, y1))
print("spmv: %.1f us" % (t1 * 1e3))
for k in (1, 4, 8, 16, 32, 64):
    X = B.DeviceArray.from_host(np.ones(n * k, np.float32))
    Y = B.DeviceArray(n * k, np.float32)
    t = timeit(lambda: B.spmm(A, X, k, Y))
    print("spmm k=%3d: %8.1f us  (%.2f us per vector, %.1fx vs k sweeps, %.1f GFLOP/s)" % (k, t * 1e3, t * 1e3 / k, t1 * k / t, 2.0 * nnz * k / t / 1e6))
