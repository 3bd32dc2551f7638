import sys, os, time
sys.path.insert(0, "/root/repo/bmsparse-spgemm-spmv_amd")
import numpy as np, pybmsp as B
from pybmsp import gen
n, _, r, c, v = gen.fem_like(47, "27pt")
A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=B.F16)
At = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=B.F16)
B.synchronize()
for it in range(3):
    t0 = time.perf_counter(); Cm, st = B.spgemm(A, At, tc_version=5); B.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("call %d: wall %.3f ms, device total %.0f us, stages T1 %.0f T2 %.0f T3 %.0f T9 %.0f T7 %.0f path %d" % (it, dt, st["t_us"][0], st["t_us"][1], st["t_us"][2], st["t_us"][3], st["t_us"][9], st["t_us"][7], st["sort_path"]))
    del Cm
A2 = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=B.F16); At2 = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=B.F16); B.synchronize()
t0 = time.perf_counter(); Cm, st = B.spgemm(A2, At2, tc_version=5); B.synchronize(); print("fresh operands, warm process: wall %.3f ms device %.0f" % ((time.perf_counter() - t0) * 1e3, st["t_us"][0]))
t0 = time.perf_counter(); A2.prepare(2); At2.prepare(2); B.synchronize(); print("prepare again (cached) %.3f ms" % ((time.perf_counter() - t0) * 1e3))
A3 = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=B.F16); At3 = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=B.F16); B.synchronize()
t0 = time.perf_counter(); A3.prepare(2); At3.prepare(2); B.synchronize(); tp = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); Cm, st = B.spgemm(A3, At3, tc_version=5); B.synchronize(); print("fresh operands prepared first: prepare %.3f ms, product wall %.3f ms device %.0f" % (tp, (time.perf_counter() - t0) * 1e3, st["t_us"][0]))
