#!/usr/bin/env python3
"""experiment helper: the sharded SpGEMM through the LOOPBACK transport on one GPU (P panels per round computed one after the other, a
"broadcast" = a device copy on the exchange stream): how much of the exchange hides behind the next round's products.
usage: shard_overlap.py [scale] [edge_factor] [P]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bmsparse-spgemm-spmv_amd"))
import numpy as np, pybmsp as B
from pybmsp import gen
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ef = float(sys.argv[2]) if len(sys.argv) > 2 else 8
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n, _, r, c, v = gen.rmat(scale, ef)
A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=B.F16).prepare(2)
Bt = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=B.F16).prepare(2)
comm = B.Comm.loopback(P)
for it in range(2):
    Cm, st = B.spgemm(A, Bt, tc_version=4)
    B.synchronize(); t0 = time.perf_counter(); Cm, st = B.spgemm(A, Bt, tc_version=4); B.synchronize(); single = time.perf_counter() - t0
print("single product %.3f ms, C: %d blocks %d values (%.1f MB)" % (single * 1e3, st["c_blocks"], st["c_nnz"], (24 * st["c_blocks"] + 4 * st["c_nnz"]) / 1e6))
for rounds, gather in ((1, True), (2, True), (4, True), (8, True), (1, False)):
    best = None
    for it in range(3):
        B.synchronize(); t0 = time.perf_counter()
        Cs, st, sh = B.spgemm_sharded(comm, A, Bt, tc_version=4, rounds=rounds, gather=gather)
        B.synchronize(); dt = time.perf_counter() - t0
        if it and (best is None or dt < best[0]): best = (dt, sh)
        del Cs
    dt, sh = best
    print("P=%d rounds=%d gather=%d: total %.3f ms | exchange %.3f ms, exposed %.3f ms, hidden %.2f | bytes %.1f MB" % (
        P, sh["rounds"], sh["gathered"], dt * 1e3, sh["exchange_us"] * 1e-3, sh["exchange_exposed_us"] * 1e-3, sh["exchange_hidden_frac"], sh["exchange_bytes"] / 1e6), flush=True)
comm.free()
