#!/usr/bin/env python3
"""experiment helper: SpGEMM stage times for the bench matrices.  usage: spgemm_stages.py [case-substring] [--quick]
cases: fem (fp32 + fp16), cage, dense (banded hb 32), banded (hb 8), rmat16, ceiling (banded hb 256: 77.9 M tasks; only when named)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bmsparse-spgemm-spmv_amd"))
import numpy as np, pybmsp as B
from pybmsp import gen
cases = [("fem_like(47,27pt)", lambda: gen.fem_like(47, "27pt")), ("cage_like(130228)", lambda: gen.cage_like(130228)),
         ("dense banded(131072,32)", lambda: gen.banded(131072, 32)), ("banded(101492,8)", lambda: gen.banded(101492, 8)),
         ("rmat16(16,8)", lambda: gen.rmat(16, 8)), ("ceiling banded(147456,256)", lambda: gen.banded(147456, 256)),
         ("rmat22(22,1)", lambda: gen.rmat(22, 1)), ("rmat20(20,2)", lambda: gen.rmat(20, 2))]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
quick = "--quick" in sys.argv
fp32 = "--fp32" in sys.argv  # with --quick: the fp32 V15 configuration instead of the fp16 MFMA one
half5 = "--half5" in sys.argv  # with --quick: fp16 operands with V15 numerics (tc_version 5: the reference's default configuration)
cases = [c for c in cases if args[0] in c[0]] if args else cases[:5]
for name, mk in cases:
    n, _, r, c, v = mk()
    for dtype, tc in ((((B.F32, 5),) if fp32 else ((B.F16, 5),) if half5 else ((B.F16, 4),)) if quick else ((B.F32, 5), (B.F16, 5), (B.F16, 4))):
        A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype).prepare(2)
        At = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=dtype).prepare(2)
        for mode in ((0,) if quick else (2, 1, 0)):
            best = None
            for it in range(4):
                Cm, st = B.spgemm(A, At, mode=mode, tc_version=tc)
                if it and (best is None or st["t_us"][0] < best["t_us"][0]): best = st
                del Cm
            t = best["t_us"]
            print("%-24s dt=%d tc=%d mode=%d path=%d tasks=%9d surv=%9d C=%8d | total %8.0f us | T2 %5.0f T3 %6.0f T4 %6.0f T5 %7.0f T6 %5.0f T9 %6.0f T7 %7.0f | MAC %.2f TF/s" % (
                name, dtype, tc, mode, best["sort_path"], best["task_list_size"], best["surviving_tasks"], best["c_blocks"], t[0], t[2], t[3], t[4], t[5], t[6], t[9], t[7],
                1024.0 * best["surviving_tasks"] / max(t[7], 1e-9) / 1e6), flush=True)
