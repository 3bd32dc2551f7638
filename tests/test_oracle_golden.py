"""CPU tests: the oracle against the reference's fixtures, golden vectors and independent ground truth."""
import json
import os
import numpy as np
import pytest
from conftest import GOLDEN, MTX
import util


def test_half_rounding_matches_reference_half_hpp(oracle):
    g = json.load(open(os.path.join(GOLDEN, "half_rounding.json")))
    L = oracle.lib()
    import struct
    for hexd, bits in g["f64_to_f16"]:
        d = struct.unpack(">d", bytes.fromhex(hexd))[0]
        assert "%04x" % L.orc_f64_to_f16_bits(d) == bits, (d, bits)
    for a, b, prod in g["f16_mul"]:
        fa, fb = L.orc_f16_bits_to_f64(int(a, 16)), L.orc_f16_bits_to_f64(int(b, 16))
        got = L.orc_f64_to_f16_bits(fa * fb)  # V15: product rounded to fp16 (product is exact in double)
        if int(prod, 16) & 0x7fff > 0x7c00:   # NaN payloads are not compared
            continue
        assert "%04x" % got == prod, (a, b, prod)
    # numpy's float16 follows the same IEEE rules: cross-check the decode direction
    for h in range(0, 0x7c00, 37):
        assert L.orc_f16_bits_to_f64(h) == float(np.array([h], dtype=np.uint16).view(np.float16)[0])


def test_ragusa16_known_answers(oracle):
    """SURVEY.md 8(c) / BASELINE.md 2, recomputed independently by tests/golden/make_golden.py."""
    k = json.load(open(os.path.join(GOLDEN, "ragusa16_known.json")))
    coo_a = oracle.mtx_read(os.path.join(MTX, "real", "A_matrix.mtx"))
    coo_b = oracle.mtx_read(os.path.join(MTX, "real", "B_matrix.mtx"))
    for dt in (oracle.F32, oracle.F16):
        A = oracle.bmsp_from_coo(coo_a, dt, False)
        assert A.block_num == 9 and A.nnz == 81
        assert ["%016x" % x for x in A.keys] == k["a_keys"]
        assert ["%016x" % x for x in A.bmps] == k["a_bmps"]
        assert [bin(int(x)).count("1") for x in A.bmps] == k["a_popcounts"] == [8, 12, 5, 12, 12, 8, 7, 12, 5]
        np.testing.assert_array_equal(oracle.spmv_f32(A, np.ones(24, np.float32)), np.array(k["y_ones"], np.float32))
        for name, other in (("AxB", coo_b), ("AxA", coo_a)):
            Bt = oracle.bmsp_from_coo(other, dt, True)
            for exact in (False, True):
                Cm, st = oracle.spgemm(A, Bt, exact_products=exact)
                ka = k[name]
                assert st["task_list_size"] == ka["candidate_tasks"] == 27
                assert st["surviving_tasks"] == ka["surviving_tasks"] == 27
                assert st["c_blocks"] == ka["c_blocks"] == 9 and st["c_nnz"] == ka["c_nnz"] == 255
                assert st["scalar_products"] == ka["scalar_products"] == 446
                assert ["%016x" % x for x in Cm.keys] == ka["c_keys"]
                assert ["%016x" % x for x in Cm.bmps] == ka["c_bmps"]
                assert Cm.values.sum() == ka["sum"] and Cm.values.max() == ka["max"] == 51
                d = util.bmsp_host_to_dok(24, 24, Cm.keys, Cm.bmps, Cm.offsets, Cm.values)
                assert sorted([i, j, v] for (i, j), v in d.items()) == ka["entries"]


@pytest.mark.parametrize("path", util.all_fixture_mtx())
def test_builder_roundtrip_and_spmv_on_fixtures(oracle, path):
    coo = oracle.mtx_read(path)
    truth = util.dok_from_coo(coo.rows, coo.cols, coo.vals)
    for transposed in (False, True):
        A = oracle.bmsp_from_coo(coo, oracle.F32, transposed)
        assert util.bmsp_host_to_dok(A.num_rows, A.num_cols, A.keys, A.bmps, A.offsets, A.values, transposed) == truth
        back = oracle.bmsp_to_coo(A)
        assert util.dok_from_coo(back.rows, back.cols, back.vals) == truth
        assert np.all(np.diff(A.keys.astype(np.int64)) > 0) if A.block_num > 1 else True
    A = oracle.bmsp_from_coo(coo, oracle.F32, False)
    x = ((np.arange(coo.num_cols) % 21) - 10).astype(np.float32)
    y = oracle.spmv_f32(A, x)
    ref = util.scipy_csr(coo.num_rows, coo.num_cols, coo.rows, coo.cols, coo.vals) @ x.astype(np.float64)
    np.testing.assert_allclose(y, ref, rtol=1e-6, atol=1e-4)
    assert oracle.bmsp_compare(A, coo) == 0.0


@pytest.mark.parametrize("path", util.all_fixture_mtx())
def test_spgemm_square_on_fixtures(oracle, path):
    """A*A on every square fixture against scipy (integer-valued fixtures: exact)."""
    coo = oracle.mtx_read(path)
    if coo.num_rows != coo.num_cols:
        pytest.skip("not square")
    A = oracle.bmsp_from_coo(coo, oracle.F32, False)
    At = oracle.bmsp_from_coo(coo, oracle.F32, True)
    Cm, st = oracle.spgemm(A, At)
    S = util.scipy_csr(coo.num_rows, coo.num_cols, coo.rows, coo.cols, coo.vals)
    ref = (S @ S).tocoo()
    got = util.bmsp_host_to_dok(coo.num_rows, coo.num_cols, Cm.keys, Cm.bmps, Cm.offsets, Cm.values)
    refd = {(int(r), int(c)): float(v) for r, c, v in zip(ref.row, ref.col, ref.data)}
    # bmSparse keeps symbolic entries (numeric zeros stay); scipy may drop or keep them
    for kk, v in refd.items():
        if v != 0.0:
            assert kk in got
    for kk, v in got.items():
        assert abs(v - refd.get(kk, 0.0)) <= 1e-5 * max(1.0, abs(v)), (kk, v, refd.get(kk))
    assert st["c_nnz"] == len(got) == int(sum(bin(int(b)).count("1") for b in Cm.bmps))


def test_bmp_product_matches_bruteforce(oracle):
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for _ in range(300):
        dens = rng.choice([0.02, 0.1, 0.5])
        a = sum(1 << i for i in range(64) if rng.random() < dens)
        b = sum(1 << i for i in range(64) if rng.random() < dens)
        exp = 0
        for i in range(8):
            for j in range(8):
                for kk in range(8):
                    if (a >> (63 - (i * 8 + kk))) & 1 and (b >> (63 - (j * 8 + kk))) & 1:
                        exp |= 1 << (63 - (i * 8 + j))
        assert L.orc_bmp_product(a, b) == exp
        assert bool(L.orc_bmp_product_empty(a, b)) == (exp == 0)


def test_fp16_semantics_v15_vs_exact(oracle):
    """rounded-product (V15) and exact-product (tensor) results both sit inside the stated fp16 tolerance."""
    from pybmsp import gen
    n, _, r, c, v = gen.random_coo(96, 96, 1500, seed=5, lo=0.0, hi=1.0)
    coo = oracle.Coo(n, n, r, c, v)
    A = oracle.bmsp_from_coo(coo, oracle.F16, False)
    At = oracle.bmsp_from_coo(coo, oracle.F16, True)
    C15, _ = oracle.spgemm(A, At, exact_products=False)
    Cex, _ = oracle.spgemm(A, At, exact_products=True)
    np.testing.assert_array_equal(C15.keys, Cex.keys)
    np.testing.assert_array_equal(C15.bmps, Cex.bmps)
    Ar = util.scipy_csr(n, n, *[getattr(oracle.bmsp_to_coo(A), k) for k in ("rows", "cols", "vals")])
    ref = (Ar @ Ar).todok()
    absref = (abs(Ar) @ abs(Ar)).todok()
    for Cm in (C15, Cex):
        for (i, j), val in util.bmsp_host_to_dok(n, n, Cm.keys, Cm.bmps, Cm.offsets, Cm.values).items():
            assert abs(val - ref[i, j]) <= 2.0 ** -10 * absref[i, j] + 1e-6


def test_segsort_gold_is_stable_sort(oracle):
    rng = np.random.default_rng(0)
    n = 5000
    keys = rng.integers(0, 50, n).astype(np.uint64)
    vals = np.stack([np.arange(n, dtype=np.uint64), rng.integers(0, 1 << 60, n).astype(np.uint64)], axis=1)
    segs = np.array([0, 0, 10, 11, 700, 700, 4000], dtype=np.int64)
    k2, v2 = oracle.segsort(keys, vals, segs)
    bounds = list(segs) + [n]
    for s in range(len(segs)):
        lo, hi = bounds[s], bounds[s + 1]
        order = np.argsort(keys[lo:hi], kind="stable")
        np.testing.assert_array_equal(k2[lo:hi], keys[lo:hi][order])
        np.testing.assert_array_equal(v2[lo:hi], vals[lo:hi][order])


@pytest.mark.parametrize("path", util.all_fixture_mtx(include_pattern=True))
def test_csr_baseline_vs_scipy(oracle, path):
    """the cusp::multiply restatement (CPU baseline) against scipy on the reference's fixtures."""
    if "complex" in path:
        pytest.skip("complex")
    coo = oracle.mtx_read(path, strict=True)
    A = oracle.csr_from_coo(coo)
    S = util.scipy_csr(coo.num_rows, coo.num_cols, coo.rows, coo.cols, coo.vals, np.float32)
    x = (np.arange(coo.num_cols) % 10).astype(np.float32)  # cusp/testing/multiply.cu:390
    for th in (1, 2):
        np.testing.assert_allclose(oracle.csr_spmv(A, x, th), S @ x, rtol=1e-6, atol=1e-5)
    if coo.num_rows == coo.num_cols:
        ref = (S @ S).toarray()
        for th in (1, 2):
            Cm, prods = oracle.csr_spgemm(A, A, th)
            dense = np.zeros((coo.num_rows, coo.num_cols), np.float32)
            for i in range(Cm.num_rows):
                for kk in range(Cm.row_offsets[i], Cm.row_offsets[i + 1]):
                    dense[i, Cm.cols[kk]] += Cm.vals[kk]
            np.testing.assert_allclose(dense, ref, rtol=1e-6, atol=1e-5)
            if th == 1:  # sequential path drops numeric zeros (csr_spgemm.h:135)
                assert np.all(Cm.vals != 0)


def test_pattern_symmetric_reader(oracle):
    coo = oracle.mtx_read(os.path.join(MTX, "test", "coordinate_pattern_symmetric.mtx"), strict=True)
    assert (coo.num_rows, coo.num_cols, coo.nnz) == (5, 5, 9)
    assert set(zip(coo.rows.tolist(), coo.cols.tolist())) == {(0, 0), (1, 1), (3, 1), (1, 3), (2, 2), (3, 3), (4, 4), (4, 3), (3, 4)}
    assert np.all(coo.vals == 1.0)
    with pytest.raises(IOError):
        oracle.mtx_read(os.path.join(MTX, "test", "coordinate_pattern_symmetric.mtx"), strict=False)  # reference: undefined
    with pytest.raises(IOError):
        oracle.mtx_read("/nonexistent.mtx")


def test_cusp_literal_spmv_known_answer(oracle):
    """cusp/testing/multiply.cu:441-461: literal 5x4 matrix, x[i]=i%10, known y."""
    dense = np.array([[13, 80, 0, 0], [0, 27, 0, 0], [55, 0, 24, 42], [0, 69, 0, 83], [0, 0, 27, 0]], dtype=np.float32)
    r, c = np.nonzero(dense)
    coo = oracle.Coo(5, 4, r, c, dense[r, c])
    A = oracle.csr_from_coo(coo)
    x = (np.arange(4) % 10).astype(np.float32)
    np.testing.assert_array_equal(oracle.csr_spmv(A, x, 1), dense @ x)
    M = oracle.bmsp_from_coo(coo, oracle.F32, False)
    np.testing.assert_array_equal(oracle.spmv_f32(M, x), dense @ x)


def _cusp_cases():
    g = json.load(open(os.path.join(GOLDEN, "cusp_multiply.json")))
    return g, [(p["left"], p["right"]) for p in g["products"]]


def _coo_of_dense(d):
    a = np.asarray(d, dtype=np.float64)
    r, c = np.nonzero(a)
    return a.shape[0], a.shape[1], r.astype(np.int32), c.astype(np.int32), a[r, c]


@pytest.mark.parametrize("pair", _cusp_cases()[1], ids=lambda p: "%sx%s" % p)
def test_cusp_multiply_known_answers(oracle, pair):
    """cusp/testing/multiply.cu:39-128 -- every compatible pair of A..K, dense product as the expected value (the reference
    test's own method), through BOTH CPU restatements: the CSR Gustavson baseline (csr_spgemm.h:39-157, seq + OpenMP) and the
    bmSparse pipeline (fp32 V15 order, fp16 rounded- and exact-product).  All values are small multiples of 0.5: exact."""
    g, _ = _cusp_cases()
    L, R = g["matrices"][pair[0]], g["matrices"][pair[1]]
    want = np.asarray([p for p in g["products"] if (p["left"], p["right"]) == pair][0]["dense"], dtype=np.float64)
    ar, ac, r1, c1, v1 = _coo_of_dense(L["dense"])
    br, bc, r2, c2, v2 = _coo_of_dense(R["dense"])
    # CSR baseline: compare dense images
    A, B = oracle.csr_from_coo(oracle.Coo(ar, ac, r1, c1, v1)), oracle.csr_from_coo(oracle.Coo(br, bc, r2, c2, v2))
    for th in (1, 4):
        Cm, _ = oracle.csr_spgemm(A, B, th)
        got = np.zeros((ar, bc))
        for i in range(ar):
            for q in range(Cm.row_offsets[i], Cm.row_offsets[i + 1]):
                got[i, Cm.cols[q]] += Cm.vals[q]
        np.testing.assert_array_equal(got, want)
        if th == 1:  # the sequential pass drops numeric zeros (csr_spgemm.h:135); the OpenMP pass keeps them (omp/.../csr_spgemm.h:139-152)
            assert not np.any(Cm.vals == 0.0)
    # bmSparse pipeline: symbolic nnz keeps cancelled entries as explicit zeros (SPGEMM.cu:1085-1107)
    for dt in (oracle.F32, oracle.F16):
        a = oracle.bmsp_from_coo(oracle.Coo(ar, ac, r1, c1, v1), dt, False)
        b = oracle.bmsp_from_coo(oracle.Coo(br, bc, r2, c2, v2), dt, True)
        for exact in (False, True):
            Cb, st = oracle.spgemm(a, b, exact_products=exact)
            d = util.bmsp_host_to_dok(ar, bc, Cb.keys, Cb.bmps, Cb.offsets, Cb.values)
            got = np.zeros((ar, bc))
            for (i, j), v in d.items():
                got[i, j] = v
            np.testing.assert_array_equal(got, want)
            assert set(zip(*np.nonzero(want))) <= set(d)
