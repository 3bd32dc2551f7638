"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar: bit-exact for every integer array (keys, bitmaps, offsets, nnz, block counts) and for values whenever the
arithmetic is exact or order-identical; fp32 values within 1e-5 relative (the reference authors' own SpMV check
was abs 1e-5); fp16 within 2^-10 * sum|a*b| + fp32 slack.
"""
import json
import os
import numpy as np
import pytest
from conftest import GOLDEN, MTX
import util

pytestmark = pytest.mark.gpu

NPDT = {0: np.float32, 1: np.float16, 2: np.float64}


def build_both(oracle, bmsp, nr, nc, rows, cols, vals, dtype, transposed):
    ref = oracle.bmsp_from_coo(oracle.Coo(nr, nc, rows, cols, vals), dtype, transposed)
    got = bmsp.BmSpMatrix.from_coo(nr, nc, rows, cols, vals, transposed=transposed, dtype=dtype)
    return ref, got


def check_builder(oracle, bmsp, nr, nc, rows, cols, vals, dtype, transposed):
    ref, got = build_both(oracle, bmsp, nr, nc, rows, cols, vals, dtype, transposed)
    info = got.info()
    assert (info["num_rows"], info["num_cols"], info["nnz"], info["block_num"]) == (nr, nc, ref.nnz, ref.block_num)
    k, b, o, v = got.host_arrays()
    util.assert_bmsp_equal_exact(ref, k, b, o, v, NPDT[dtype])
    # dense block-row pointer
    rp = got.block_row_ptr()
    nbr = (nr + 7) // 8
    exp = np.searchsorted((ref.keys >> np.uint64(32)).astype(np.int64), np.arange(nbr + 1), side="left")
    np.testing.assert_array_equal(rp, exp.astype(np.uint32))
    return ref, got


# ---------------------------------------------------------------------------------------------------------
# builder (B1/B2/B3/B4)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", util.all_fixture_mtx())
@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_builder_from_mtx_fixtures(oracle, bmsp, path, dtype):
    coo = oracle.mtx_read(path)
    for transposed in (False, True):
        ref = oracle.bmsp_from_coo(coo, dtype, transposed)
        got = bmsp.BmSpMatrix.from_mtx(path, transposed=transposed, dtype=dtype)
        k, b, o, v = got.host_arrays()
        util.assert_bmsp_equal_exact(ref, k, b, o, v, NPDT[dtype])
        # generate_coo round trip (honours the transposed layout)
        r, c, vals = got.to_coo()
        back = oracle.bmsp_to_coo(ref)
        np.testing.assert_array_equal(r, back.rows)
        np.testing.assert_array_equal(c, back.cols)
        np.testing.assert_array_equal(vals, back.vals)
        err, missing = got.compare(coo.rows, coo.cols, [oracle.lib().orc_round_to_dtype(x, dtype) for x in coo.vals])
        assert err == 0.0 and missing == 0


def test_builder_pattern_symmetric_and_suffix(oracle, bmsp):
    path = os.path.join(MTX, "test", "coordinate_pattern_symmetric.mtx")
    coo = oracle.mtx_read(path, strict=True)  # CUSP reader semantics: pattern -> 1, mirrored
    got = bmsp.BmSpMatrix.from_mtx(path)
    ref = oracle.bmsp_from_coo(coo, 0, False)
    util.assert_bmsp_equal_exact(ref, *got.host_arrays(), np.float32)
    got2 = bmsp.BmSpMatrix.from_mtx(os.path.join(MTX, "real", "A_matrix"))  # without ".mtx"
    assert got2.block_num == 9
    with pytest.raises(bmsp.BmspError):
        bmsp.BmSpMatrix.from_mtx("/nonexistent")


@pytest.mark.parametrize("shape", [(1, 1, 1), (8, 8, 64), (9, 17, 40), (100, 37, 300), (1000, 1000, 0), (5, 3000, 2000),
                                   (4096, 4096, 50000), (333, 777, 9000)])
@pytest.mark.parametrize("dtype", [0, 1])
def test_builder_random_ragged_and_duplicates(oracle, bmsp, shape, dtype):
    from pybmsp import gen
    nr, nc, nnz = shape
    _, _, r, c, v = gen.random_coo(nr, nc, nnz, seed=nnz + 3)
    check_builder(oracle, bmsp, nr, nc, r, c, v, dtype, False)
    check_builder(oracle, bmsp, nr, nc, r, c, v, dtype, True)
    if nnz:
        # unsorted input with duplicate coordinates: summed in input order in the matrix's precision
        rng = np.random.default_rng(nnz)
        idx = rng.integers(0, r.size, size=r.size * 2)
        rr, cc, vv = r[idx], c[idx], rng.uniform(-1, 1, idx.size)
        check_builder(oracle, bmsp, nr, nc, rr, cc, vv, dtype, False)


def test_builder_fp16_rounding_and_overflow(oracle, bmsp):
    g = json.load(open(os.path.join(GOLDEN, "half_rounding.json")))
    import struct
    ds = [struct.unpack(">d", bytes.fromhex(h))[0] for h, _ in g["f64_to_f16"]]
    ds = [d for d in ds if d == d]
    n = len(ds)
    rows = np.arange(n) // 30
    cols = np.arange(n) % 30
    got = bmsp.BmSpMatrix.from_coo(rows.max() + 1, 30, rows, cols, ds, dtype=1)
    ref = oracle.bmsp_from_coo(oracle.Coo(rows.max() + 1, 30, rows, cols, ds), 1, False)
    util.assert_bmsp_equal_exact(ref, *got.host_arrays(), np.float16)  # includes 65520 -> inf, subnormals, ties


def test_adopt_arrays_constructor(oracle, bmsp):
    from pybmsp import gen
    _, _, r, c, v = gen.random_coo(200, 150, 3000, seed=9)
    ref = oracle.bmsp_from_coo(oracle.Coo(200, 150, r, c, v), 0, False)
    m = bmsp.BmSpMatrix.from_arrays(200, 150, ref.keys, ref.bmps, ref.offsets[:-1], ref.values.astype(np.float32))
    util.assert_bmsp_equal_exact(ref, *m.host_arrays(), np.float32)
    x = np.ones(150, np.float32)
    u = bmsp.spmv(m, bmsp.DeviceArray.from_host(x)).to_host()
    np.testing.assert_allclose(u, oracle.spmv_f32(ref, x), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------
# SpMV (M1/M2/M3)
# ---------------------------------------------------------------------------------------------------------
def check_spmv(oracle, bmsp, nr, nc, r, c, v, xkind="cusp"):
    from pybmsp import gen
    ref, got = build_both(oracle, bmsp, nr, nc, r, c, v, 0, False)
    x = gen.spmv_x(nc, "ones" if xkind == "ones" else "cusp")
    y_ref = oracle.spmv_f32(ref, x)
    S = util.scipy_csr(nr, nc, r, c, np.asarray(v, np.float32).astype(np.float64))
    y64 = S @ x.astype(np.float64)
    bound = 1e-5 * (abs(S) @ np.abs(x.astype(np.float64))) + 1e-30
    dx = bmsp.DeviceArray.from_host(x)
    first = None
    for variant in (0, 0, 1, 2):  # sweep (twice: arrival counters must reset, result bitwise reproducible), batched, row-group
        du = bmsp.DeviceArray(nr, np.float32)
        assert bmsp.lib().bmsp_memset(du.ptr, 0xFF, nr * 4) == 0  # NaN poison: every row must be written
        bmsp.check(bmsp.lib().bmsp_spmv(got.h, dx.ptr, du.ptr, variant, None))
        y = du.to_host()
        assert np.all(np.isfinite(y))
        assert np.all(np.abs(y - y_ref) <= bound + 1e-5 * np.abs(y_ref)), np.max(np.abs(y - y_ref))
        assert np.all(np.abs(y - y64) <= 2 * bound + 1e-5 * np.abs(y64))
        if variant == 0:
            if first is None:
                first = y
            else:
                np.testing.assert_array_equal(first.view(np.uint32), y.view(np.uint32))
    return y_ref


def test_spmv_ragusa_known_answer(oracle, bmsp):
    k = json.load(open(os.path.join(GOLDEN, "ragusa16_known.json")))
    A = bmsp.BmSpMatrix.from_mtx(os.path.join(MTX, "real", "A_matrix.mtx"))
    for batched in (False, True):
        u = bmsp.spmv(A, bmsp.DeviceArray.from_host(np.ones(24, np.float32)), batched=batched).to_host()
        np.testing.assert_array_equal(u, np.array(k["y_ones"], np.float32))


@pytest.mark.parametrize("path", util.all_fixture_mtx())
def test_spmv_fixtures(oracle, bmsp, path):
    coo = oracle.mtx_read(path)
    check_spmv(oracle, bmsp, coo.num_rows, coo.num_cols, coo.rows, coo.cols, coo.vals)  # includes empty rows / ragged edges


@pytest.mark.parametrize("case", ["banded", "rmat", "ragged", "empty_rows", "wide", "hub", "one_block_rows", "gap", "full_tiles", "full_tiles_odd"])
def test_spmv_synthetic(oracle, bmsp, case):
    from pybmsp import gen
    if case == "banded":
        nr, nc, r, c, v = gen.banded(5000, 8)
    elif case == "rmat":
        nr, nc, r, c, v = gen.rmat(12, 8)
    elif case == "ragged":
        nr, nc, r, c, v = gen.random_coo(1003, 517, 20000, seed=4)
    elif case == "empty_rows":
        nr, nc, r, c, v = gen.random_coo(4000, 4000, 300, seed=5)  # most block-rows empty (reference bug)
    elif case == "wide":
        nr, nc, r, c, v = gen.random_coo(16, 50000, 60000, seed=6)  # two block-rows with thousands of tiles (long-row items)
    elif case == "hub":
        # short rows, then a hub block-row of ~3000 tiles in the middle of a 64-row window, then short rows again
        _, _, r1, c1, v1 = gen.random_coo(2000, 30000, 6000, seed=7)
        _, _, r2, c2, v2 = gen.random_coo(8, 30000, 9000, seed=8)
        nr, nc = 2000, 30000
        r, c, v = np.concatenate([r1, r2 + 1000]), np.concatenate([c1, c2]), np.concatenate([v1, v2])
        _, idx = np.unique(r.astype(np.int64) * nc + c, return_index=True)
        r, c, v = r[idx], c[idx], v[idx]
    elif case == "full_tiles":
        nr, nc, r, c, v = gen.banded(3001, 24)  # 5 of 7 tiles per block-row hold all 64 values: the 16-byte full-tile pass
    elif case == "full_tiles_odd":
        # full tiles behind tiles with odd value counts: fp16 values of a full tile start 2-byte aligned (takes the general path)
        _, _, r1, c1, v1 = gen.banded(1601, 24)
        r2 = np.arange(0, 1601, 3); c2 = (r2 + 200) % 1601; v2 = np.linspace(0.25, 1.0, r2.size)
        nr = nc = 1601
        r, c, v = np.concatenate([r1, r2]), np.concatenate([c1, c2]), np.concatenate([v1, v2])
        _, idx = np.unique(r.astype(np.int64) * nc + c, return_index=True)
        r, c, v = r[idx], c[idx], v[idx]
    elif case == "one_block_rows":
        n = 3000  # one tile per block-row: items are bounded by the 64-row window, not by the tile budget
        r = np.arange(n); c = (np.arange(n) * 7) % n; v = np.linspace(-1, 1, n)
        nr, nc = n, n
    else:
        # blocks only at the very beginning and the very end: huge runs of empty block-rows between items
        nr, nc = 100000, 100000
        r = np.array([0, 3, 9, 99990, 99999]); c = np.array([5, 99999, 0, 17, 99999]); v = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    check_spmv(oracle, bmsp, nr, nc, r, c, v)
    check_spmv(oracle, bmsp, nr, nc, r, c, v, "ones")


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("case", ["rmat", "full_tiles", "full_tiles_odd", "hub"])
def test_spmv_half_and_double(bmsp, dtype, case):
    """bmSpMatrix<half> / <double> sweeps (instantiated at src/bmSpMatrix.cu:435-437): fp16 values and x with fp32 accumulation,
    fp64 throughout; against scipy in float64 on the rounded inputs."""
    import scipy.sparse as sp
    from pybmsp import gen
    if case == "rmat":
        n, _, r, c, v = gen.rmat(11, 8)
    elif case == "full_tiles":
        n, _, r, c, v = gen.banded(3001, 24)
    elif case == "full_tiles_odd":
        _, _, r1, c1, v1 = gen.banded(1601, 24)
        r2 = np.arange(0, 1601, 3); c2 = (r2 + 200) % 1601; v2 = np.linspace(0.25, 1.0, r2.size)
        n = 1601
        r, c, v = np.concatenate([r1, r2]), np.concatenate([c1, c2]), np.concatenate([v1, v2])
        _, idx = np.unique(r.astype(np.int64) * n + c, return_index=True)
        r, c, v = r[idx], c[idx], v[idx]
    else:
        n, _, r, c, v = gen.random_coo(16, 40000, 50000, seed=6)
        n = 40000
    np_in = bmsp.NP_DTYPE[dtype]
    vq = np.asarray(v, np.float64).astype(np_in)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
    x = gen.spmv_x(n, "cusp").astype(np_in)
    S = sp.coo_matrix((vq.astype(np.float64), (r, c)), shape=(n, n)).tocsr()
    want = S @ x.astype(np.float64)
    mag = abs(S) @ np.abs(x.astype(np.float64))
    tol = 1e-5 if dtype == 1 else 1e-13  # fp32 accumulation of exact fp16 products / fp64
    for variant in (0, 1, 3):
        y = bmsp.DeviceArray(n, bmsp.OUT_DTYPE[dtype])
        bmsp.check(bmsp.lib().bmsp_spmv(A.h, bmsp.DeviceArray.from_host(x).ptr, y.ptr, variant, None))
        assert np.all(np.abs(y.to_host().astype(np.float64) - want) <= tol * mag + 1e-30)


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("case", ["banded_ragged", "partial_nibbles", "empty_rows_wide", "long_row", "fem"])
def test_spmv_rowgroup_kernel(bmsp, dtype, case):
    """variant 3 (16 lanes per block-row, 16-byte value loads; the default for matrices with dense tiles): ragged last block row
    AND column (num_cols % 4 != 0: the 16-byte x load must not be used across the end of x), partly filled nibbles at every
    alignment (fp16 values start at odd halves), empty block-rows (written as 0 over a NaN-poisoned y), a block-row longer than
    one trip; against scipy float64 on the rounded inputs; two sweeps bit-identical (register sums, no atomics)."""
    import scipy.sparse as sp
    from pybmsp import gen
    if case == "banded_ragged":
        nr = nc = 1003
        _, _, r, c, v = gen.banded(1003, 21)
    elif case == "partial_nibbles":
        nr, nc = 777, 1234
        _, _, r, c, v = gen.random_coo(nr, nc, 120000, seed=11)  # ~12 values per tile, every nibble pattern
    elif case == "empty_rows_wide":
        nr, nc = 4001, 515
        _, _, r, c, v = gen.random_coo(nr, nc, 60000, seed=12)
        keep = (r // 8) % 3 != 1  # every third block-row empty
        r, c, v = r[keep], c[keep], v[keep]
    elif case == "long_row":
        nr, nc = 40, 30011
        _, _, r, c, v = gen.random_coo(nr, nc, 400000, seed=13)
    else:
        nr, _, r, c, v = gen.fem_like(12, "27pt", window=8)
        nc = nr
    np_in = bmsp.NP_DTYPE[dtype]
    vq = np.asarray(v, np.float64).astype(np_in)
    A = bmsp.BmSpMatrix.from_coo(nr, nc, r, c, v, dtype=dtype)
    x = gen.spmv_x(nc, "cusp").astype(np_in)
    S = sp.coo_matrix((vq.astype(np.float64), (r, c)), shape=(nr, nc)).tocsr()
    want = S @ x.astype(np.float64)
    mag = abs(S) @ np.abs(x.astype(np.float64))
    tol = {0: 1e-5, 1: 1e-5, 2: 1e-13}[dtype]
    dx = bmsp.DeviceArray.from_host(x)
    got = []
    for rep in range(2):
        y = bmsp.DeviceArray(nr, bmsp.OUT_DTYPE[dtype])
        assert bmsp.lib().bmsp_memset(y.ptr, 0xFF, nr * y.dtype.itemsize) == 0
        bmsp.check(bmsp.lib().bmsp_spmv(A.h, dx.ptr, y.ptr, 3, None))
        got.append(y.to_host())
        assert np.all(np.abs(got[-1].astype(np.float64) - want) <= tol * mag + 1e-30)
    np.testing.assert_array_equal(got[0].view(np.uint8), got[1].view(np.uint8))


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("mode", ["cached+auto", "cached+atomic", "cached+sorted", "decode+atomic", "decode+sorted"])
@pytest.mark.parametrize("case", ["rmat_hubs", "fem", "mid_density", "long_row", "empty_rows_ragged", "panel_view"])
def test_spmv_value_stream_modes(bmsp, monkeypatch, dtype, mode, case):
    """the value-stream kernel (lane per stored value; default for matrices of sparse tiles) in every entry source x reduction:
    entries from the per-matrix position cache or decoded from the bitmaps in LDS; LDS float atomics or counting sort + run sums.
    Cases: hub rows cut into 256-tile items with carry slots, items of several batches (> 128 tiles / > 512 values per batch), a
    block-row longer than one item, empty block-rows over a NaN-poisoned y with ragged last row / column, and a row-panel view
    (absolute offsets into the parent's values: the cache is indexed relative to the panel's first value).  Against scipy float64 on
    the rounded inputs; the sorted reduction must be bit-reproducible across sweeps."""
    import scipy.sparse as sp
    from pybmsp import gen
    src, red = mode.split("+")
    if src == "decode":
        monkeypatch.setenv("BMSP_SPMV_NO_POSCACHE", "1")
    if red != "auto":
        monkeypatch.setenv("BMSP_SPMV_RED", "0" if red == "atomic" else "1")
    lo = hi = None
    if case == "rmat_hubs":
        nr, _, r, c, v = gen.rmat(13, 6.0)
        nc = nr
    elif case == "fem":
        nr, _, r, c, v = gen.fem_like(14, "27pt", window=8)
        nc = nr
    elif case == "mid_density":
        nr, nc = 1501, 2003
        _, _, r, c, v = gen.random_coo(nr, nc, 260000, seed=21)  # ~6 values per tile: batches hit the 512-value cut
    elif case == "long_row":
        nr, nc = 24, 60011
        _, _, r, c, v = gen.random_coo(nr, nc, 90000, seed=22)   # ~7500 tiles per block-row: long-row items + fold
    elif case == "empty_rows_ragged":
        nr, nc = 4003, 517
        _, _, r, c, v = gen.random_coo(nr, nc, 9000, seed=23)
        keep = (r // 8) % 3 != 1
        r, c, v = r[keep], c[keep], v[keep]
    else:
        nr, _, r, c, v = gen.rmat(12, 4.0)
        nc = nr
        lo, hi = 100, 390  # block-rows of the panel
    np_in = bmsp.NP_DTYPE[dtype]
    vq = np.asarray(v, np.float64).astype(np_in)
    A = bmsp.BmSpMatrix.from_coo(nr, nc, r, c, v, dtype=dtype)
    info = A.info()
    assert info["nnz"] < 16 * info["block_num"]  # sparse tiles: the default variant takes the value-stream kernel
    x = gen.spmv_x(nc, "cusp").astype(np_in)
    S = sp.coo_matrix((vq.astype(np.float64), (r, c)), shape=(nr, nc)).tocsr()
    M, sel = A, slice(0, nr)
    if lo is not None:
        M, sel = A.row_panel(lo, hi), slice(lo * 8, hi * 8)  # a view keeps the parent's row numbering; only its rows are written
    want = (S @ x.astype(np.float64))[sel]
    mag = (abs(S) @ np.abs(x.astype(np.float64)))[sel]
    tol = {0: 1e-5, 1: 1e-5, 2: 1e-13}[dtype]
    dx = bmsp.DeviceArray.from_host(x)
    for rep in range(2):
        y = bmsp.DeviceArray(nr, bmsp.OUT_DTYPE[dtype])
        assert bmsp.lib().bmsp_memset(y.ptr, 0xFF, nr * y.dtype.itemsize) == 0
        bmsp.check(bmsp.lib().bmsp_spmv(M.h, dx.ptr, y.ptr, 0, None))
        got = y.to_host()[sel].astype(np.float64)
        assert np.all(np.abs(got - want) <= tol * mag + 1e-30)


def test_spmv_bench_size_properties(bmsp):
    """the SpMV bench workload at its full size (R-MAT 2^20 x 2 + I, the webbase-1M stand-in): A*1 equals the host row sums, two
    sweeps are bit-identical (arrival counters reset, fixed fold order of hub rows), the block-row variants agree within tolerance."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(20, 2.0, seed=1)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    v32 = np.asarray(v, np.float32).astype(np.float64)
    rowsum = np.bincount(r, weights=v32, minlength=n)
    absrow = np.bincount(r, weights=np.abs(v32), minlength=n)
    ones = bmsp.DeviceArray.from_host(np.ones(n, np.float32))
    first = None
    for variant in (0, 0, 1):
        du = bmsp.DeviceArray(n, np.float32)
        assert bmsp.lib().bmsp_memset(du.ptr, 0xFF, n * 4) == 0
        bmsp.check(bmsp.lib().bmsp_spmv(A.h, ones.ptr, du.ptr, variant, None))
        y = du.to_host()
        assert np.all(np.abs(y - rowsum) <= 1e-5 * absrow + 1e-6)
        if variant == 0:
            if first is None:
                first = y
            else:
                np.testing.assert_array_equal(first.view(np.uint32), y.view(np.uint32))


def test_spmv_linearity_large(bmsp):
    """size-independent property at a BASELINE-scale input (webbase-1M-like R-MAT): A(x+y) = Ax + Ay, and
    A*1 equals the row sums computed on the host."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(18, 4)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    rowsum = np.bincount(r, weights=np.asarray(v, np.float32).astype(np.float64), minlength=n)
    ones = bmsp.DeviceArray.from_host(np.ones(n, np.float32))
    for variant in (0, 1):
        du = bmsp.DeviceArray(n, np.float32)
        bmsp.check(bmsp.lib().bmsp_spmv(A.h, ones.ptr, du.ptr, variant, None))
        np.testing.assert_allclose(du.to_host(), rowsum, rtol=1e-5, atol=1e-6)
    x = gen.spmv_x(n, "cusp")
    y = ((np.arange(n) % 7) - 3).astype(np.float32)
    ax = bmsp.spmv(A, bmsp.DeviceArray.from_host(x)).to_host().astype(np.float64)
    ay = bmsp.spmv(A, bmsp.DeviceArray.from_host(y)).to_host().astype(np.float64)
    axy = bmsp.spmv(A, bmsp.DeviceArray.from_host(x + y)).to_host().astype(np.float64)
    absrow = np.bincount(r, weights=np.abs(v) * 20, minlength=n)
    assert np.all(np.abs(axy - (ax + ay)) <= 1e-5 * absrow + 1e-6)


# ---------------------------------------------------------------------------------------------------------
# SpGEMM (G1..G9) and the segmented sort (S1)
# ---------------------------------------------------------------------------------------------------------
def check_spgemm(oracle, bmsp, A_coo, B_coo, dtype, mode, tc, exact_expected=False):
    (ar, ac, r1, c1, v1), (br, bc, r2, c2, v2) = A_coo, B_coo
    refA = oracle.bmsp_from_coo(oracle.Coo(ar, ac, r1, c1, v1), dtype, False)
    refB = oracle.bmsp_from_coo(oracle.Coo(br, bc, r2, c2, v2), dtype, True)
    A = bmsp.BmSpMatrix.from_coo(ar, ac, r1, c1, v1, transposed=False, dtype=dtype)
    B = bmsp.BmSpMatrix.from_coo(br, bc, r2, c2, v2, transposed=True, dtype=dtype)
    mfma = tc != 5 and dtype == 1
    refC, rst = oracle.spgemm(refA, refB, exact_products=mfma)
    Cm, st = bmsp.spgemm(A, B, mode=mode, tc_version=tc)
    # stage counters and structure: bit-exact
    assert st["task_list_size"] == rst["task_list_size"]
    assert st["bmp_reduction"] == rst["bmp_reduction"]
    assert st["surviving_tasks"] == rst["surviving_tasks"]
    assert st["c_blocks"] == rst["c_blocks"] and st["c_nnz"] == rst["c_nnz"]
    info = Cm.info()
    assert (info["num_rows"], info["num_cols"], info["block_num"], info["nnz"], info["transposed"]) == (ar, bc, refC.block_num, refC.nnz, 0)
    k, b, o, v = Cm.host_arrays()
    np.testing.assert_array_equal(k, refC.keys)
    np.testing.assert_array_equal(b, refC.bmps)
    np.testing.assert_array_equal(o, refC.offsets)
    ref_vals = refC.values
    if exact_expected or (not mfma):
        # same operation order as the oracle (tasks k-ascending, k = 0..7 inside a task): bit-exact
        np.testing.assert_array_equal(v.astype(np.float64), ref_vals)
    else:
        # matrix-core path: exact products, hardware accumulation order -> stated fp16 tolerance
        Aabs = util.scipy_csr(ar, ac, *[getattr(oracle.bmsp_to_coo(refA), n) for n in ("rows", "cols", "vals")])
        Bn = oracle.bmsp_to_coo(refB)
        Babs = util.scipy_csr(br, bc, Bn.rows, Bn.cols, Bn.vals)
        ref64 = (Aabs @ Babs).todok()
        mag = (abs(Aabs) @ abs(Babs)).todok()
        got = util.bmsp_host_to_dok(ar, bc, k, b, o, v)
        for (i, j), val in got.items():
            assert abs(val - ref64.get((i, j), 0.0)) <= 2.0 ** -10 * mag.get((i, j), 0.0) + 1e-6, (i, j, val)
        np.testing.assert_allclose(v.astype(np.float64), ref_vals, rtol=2e-3, atol=1e-6)
    return st


def test_spgemm_ragusa_known_answers(oracle, bmsp):
    k = json.load(open(os.path.join(GOLDEN, "ragusa16_known.json")))
    pa, pb = os.path.join(MTX, "real", "A_matrix.mtx"), os.path.join(MTX, "real", "B_matrix.mtx")
    for dtype in (0, 1):
        A = bmsp.BmSpMatrix.from_mtx(pa, False, dtype)
        for name, p in (("AxB", pb), ("AxA", pa)):
            B = bmsp.BmSpMatrix.from_mtx(p, True, dtype)
            for mode in (0, 1, 2):
                for tc in (5, 4, 1):
                    Cm, st = bmsp.spgemm(A, B, mode=mode, tc_version=tc)
                    ka = k[name]
                    assert (st["task_list_size"], st["surviving_tasks"], st["c_blocks"], st["c_nnz"]) == (27, 27, 9, 255)
                    kk, bb, oo, vv = Cm.host_arrays()
                    assert ["%016x" % x for x in kk] == ka["c_keys"] and ["%016x" % x for x in bb] == ka["c_bmps"]
                    assert float(vv.sum()) == ka["sum"] and float(vv.max()) == 51.0  # integer-exact in fp32 AND fp16
                    d = util.bmsp_host_to_dok(24, 24, kk, bb, oo, vv)
                    assert sorted([i, j, val] for (i, j), val in d.items()) == ka["entries"]


def test_mfma_16x16x32_lane_layout(bmsp):
    """the operand / result lane maps of v_mfma_f32_16x16x32_f16 that blockmac32.hip packs tiles by, checked on the hardware with
    asymmetric integer operands (a transposed or permuted map cannot pass)."""
    import ctypes
    bad = ctypes.c_int(-1)
    bmsp.check(bmsp.lib().bmsp_selftest_mfma_layout(ctypes.byref(bad)))
    assert bad.value == 0


def test_fp32_mfma_cancellation_corner(oracle, bmsp):
    """the fp32 matrix-core kernels are V15's fmaf chain while every product is a normal number -- but normal products of
    alternating sign can still CANCEL into subnormal sums.  The hardware self test runs exactly that (products ~2^-123, sums ~2^-140 and
    below, subnormal starting values); the library derives its exponent floor from the outcome (128, or 174 where the pipe differs), and a
    product built to cancel that way must equal the oracle bit for bit whichever kernel the floor sends it to."""
    import ctypes
    from pybmsp import gen
    bad, floor = ctypes.c_int(-1), ctypes.c_int(-1)
    bmsp.check(bmsp.lib().bmsp_selftest_mfma_f32_cancel(ctypes.byref(bad), ctypes.byref(floor)))
    assert bad.value >= 0 and floor.value == (128 if bad.value == 0 else 174), (bad.value, floor.value)
    n, _, r, c, v = gen.fem_like(10, "27pt")
    sign = np.where((r + c) % 2 == 0, 1.0, -1.0)
    va = (sign * (1.0 + (np.arange(r.size) % 7) * 2.0 ** -21) * 2.0 ** -60).astype(np.float32).astype(np.float64)
    vb = ((1.0 + (np.arange(r.size) % 5) * 2.0 ** -22) * 2.0 ** -63).astype(np.float32).astype(np.float64)
    st = check_spgemm(oracle, bmsp, (n, n, r, c, va), (n, n, r, c, vb), 0, 0, 5, exact_expected=True)
    assert st["sort_path"] == 2, st
    assert (st["mac_variant"] in (3, 5)) == (floor.value == 128), (st, floor.value)   # exponents sum to ~2 * 127 - 123 = 131: above 128, below 174
    # (3: the matrix-core strip kernel, 5: the row-sparse kernel -- V15's chain over the stored products only; both need normal products)


def test_tile_product_byte_permute_form(bmsp):
    """the v_perm_b32 form of the 8x8 boolean tile product (bmp_calculator) used by the column-window passes, against the multiply form,
    on the hardware: the selector semantics (8..11 = sign of the odd bytes) are an ISA detail worth pinning."""
    import ctypes
    bad = ctypes.c_int(-1)
    bmsp.check(bmsp.lib().bmsp_selftest_tile_product(ctypes.byref(bad)))
    assert bad.value == 0


def test_segmented_task_sort_wide_words(oracle, bmsp, monkeypatch):
    """the register segment sort with 64-bit sort words (products whose C has more than 2^20 block columns use them; forced here):
    segments of every size class (2 .. 64 words per lane would need > 2048 tasks per block-row: the rmat case reaches the 16 / 32
    classes, the hub case the LDS block kernel), structure and values against the oracle."""
    from pybmsp import gen
    monkeypatch.setenv("BMSP_SEGSORT_WIDE", "1")
    n, _, r, c, v = gen.rmat(11, 8)
    st = check_spgemm(oracle, bmsp, (n, n, r, c, v), (n, n, r, c, v), 0, 1, 5)
    n, _, r, c, v = gen.banded(2000, 40)
    check_spgemm(oracle, bmsp, (n, n, r, c, v), (n, n, r, c, v), 0, 1, 5)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("staging", ["0", "1"])
@pytest.mark.parametrize("case", ["rmat", "banded_full", "hub_c_blocks", "rect_ragged"])
def test_valu_block_mac_staging(oracle, bmsp, monkeypatch, case, staging, dtype):
    """tc_version 5 (V15 numerics: k order inside a task, tasks in list order) with the tiles staged from the dense copies (16-byte
    lines, four fp16 / two fp32 tasks per load) and with the element gathers: both bit-exact against the oracle, fp32 and fp16."""
    from pybmsp import gen
    monkeypatch.setenv("BMSP_MAC_VALU_DENSE", staging)
    if case == "rmat":
        n, _, r, c, v = gen.rmat(11, 8)
        A = Bc = (n, n, r, c, v)
    elif case == "banded_full":
        n, _, r, c, v = gen.banded(515, 20)
        A = Bc = (n, n, r, c, np.round(v * 8) / 8)
    elif case == "hub_c_blocks":
        nk = 8 * 300  # C blocks with hundreds of tasks: the 64-task windows of a hub group are refetched in place
        r = np.repeat(np.arange(16), nk); c = np.tile(np.arange(nk), 16)
        A = (16, nk, r, c, ((r * 7 + c) % 5 - 2).astype(np.float64))
        Bc = (nk, 16, c, r, ((c * 3 + r) % 7 - 3).astype(np.float64))
    else:
        _, _, r1, c1, v1 = gen.random_coo(203, 77, 2500, seed=5, integer=True)
        _, _, r2, c2, v2 = gen.random_coo(77, 331, 3000, seed=6, integer=True)
        A, Bc = (203, 77, r1, c1, v1), (77, 331, r2, c2, v2)
    check_spgemm(oracle, bmsp, A, Bc, dtype, 0, 5)


@pytest.mark.parametrize("b_dense", ["0", "1", "direct"])
@pytest.mark.parametrize("quota", ["64", ""])
@pytest.mark.parametrize("case", ["rmat", "banded_full", "hub_c_blocks", "rect_ragged", "ragusa"])
def test_mfma32_block_mac_paths(oracle, bmsp, monkeypatch, case, b_dense, quota):
    """the K = 32 block-MAC (tc_version 4) with B taken from its dense copy and from the compact values, and the direct kernel
    (operand lines straight into the MFMA lanes, pairs of C tiles, no LDS; forced here on every task-count profile), with the
    default wave quota and with the smallest one (64 tasks per wave: every window boundary, hub slices crossing quota
    boundaries, empty wave ranges), against the oracle's exact-product numerics; the r1 kernel (BMSP_MAC_OLD) must agree on
    the same inputs."""
    from pybmsp import gen
    if b_dense == "direct":
        monkeypatch.setenv("BMSP_MAC_B_DENSE", "1")
        monkeypatch.setenv("BMSP_MAC_DIRECT", "1")
    else:
        monkeypatch.setenv("BMSP_MAC_B_DENSE", b_dense)
        monkeypatch.setenv("BMSP_MAC_DIRECT", "0")
    if quota:
        monkeypatch.setenv("BMSP_MAC_QUOTA", quota)
    exact = False
    if case == "rmat":
        n, _, r, c, v = gen.rmat(11, 8)
        A = Bc = (n, n, r, c, v)
    elif case == "banded_full":
        n, _, r, c, v = gen.banded(515, 20)
        A = Bc = (n, n, r, c, np.round(v * 8) / 8)
    elif case == "hub_c_blocks":
        # one dense block-row times one dense block-column: C blocks with hundreds of tasks (W-task slices, carried accumulator)
        nk = 8 * 300
        r = np.repeat(np.arange(16), nk); c = np.tile(np.arange(nk), 16)
        A = (16, nk, r, c, ((r * 7 + c) % 5 - 2).astype(np.float64))
        Bc = (nk, 16, c, r, ((c * 3 + r) % 7 - 3).astype(np.float64))
        exact = True
    elif case == "rect_ragged":
        _, _, r1, c1, v1 = gen.random_coo(203, 77, 2500, seed=5, integer=True)
        _, _, r2, c2, v2 = gen.random_coo(77, 331, 3000, seed=6, integer=True)
        A, Bc = (203, 77, r1, c1, v1), (77, 331, r2, c2, v2)
        exact = True
    else:
        coo = oracle.mtx_read(os.path.join(MTX, "real", "A_matrix.mtx"))
        A = Bc = (24, 24, coo.rows, coo.cols, coo.vals)
        exact = True
    st = check_spgemm(oracle, bmsp, A, Bc, 1, 0, 4, exact_expected=exact)
    assert st["mac_kernel"] == 4
    # old and new kernels on the same operands
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=1)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=1)
    new, _ = bmsp.spgemm(a, b, tc_version=4)
    monkeypatch.setenv("BMSP_MAC_OLD", "1")
    old, _ = bmsp.spgemm(a, b, tc_version=4)
    monkeypatch.delenv("BMSP_MAC_OLD")
    vo, vn = old.host_arrays()[3], new.host_arrays()[3]
    if exact:
        np.testing.assert_array_equal(vo, vn)
    else:
        np.testing.assert_allclose(vn, vo, rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("long_sort", ["merge", "radix"])
@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("case", ["two_pieces", "three_pieces", "rmat_hubs", "wide_cols", "huge_cols"])
def test_spgemm_long_segments_merge(oracle, bmsp, monkeypatch, case, wide, long_sort):
    """T_5, segmented path, block-rows whose task segment exceeds what a wave sorts in registers (4096 words, 2048 with 64-bit sort
    words).  `merge`: the segment is cut into pieces, the pieces are sorted like ordinary segments and merged by merge-path passes (one
    pass + the copy back for two pieces, two passes for three or four, a hub-row mix on R-MAT).  `radix` (round 4, what long hub
    segments take by default): stable counting passes on 7 column bits at a time over tiles of 4096 -- one pass for the narrow B's, two
    on R-MAT, three for `wide_cols` (36 000 block columns), with a ragged last tile, two long segments side by side and short segments
    around them written into the other array when the pass count is odd.  The fp32 V15 values are compared BIT FOR BIT: they depend
    on the order of the tasks inside every C tile, i.e. on the sort being stable across piece / tile boundaries."""
    from pybmsp import gen
    if wide:
        monkeypatch.setenv("BMSP_SEGSORT_WIDE", "1")
    monkeypatch.setenv("BMSP_SEGSORT_RADIX", "1" if long_sort == "radix" else "0")
    if case == "rmat_hubs":
        n, _, r, c, v = gen.rmat(12, 8)
        A = Bc = (n, n, r, c, v)
    elif case in ("wide_cols", "huge_cols"):
        # block-rows 0 and 1 of A full over 300 block columns (6000 tasks each), block-rows 2 .. 5 short (one, two, 40 and 300 tiles);
        # every block-row of B: 20 one-value tiles, 12 of them at random block columns of 36 000, 8 at columns shared by all rows
        # (C tiles with 300 tasks whose order the sort must keep)
        # (huge_cols: 5 000 000 block columns = 23 column bits: FOUR counting passes of 6 bits, sort words of 64 bits whatever the switch)
        rng = np.random.default_rng(17)
        nk, nbc = 8 * 300, (36000 if case == "wide_cols" else 5000000)
        r = np.repeat(np.arange(16), nk); c = np.tile(np.arange(nk), 16)
        extra_r = np.concatenate([[16], [24, 24], np.full(40, 32), np.full(300, 40)])
        extra_c = np.concatenate([[8 * 7], [8 * 3, 8 * 250], 8 * np.arange(40) * 7, 8 * np.arange(300)])
        r = np.concatenate([r, extra_r]); c = np.concatenate([c, extra_c])
        A = (48, nk, r, c, np.sin(r * 7.0 + c))
        shared = rng.choice(nbc, 8, replace=False)
        rb, cb = [], []
        for br in range(300):
            cols = np.unique(np.concatenate([shared, rng.choice(nbc, 12, replace=False)]))
            for k in range(8):
                rb.append(np.full(cols.size, 8 * br + k)); cb.append(8 * cols + (k + br) % 8)
        rb = np.concatenate(rb); cb = np.concatenate(cb)
        Bc = (nk, 8 * nbc, rb, cb, np.cos(rb * 3.0 + cb))
    else:
        # 2 block-rows x 300 block-columns of A, all full; B's block-rows hold 20 / 40 tiles: 6000 / 12000 tasks per block-row of C
        nb_cols = 20 if case == "two_pieces" else 40
        nk = 8 * 300
        r = np.repeat(np.arange(16), nk); c = np.tile(np.arange(nk), 16)
        A = (16, nk, r, c, np.sin(r * 7.0 + c))
        rb = np.repeat(np.arange(nk), 8 * nb_cols); cb = np.tile(np.arange(8 * nb_cols), nk)
        Bc = (nk, 8 * nb_cols, rb, cb, np.cos(rb * 3.0 + cb))
    st = check_spgemm(oracle, bmsp, A, Bc, 0, 1, 5)
    assert st["sort_path"] == 1, st          # the segmented path took it (before round 3 a long segment fell back to the radix sort)
    st2 = check_spgemm(oracle, bmsp, A, Bc, 1, 1, 4)
    assert st2["sort_path"] == 1


@pytest.mark.parametrize("scale,ef", [(15, 8), (18, 1)])
def test_spgemm_long_segments_radix_by_default(bmsp, monkeypatch, scale, ef):
    """Hub block-rows on the expand-sort-compress pipeline (R-MAT, explicit segmented sort, no row-merge / window passes): segments of
    10^5 .. 10^6 tasks take the counting passes without being asked (stats.sort_long = 2; two passes at scale 15, three at 18), and C --
    keys, bitmaps, offsets and the fp32 V15 values, which depend on the task order inside every C tile -- equals the merge passes' C bit
    for bit.  (The merge form is pinned against the oracle by test_spgemm_long_segments_merge.)"""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(scale, ef)
    a = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=0)
    b = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=0)
    c_radix, st = bmsp.spgemm(a, b, mode=1, tc_version=5)
    assert st["sort_path"] == 1 and st["sort_long"] == 2, st
    monkeypatch.setenv("BMSP_SEGSORT_RADIX", "0")
    c_merge, st0 = bmsp.spgemm(a, b, mode=1, tc_version=5)
    assert st0["sort_path"] == 1 and st0["sort_long"] == 1, st0
    for x, y in zip(c_radix.host_arrays(), c_merge.host_arrays()):
        np.testing.assert_array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))


@pytest.mark.parametrize("lanes", ["", "8", "16", "32"])
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", ["fem", "banded64", "banded_wide", "rect_ragged", "filtered_run", "empty_strips", "long_b_rows", "cancel"])
def test_spgemm_rowsparse_block_mac(oracle, bmsp, monkeypatch, case, dtype, lanes):
    """block_mac_rowsparse_kernel forced on operands of every fill (BMSP_MAC_ROWSPARSE=1), V15's numerics (tc_version 5) on fp32 and on fp16
    operands (the reference's default configuration: products rounded to fp16, fp32 adds): the chain over the products of STORED values
    only, row-wise over CSR copies, accumulators per C value in LDS -- the values must be the oracle's bit for bit, with 8, 16 or 32
    lanes per row of C or the launcher's choice.  `fem`: nearly empty tiles (what the launcher picks the kernel for); `banded64` /
    `banded_wide`: full tiles -- rows of B of 129 and 513 entries (several iterations per entry of A), block-rows of C of up to 8 K values
    (four accumulator windows, each a walk of its own); `long_b_rows`: rows of B of ~190 entries against rows of A of a few;
    `cancel`: sums that end in +0 after exact cancellation."""
    from pybmsp import gen
    if case == "long_b_rows":
        rng = np.random.default_rng(5)
        ra = rng.integers(0, 40, 300); ca = rng.integers(0, 32, 300)
        rb = rng.integers(0, 32, 6000); cb = rng.integers(0, 8 * 150, 6000)
        A = (40, 32, ra, ca, rng.standard_normal(300))
        Bc = (32, 8 * 150, rb, cb, rng.standard_normal(6000))
        ka = ra.astype(np.int64) * 32 + ca; kb = rb.astype(np.int64) * 1200 + cb
        _, ia = np.unique(ka, return_index=True); _, ib = np.unique(kb, return_index=True)
        A = (40, 32, ra[ia], ca[ia], A[4][ia]); Bc = (32, 8 * 150, rb[ib], cb[ib], Bc[4][ib])
    elif case == "cancel":
        # rows of A = (x, -x, y) against columns of B = (1, 1, 0): the first two products cancel exactly, some sums end in +0
        n = 64
        r = np.repeat(np.arange(n), 3); c = (np.repeat(np.arange(n), 3) + np.tile([0, 1, 9], n)) % n
        va = np.tile([1.5, -1.5, 0.25], n) * (1 + np.repeat(np.arange(n), 3) % 3)
        rb = np.repeat(np.arange(n), 2); cb = (np.repeat(np.arange(n), 2) * 0 + np.tile([3, 17], n) + np.repeat(np.arange(n), 2) // 8 * 8) % n
        kb = rb * n + cb; _, ib = np.unique(kb, return_index=True)
        A, Bc = (n, n, r, c, va), (n, n, rb[ib], cb[ib], np.ones(ib.size))
    else:
        A, Bc, _ = _strip_case(gen, oracle, case)
        if Bc is None:
            Bc = A
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "1")
    monkeypatch.setenv("BMSP_MAC_ROWSPARSE", "1")
    if lanes:
        monkeypatch.setenv("BMSP_RS_LANES", lanes)
    st = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, 5)
    assert st["sort_path"] == 2 and (st["c_blocks"] == 0 or (st["mac_variant"] == 5 and st["mac_kernel"] == 5)), st


def test_spgemm_task_list_table_overflow(oracle, bmsp, monkeypatch):
    """A block-row of C with far more columns than the build pass's table holds (2400 against 1024 slots), every wave of the workgroup
    inserting from its first step on: the pass must give up -- a probe of a FULL table that never ended hung test_spgemm_synthetic[0-5-wide]
    once the four waves' inserts happened to pass the cap together -- and the product must come out of the next path, bit for bit the
    oracle's.  Several products in a row: the overflow is a matter of timing."""
    nb = 8
    ra = np.repeat(np.arange(8), 8 * nb); ca = np.tile(np.arange(8 * nb), 8)
    A = (8, 8 * nb, ra, ca, 1.0 + (ra * 3 + ca) % 5)
    rb, cb = [], []
    for k in range(nb):                      # B's block-row k: 300 tiles at block columns of its own
        cols = 8 * (300 * k + np.arange(300)) + (np.arange(300) % 8)
        for kk in range(8):
            rb.append(np.full(300, 8 * k + kk)); cb.append(cols)
    rb = np.concatenate(rb); cb = np.concatenate(cb)
    Bc = (8 * nb, 8 * 300 * nb, rb, cb, 1.0 + (rb + cb) % 3)
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "2")   # no strip mode: the task-list build pass is the first thing tried
    for _ in range(4):
        st = check_spgemm(oracle, bmsp, A, Bc, 0, 0, 5)
        assert st["c_blocks"] == 300 * nb, st


def test_spgemm_task_list_build_one_wave_form(oracle, bmsp, monkeypatch):
    """BMSP_RM_BUILD_WAVE=1: the row-merge build pass with one wave per block-row (the form before round 4's workgroup per block-row, kept
    for A/B runs): same C, same task order -- the fp32 V15 values, which depend on it, are the oracle's bit for bit (the vector-ALU
    kernel reads the task list: BMSP_MAC_ROWSPARSE=0)."""
    from pybmsp import gen
    n, _, r, c, v = gen.cage_like(40000, per_row=9.0)
    A = (n, n, r, c, np.round(v * 64) / 64)
    monkeypatch.setenv("BMSP_MAC_ROWSPARSE", "0")
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "2")   # task-list mode only
    st_wg = check_spgemm(oracle, bmsp, A, A, 0, 0, 5)
    monkeypatch.setenv("BMSP_RM_BUILD_WAVE", "1")
    st_wave = check_spgemm(oracle, bmsp, A, A, 0, 0, 5)
    assert st_wg["sort_path"] == 2 and st_wave["sort_path"] == 2 and st_wg["mac_variant"] == 0 and st_wave["mac_variant"] == 0, (st_wg, st_wave)


@pytest.mark.parametrize("dtype", [0, 1])
def test_spgemm_rowsparse_after_task_list(oracle, bmsp, dtype):
    """V15 numerics on operands of nearly empty tiles whose block-rows of C exceed strip mode's 256 tiles (here ~470 of them): the product
    takes the row-merge task-list mode for C's structure, and the numeric stage is still the row-sparse kernel -- with its 1024-slot table
    -- instead of the vector-ALU kernel on the task list.  Bit for bit the oracle's, fp32 and fp16 / tc_version 5."""
    rng = np.random.default_rng(23)
    n = 8 * 512
    def coo(per_block_row, seed, full_rows):
        # `per_block_row` tiles per block-row at random block columns; A's tiles hold one full row (all eight k of the tile: every tile of B's
        # block-row survives the bitmap filter), B's tiles two values
        r = np.random.default_rng(seed)
        br = np.repeat(np.arange(512), per_block_row); bc = r.integers(0, 512, br.size)
        key = np.unique(br.astype(np.int64) * 512 + bc)
        br, bc = key // 512, key % 512
        if full_rows:
            rows = np.repeat(br * 8 + r.integers(0, 8, br.size), 8); cols = (np.repeat(bc * 8, 8) + np.tile(np.arange(8), br.size))
        else:
            rows = np.concatenate([br * 8 + r.integers(0, 4, br.size), br * 8 + 4 + r.integers(0, 4, br.size)])
            cols = np.concatenate([bc * 8 + r.integers(0, 8, br.size), bc * 8 + r.integers(0, 8, br.size)])
        return (n, n, rows, cols, np.round(r.standard_normal(rows.size) * 16) / 16 + 0.03125)
    A, Bc = coo(20, 1, True), coo(25, 2, False)
    st = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, 5)
    per_row = st["c_blocks"] / 512.0
    assert 256 < per_row < 700, per_row
    assert st["sort_path"] == 2 and st["mac_variant"] == 5 and st["mac_kernel"] == 5, st


def test_mfma_f32_accumulation_order(bmsp):
    """v_mfma_f32_16x16x4_f32 accumulates its four k as the ascending fmaf chain (checked on the hardware against a host fmaf chain on
    random operands of mixed magnitude): the property the fp32 matrix-core block-MAC (BMSP_MAC_F32MFMA) rests on."""
    import ctypes as C
    bad = C.c_int(-1)
    bmsp.check(bmsp.lib().bmsp_selftest_mfma_f32_chain(C.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize("quota", ["64", ""])
@pytest.mark.parametrize("case", ["rmat", "banded_full", "hub_c_blocks", "rect_ragged", "fem", "cancel"])
def test_f32_mfma_block_mac(oracle, bmsp, monkeypatch, case, quota):
    """tc_version 5 on fp32 operands through block_mac_f32_mfma_kernel (opt-in, operands from lane-ordered tile copies; every task-count
    profile, default and smallest wave quota): V15's numerics -- the values must equal the oracle's k-ascending fmaf chain BIT FOR BIT on
    random fp32 values, and the vector-ALU kernel's on the same operands; `cancel` has sums that end in +0 / -0 and C tiles of unequal
    task counts in one pair."""
    from pybmsp import gen
    monkeypatch.setenv("BMSP_MAC_F32MFMA", "1")
    if quota:
        monkeypatch.setenv("BMSP_MAC_QUOTA", quota)
    if case == "rmat":
        n, _, r, c, v = gen.rmat(11, 8)
        A = Bc = (n, n, r, c, v)
    elif case == "banded_full":
        n, _, r, c, v = gen.banded(515, 20)
        A = Bc = (n, n, r, c, v)
    elif case == "hub_c_blocks":
        nk = 8 * 300
        r = np.repeat(np.arange(16), nk); c = np.tile(np.arange(nk), 16)
        A = (16, nk, r, c, np.sin(r * 7.0 + c))
        Bc = (nk, 16, c, r, np.cos(c * 3.0 + r))
    elif case == "rect_ragged":
        _, _, r1, c1, v1 = gen.random_coo(203, 77, 2500, seed=5)
        _, _, r2, c2, v2 = gen.random_coo(77, 331, 3000, seed=6)
        A, Bc = (203, 77, r1, c1, v1), (77, 331, r2, c2, v2)
    elif case == "fem":
        n, _, r, c, v = gen.fem_like(12, "27pt")
        A = Bc = (n, n, r, c, v)
    else:
        # rows of +-1 against columns of +-1: exact cancellations (sums that pass through and end in zero), next to longer task lists
        n = 64
        r = np.repeat(np.arange(n), n); c = np.tile(np.arange(n), n)
        keep = ((r // 8 + c // 8) % 3 != 1) | (r // 8 == 2)
        r, c = r[keep], c[keep]
        A = (n, n, r, c, np.where((r + c) % 2 == 0, 1.0, -1.0))
        Bc = (n, n, r, c, np.where((r * 3 + c) % 4 < 2, 1.0, -1.0) * np.where(r % 8 < 4, 1.0, -1.0))
    st = check_spgemm(oracle, bmsp, A, Bc, 0, 0, 5)       # not the MFMA tolerance branch: fp32 + tc 5 compares bit for bit
    assert st["mac_kernel"] == 5 and st["mac_variant"] == 4, st
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=0)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=0)
    new, _ = bmsp.spgemm(a, b, tc_version=5)
    monkeypatch.setenv("BMSP_MAC_F32MFMA", "0")
    old, sto = bmsp.spgemm(a, b, tc_version=5)
    assert sto["mac_variant"] == 0
    vo, vn = old.host_arrays()[3], new.host_arrays()[3]
    np.testing.assert_array_equal(vo.view(np.uint32), vn.view(np.uint32))


def _strip_case(gen, oracle, case):
    """(A, B, exact) for test_strip_block_mac: shapes that walk every leg of block_mac_strip_kernel."""
    if case == "banded64":          # 17 tiles per block-row, 33 C tiles: merged lists of 18 / 34, two windows, cursors carried across
        n, _, r, c, v = gen.banded(1029, 64)
        return (n, n, r, c, np.round(v * 8) / 8), None, False
    if case == "banded_wide":       # 5 windows per strip, 17 k-groups, odd number of block-rows (the last strip is a single row)
        n, _, r, c, v = gen.banded(8 * 71 + 3, 256)
        return (n, n, r, c, np.round(v * 4) / 4), None, False
    if case == "fem":               # sparse tiles, ~40 % of the candidate pairs dropped by the bitmap filter
        n, _, r, c, v = gen.fem_like(12, "27pt")
        return (n, n, r, c, np.round(v * 16) / 16), None, False
    if case == "fem_int":           # the integer stencil: every path exact
        n, _, r, c, v = gen.fem_like(10, "7pt", values="stencil")
        return (n, n, r, c, v), None, True
    if case == "rect_ragged":
        _, _, r1, c1, v1 = gen.random_coo(203, 77, 2500, seed=5, integer=True)
        _, _, r2, c2, v2 = gen.random_coo(77, 331, 3000, seed=6, integer=True)
        return (203, 77, r1, c1, v1), (77, 331, r2, c2, v2), True
    if case == "filtered_run":
        # block-row 0 of A: tile (0, 0) with ONLY its column 0 stored, tile (0, 1) full.  B's block-row 0: 100 tiles whose ONLY stored
        # row is 7 (every pair with A(0, 0) is dropped by the filter and none of their columns is a C column); B's block-row 1: tiles at
        # block-columns 0 and 120.  The strip's C columns are {0, 120}: the scan of B's block-row 0 meets > 32 tiles inside the window
        # that are not C columns (the refill loop), and must skip them all.
        ra, ca, va = [], [], []
        for i in range(8):
            ra.append(i); ca.append(0); va.append(1.0 + i)
            for k in range(8):
                ra.append(i); ca.append(8 + k); va.append(float((i + 2 * k) % 5 - 2) or 1.0)
        rb, cb, vb = [], [], []
        for j in range(1, 101):
            for cc in range(8):
                rb.append(7); cb.append(8 * j + cc); vb.append(float(cc + 1))
        for j in (0, 120):
            for kk in range(8):
                for cc in range(8):
                    rb.append(8 + kk); cb.append(8 * j + cc); vb.append(float((kk * 3 + cc) % 7 - 3) or 2.0)
        A = (13, 16, np.array(ra), np.array(ca), np.array(va))
        B = (16, 8 * 121, np.array(rb), np.array(cb), np.array(vb))
        return A, B, True
    if case == "empty_strips":      # block-rows 2..5 of A hold nothing (strips without a C tile), the last block-row is ragged
        n, _, r, c, v = gen.banded(100, 9)
        keep = (r < 16) | (r >= 48)
        return (n, n, r[keep], c[keep], np.round(v[keep] * 8) / 8), None, False
    raise ValueError(case)


@pytest.mark.parametrize("case", ["banded64", "banded_wide", "fem", "fem_int", "rect_ragged", "filtered_run", "empty_strips"])
def test_strip_block_mac(oracle, bmsp, monkeypatch, case):
    """block_mac_strip_kernel (tc_version 4, BMSP_MAC_STRIP): two block-rows of C per wave, k-groups of the merged column list of A,
    windows of 32 merged C columns, B's block-rows walked by cursor -- forced on small inputs that cross every boundary of that
    schedule, against the oracle's exact-product numerics (stage counters and C structure bit for bit, values exact on integer inputs
    and within the stated fp16 tolerance otherwise); the task-list kernel of the same tc_version must agree on the same operands."""
    from pybmsp import gen
    A, Bc, exact = _strip_case(gen, oracle, case)
    if Bc is None:
        Bc = A
    monkeypatch.setenv("BMSP_MAC_STRIP", "1")
    st = check_spgemm(oracle, bmsp, A, Bc, 1, 0, 4, exact_expected=exact)
    assert st["mac_kernel"] == 4 and st["mac_variant"] == 3, st
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=1)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=1)
    new, _ = bmsp.spgemm(a, b, tc_version=4)
    monkeypatch.setenv("BMSP_MAC_STRIP", "0")
    old, sto = bmsp.spgemm(a, b, tc_version=4)
    assert sto["mac_variant"] in (1, 2)
    vo, vn = old.host_arrays()[3], new.host_arrays()[3]
    if exact:
        np.testing.assert_array_equal(vo, vn)
    else:
        np.testing.assert_allclose(vn, vo, rtol=2e-3, atol=1e-5)
    if case == "banded64":  # non-finite operand values: the launcher must keep the task-list kernels (a skipped pair would turn 0 * inf into NaN)
        r, c, v = A[2], A[3], np.array(A[4], dtype=np.float64)
        v[7] = 1e9  # rounds to +inf in fp16
        a2 = bmsp.BmSpMatrix.from_coo(A[0], A[1], r, c, v, dtype=1)
        b2 = bmsp.BmSpMatrix.from_coo(A[0], A[1], r, c, v, transposed=True, dtype=1)
        monkeypatch.setenv("BMSP_MAC_STRIP", "1")
        _, st2 = bmsp.spgemm(a2, b2, tc_version=4)
        assert st2["mac_variant"] != 3


@pytest.mark.parametrize("case", ["banded64", "banded_wide", "fem", "fem_int", "rect_ragged", "filtered_run", "empty_strips", "all_filtered",
                                  "rmat_hub", "cage"])
@pytest.mark.parametrize("dtype", [1, 0])
def test_spgemm_rowmerge_path(oracle, bmsp, monkeypatch, case, dtype):
    """BMSP_SPGEMM_ROWMERGE=1: C's structure formed block-row by block-row in LDS (rowmerge.hip: hash of the surviving pairs' columns, OR of
    the tile-product bitmaps, rank by column) instead of expand - sort - compress.  Stage counters, keys, bitmaps and offsets are the
    oracle's bit for bit; the values are the strip kernel's, hence bit-identical with the pipeline + strip kernel on the same operands.  A
    block-row of C beyond the pass's capacity (rmat_hub) goes to the column-window passes (test_spgemm_rowwindow_path).  dtype 0: fp32 operands, the strip kernel on
    v_mfma_f32_16x16x4_f32 -- V15's summation order, so the values are the oracle's (and the vector-ALU kernel's) bit for bit."""
    from pybmsp import gen
    exact = False
    if case == "all_filtered":  # A only touches column 0 of every tile, B only row 7: every candidate pair dies in the bitmap filter
        n = 2048
        ra = np.arange(n); ca = (ra // 8) * 8
        rb = (np.arange(n) // 8) * 8 + 7; cb = np.arange(n)
        A, Bc, exact = (n, n, ra, ca, np.ones(n)), (n, n, rb, cb, np.ones(n)), True
    elif case == "rmat_hub":    # hub block-rows of C with thousands of tiles
        n, _, r, c, v = gen.rmat(13, 16)
        A = Bc = (n, n, r, c, np.round(v * 8) / 8)
    elif case == "cage":        # irregular columns: ~70 C tiles per block-row (at most 177), few tasks per C tile
        n, _, r, c, v = gen.cage_like(40000, per_row=3.0)
        A = Bc = (n, n, r, c, np.round(v * 64) / 64)
    else:
        A, Bc, exact = _strip_case(gen, oracle, case)
        if Bc is None:
            Bc = A
    tc = 4 if dtype == 1 else 5
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "1")
    st = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, tc, exact_expected=exact)
    if case == "rmat_hub":
        assert st["sort_path"] == 3, st  # round 4: hub block-rows take the column-window passes (rowwindow.hip), not the pipeline
        return
    # fp32: the strip kernel on the matrix cores (3) or, for nearly empty tiles, the row-sparse kernel (5); each is also run where the
    # launcher would have taken the other
    assert st["sort_path"] == 2 and (st["c_blocks"] == 0 or (st["mac_variant"] in ((3, 5) if dtype == 0 else (3,)) and st["mac_kernel"] == tc)), st
    if dtype == 0:
        for forced, variant in (("1", 5), ("0", 3)):
            monkeypatch.setenv("BMSP_MAC_ROWSPARSE", forced)
            stf = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, tc, exact_expected=exact)
            assert stf["sort_path"] == 2 and (stf["c_blocks"] == 0 or stf["mac_variant"] == variant), stf
        monkeypatch.delenv("BMSP_MAC_ROWSPARSE")
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=dtype)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=dtype)
    new, _ = bmsp.spgemm(a, b, tc_version=tc)
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "0")
    monkeypatch.setenv("BMSP_MAC_STRIP", "1")
    monkeypatch.setenv("BMSP_MAC_ROWSPARSE", "0")   # the pipeline's own numeric stage: the vector-ALU kernel on the sorted task list
    old, sto = bmsp.spgemm(a, b, tc_version=tc)
    monkeypatch.delenv("BMSP_MAC_ROWSPARSE")
    assert sto["sort_path"] in (0, 1) and (dtype == 1 or sto["mac_variant"] == 0)
    for x, y in zip(old.host_arrays(), new.host_arrays()):
        np.testing.assert_array_equal(x, y)
    # an explicit sort mode is honoured: the pipeline runs
    monkeypatch.delenv("BMSP_SPGEMM_ROWMERGE")
    _, st2 = bmsp.spgemm(a, b, mode=2, tc_version=tc)
    assert st2["sort_path"] == 0


@pytest.mark.parametrize("case", ["banded64", "banded_wide", "fem", "rect_ragged", "filtered_run", "empty_strips", "cage_wide"])
@pytest.mark.parametrize("dtype,tc", [(1, 4), (0, 5), (1, 5), (2, 5), (1, 1)])
def test_spgemm_rowmerge_task_list(oracle, bmsp, monkeypatch, case, dtype, tc):
    """Task-list mode of the row-merge path (BMSP_SPGEMM_ROWMERGE=2 skips strip mode): a count pass and a fill pass form C's keys,
    bitmaps AND the sorted task list per block-row in LDS; the block-MAC kernels of the tc_version run from it.  The list is the
    pipeline's (C key order, ascending A tile inside a C tile), so every array of C equals the pipeline's bit for bit -- for every value
    type and kernel -- and the oracle's within check_spgemm's terms.  cage_wide: block-rows of C beyond the strip kernel's capacity take
    this mode without the switch."""
    from pybmsp import gen
    exact = False
    if case == "cage_wide":
        n, _, r, c, v = gen.cage_like(24000, per_row=10.0)  # up to 310 C tiles per block-row
        A = Bc = (n, n, r, c, np.round(v * 64) / 64)
    else:
        A, Bc, exact = _strip_case(gen, oracle, case)
        if Bc is None:
            Bc = A
        monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "2")
        monkeypatch.setenv("BMSP_MAC_STRIP", "0")  # (dense bands would otherwise take the strip kernel from the task-list structure too)
    st = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, tc, exact_expected=exact)
    assert st["sort_path"] == 2 and st["mac_variant"] != 3, st
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=dtype)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=dtype)
    new, stn = bmsp.spgemm(a, b, tc_version=tc)
    new2, _ = bmsp.spgemm(a, b, tc_version=tc)  # (the pair's mode is remembered on the handle: straight to the task-list passes)
    assert stn["sort_path"] == 2
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "0")
    monkeypatch.setenv("BMSP_MAC_STRIP", "0")
    old, sto = bmsp.spgemm(a, b, tc_version=tc)
    assert sto["sort_path"] in (0, 1) and sto["mac_variant"] == stn["mac_variant"]
    for x, y, z in zip(old.host_arrays(), new.host_arrays(), new2.host_arrays()):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, z)


@pytest.mark.parametrize("case", ["rmat13", "rmat11_narrow", "hub_row_diag", "hub_rmat", "rect_wide", "long_a_rows", "filtered", "cage", "empty", "wide_hashed",
                                  "hash_overflow", "rmat15_wide"])
@pytest.mark.parametrize("dtype,tc", [(0, 5), (1, 4), (1, 5), (2, 5)])
def test_spgemm_rowwindow_path(oracle, bmsp, monkeypatch, case, dtype, tc):
    """Column-window passes (rowwindow.hip; round 4): a workgroup per (block-row of A, window of C's block columns), dense tables in LDS,
    the order of a C tile's tasks from 64-bit hit masks over rounds of 64 A tiles.  Taken by the library for operands with hub block-rows
    (rmat13: the row-merge task-list pass refuses them); forced here for the other shapes (BMSP_SPGEMM_ROWWINDOW=1) and with small windows
    (BMSP_WIN_CAND) so that small inputs cross every boundary: several windows per block-row, windows that hold no tile, block-rows of A of
    more than 64 tiles (several rounds), a block-row whose candidates all die in the filter, B wider than one window.  Stage counters,
    keys, bitmaps, offsets: the oracle's bit for bit; every array of C: the pipeline's bit for bit (task order = summation order)."""
    from pybmsp import gen
    exact = False
    force, cand = True, None
    if case == "rmat13":      # what the library picks by itself
        n, _, r, c, v = gen.rmat(13, 16)
        A = Bc = (n, n, r, c, np.round(v * 8) / 8)
        force = False
    elif case == "rmat11_narrow":  # 64-column windows: 4 per block-row, most of them nearly empty
        n, _, r, c, v = gen.rmat(11, 8)
        A = Bc = (n, n, r, c, np.round(v * 8) / 8)
        cand = 64
    elif case in ("hub_row_diag", "hub_rmat"):
        n, r, c, v = _hub_matrix(gen, "diag" if case == "hub_row_diag" else "rmat")
        A = Bc = (n, n, r, c, np.round(np.asarray(v, dtype=np.float64) * 16) / 16)
        cand = 512
    elif case == "rect_wide":  # B has 2400 block columns: two windows by the table size alone; ragged edges
        rng = np.random.default_rng(7)
        m, k, nn = 300, 411, 19195
        ra, ca = rng.integers(0, m, 4000), rng.integers(0, k, 4000)
        rb, cb = rng.integers(0, k, 30000), rng.integers(0, nn, 30000)
        A = (m, k, ra, ca, rng.integers(1, 5, 4000).astype(np.float64))
        Bc = (k, nn, rb, cb, rng.integers(1, 5, 30000).astype(np.float64))
        exact = True
    elif case == "long_a_rows":  # block-rows of A of ~200 tiles: four rounds of 64, tasks of one C tile spread over them
        rng = np.random.default_rng(11)
        m, k = 64, 1600
        ra = rng.integers(0, m, 20000); ca = rng.integers(0, k, 20000)
        rb = np.repeat(np.arange(k), 6); cb = (rb * 7 + np.tile(np.arange(6), k) * 5) % 300
        A = (m, k, ra, ca, rng.integers(1, 4, 20000).astype(np.float64))
        Bc = (k, 300, rb, cb, np.ones(rb.size))
        exact = True
        cand = 2048
    elif case == "filtered":
        n = 2048
        ra = np.arange(n); ca = (ra // 8) * 8
        rb = (np.arange(n) // 8) * 8 + 7; cb = np.arange(n)
        A, Bc, exact = (n, n, ra, ca, np.ones(n)), (n, n, rb, cb, np.ones(n)), True
    elif case == "cage":
        n, _, r, c, v = gen.cage_like(20000, per_row=6.0)
        A = Bc = (n, n, r, c, np.round(v * 64) / 64)
        cand = 256
    elif case in ("wide_hashed", "hash_overflow"):
        # B has 40 000 block columns: windows wider than the dense tables -> hashed slots (open addressing, C's key order by an LDS sort).
        # ~6400 candidate pairs and ~6000 C tiles per block-row.  hash_overflow: cut for 16384 pairs per window, one window would hold twice
        # what a table takes -- the pass reports it, the host cuts finer and runs it again (three times here) instead of giving up
        rng = np.random.default_rng(23)
        m, k, nn = 400, 4000, 320000
        ra = np.repeat(np.arange(m), 10); ca = rng.integers(0, k, ra.size)
        rb = np.repeat(np.arange(k), 10); cb = rng.integers(0, nn, rb.size)
        A = (m, k, ra, ca, rng.integers(1, 4, ra.size).astype(np.float64))
        Bc = (k, nn, rb, cb, rng.integers(1, 4, rb.size).astype(np.float64))
        exact = True
        if case == "hash_overflow":
            monkeypatch.setenv("BMSP_WIN_CAND_HASH", "16384")
    elif case == "rmat15_wide":  # a power-law operand wider than the dense tables reach: dense windows over the hub columns, hashed ones over the tail
        n, _, r, c, v = gen.rmat(15, 4)
        A = (n, n, r, c, np.round(v * 8) / 8)
        Bc = (n, 8 * n, r, c * 8, np.round(v * 8) / 8)   # B's columns spread over 32768 block columns
    else:  # empty: A has block-rows without a tile and tiles whose block column has no block-row in B
        A = (100, 64, np.array([0, 3, 90, 91]), np.array([1, 60, 2, 63]), np.array([1.0, 2.0, 3.0, 4.0]))
        Bc = (64, 40, np.array([1, 2, 2]), np.array([0, 39, 17]), np.array([1.0, 1.0, 2.0]))
        exact = True
    if force:
        monkeypatch.setenv("BMSP_SPGEMM_ROWWINDOW", "1")
        monkeypatch.setenv("BMSP_WIN_THIN", "1")  # (the library leaves products with a handful of pairs per (A tile, window) to the pipeline: taken here all the same)
    if cand:
        monkeypatch.setenv("BMSP_WIN_CAND", str(cand))
    st = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, tc, exact_expected=exact)
    assert st["sort_path"] == 3, st
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=dtype)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=dtype)
    new, stn = bmsp.spgemm(a, b, tc_version=tc)
    new2, stn2 = bmsp.spgemm(a, b, tc_version=tc)  # (remembered on the handle: straight to the window passes)
    assert stn["sort_path"] == 3 and stn2["sort_path"] == 3
    monkeypatch.setenv("BMSP_SPGEMM_ROWWINDOW", "0")
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "0")
    old, sto = bmsp.spgemm(a, b, tc_version=tc)
    assert sto["sort_path"] in (0, 1) and sto["mac_variant"] == stn["mac_variant"]
    for key in ("task_list_size", "bmp_reduction", "surviving_tasks", "c_blocks", "c_nnz"):
        assert stn[key] == sto[key], key
    for x, y, z in zip(old.host_arrays(), new.host_arrays(), new2.host_arrays()):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, z)


@pytest.mark.parametrize("scale", [1.0, 1e-20, 3e-23, 1e19])
def test_spgemm_fp32_exponent_range_keeps_v15(oracle, bmsp, scale):
    """fp32 products whose partial products leave the normal range: v_mfma_f32_16x16x4_f32 rounds products around 2^-149 differently
    from fmaf (measured: 3e-23-scaled FEM values), so the fp32 strip kernel is only taken while |a| |b| >= 2^-126 for every pair of stored
    values (and sums stay far from overflow); otherwise the vector-ALU kernel runs from the row-merge task list.  Either way C is the
    oracle's bit for bit -- denormal results included."""
    from pybmsp import gen
    n, _, r, c, v = gen.fem_like(10, "27pt")
    vv = (v * scale).astype(np.float32).astype(np.float64)
    A = (n, n, r, c, vv)
    st = check_spgemm(oracle, bmsp, A, A, 0, 0, 5, exact_expected=True)
    assert st["sort_path"] == 2, st
    if scale == 3e-23:
        assert st["mac_variant"] not in (3, 5), st   # products underflow: neither the matrix pipe nor the chain without its zero terms
    elif scale == 1e19:
        assert st["mac_variant"] != 3, st            # sums may overflow: not the matrix pipe (the row-sparse chain is the reference's own operations)
    elif scale == 1.0:
        assert st["mac_variant"] in (3, 5), st


@pytest.mark.parametrize("case,dtype,tc", [("fem", 0, 5), ("fem", 1, 4), ("banded64", 1, 4), ("rect_ragged", 0, 5), ("fem", 2, 5), ("fem", 1, 5),
                                           ("rmat_hub", 1, 4)])
def test_spgemm_symbolic_numeric_split(oracle, bmsp, case, dtype, tc):
    """bmsp_spgemm_symbolic gives the product's structure with zero values; bmsp_spgemm_numeric fills a C of that structure with the
    values bmsp_spgemm would store -- bit for bit -- also for NEW operand values on the same structure (the use the split exists for),
    whether a strip kernel does it alone (fp16 tc 4, fp32) or the whole product runs behind it (fp64, fp16 V15, hub rows).  A C of a
    different structure is refused."""
    from pybmsp import gen
    if case == "rmat_hub":
        n, _, r, c, v = gen.rmat(11, 8)
        A = Bc = (n, n, r, c, np.round(v * 8) / 8)
    else:
        A, Bc, _ = _strip_case(gen, oracle, case)
        if Bc is None:
            Bc = A
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=dtype)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=dtype)
    full, stf = bmsp.spgemm(a, b, tc_version=tc)
    sym, sts = bmsp.spgemm_symbolic(a, b, tc_version=tc)
    kf, bf, of, vf = full.host_arrays()
    ks, bs, os_, vs = sym.host_arrays()
    np.testing.assert_array_equal(kf, ks); np.testing.assert_array_equal(bf, bs); np.testing.assert_array_equal(of, os_)
    assert not vs.any() and sts["c_nnz"] == stf["c_nnz"] and sts["surviving_tasks"] == stf["surviving_tasks"]
    stn = bmsp.spgemm_numeric(a, b, sym, tc_version=tc)
    np.testing.assert_array_equal(sym.host_arrays()[3], vf)
    if (dtype == 0 or (dtype == 1 and tc == 4)) and case != "rmat_hub":
        assert stn["mac_variant"] in ((3, 5) if dtype == 0 else (3,)), stn  # the strip (or, fp32 with nearly empty tiles, the row-sparse) kernel alone
    assert stn["t_us"][3] == 0 and stn["t_us"][4] == 0 and stn["t_us"][5] == 0 and stn["t_us"][7] > 0, stn  # no symbolic stage ran: T_7 from C's kept task list
    # new values, same structure
    A2 = A[:4] + (np.asarray(A[4]) * 0.5,)
    B2 = Bc[:4] + (np.asarray(Bc[4]) * -2.0,)
    a2 = bmsp.BmSpMatrix.from_coo(*A2, dtype=dtype)
    b2 = bmsp.BmSpMatrix.from_coo(*B2, transposed=True, dtype=dtype)
    full2, _ = bmsp.spgemm(a2, b2, tc_version=tc)
    bmsp.spgemm_numeric(a2, b2, sym, tc_version=tc)
    np.testing.assert_array_equal(sym.host_arrays()[3], full2.host_arrays()[3])
    # a C of another structure
    if case == "fem":
        n2, _, r2, c2, v2 = gen.banded(A[0], 3)
        other, _ = bmsp.spgemm(bmsp.BmSpMatrix.from_coo(n2, n2, r2, c2, v2, dtype=dtype), bmsp.BmSpMatrix.from_coo(n2, n2, r2, c2, v2, transposed=True, dtype=dtype),
                               tc_version=tc)
        # every path refuses it: a product carries a fingerprint of its operands' structures (round 4; before, the strip kernels trusted the caller)
        with pytest.raises(Exception):
            bmsp.spgemm_numeric(a, b, other, tc_version=tc)
        # ... and so is a C whose OPERANDS changed structure while C kept the old one
        with pytest.raises(Exception):
            bmsp.spgemm_numeric(bmsp.BmSpMatrix.from_coo(n2, n2, r2, c2, v2, dtype=dtype), bmsp.BmSpMatrix.from_coo(n2, n2, r2, c2, v2, transposed=True, dtype=dtype), sym, tc_version=tc)


@pytest.mark.parametrize("seed", range(24))
def test_spgemm_rowmerge_random_shapes(oracle, bmsp, monkeypatch, seed):
    """Seeded random rectangular operands (ragged edges, empty block-rows and block-columns, clustered and scattered columns, explicit
    duplicates, 1 ... ~40 tiles per block-row): whatever mode the library picks under sort mode 0 -- strip mode, task-list mode or the
    pipeline -- every array of C and every stage counter equals the pipeline's (BMSP_SPGEMM_ROWMERGE=0) bit for bit, and for the V15
    numerics the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    m, k, n = (int(rng.integers(9, 700)) for _ in range(3))
    dtype, tc = [(0, 5), (1, 4), (1, 5), (2, 5)][seed % 4]

    def rand_coo(rows, cols, nnz, clustered):
        r = rng.integers(0, rows, nnz)
        if clustered:  # columns near a moving diagonal: few distinct C columns per block-row
            c = np.clip((r * cols) // max(rows, 1) + rng.integers(-12, 13, nnz), 0, cols - 1)
        else:
            c = rng.integers(0, cols, nnz)
        if seed % 3 == 0:  # some empty block-rows / block-columns
            keep = ((r // 8) % 5 != 3) & ((c // 8) % 7 != 2)
            r, c = r[keep], c[keep]
        v = rng.integers(-4, 5, len(r)).astype(np.float64)
        v[v == 0] = 1.0
        return rows, cols, r, c, v

    A = rand_coo(m, k, int(rng.integers(1, 12)) * m, seed % 2 == 0)
    Bc = rand_coo(k, n, int(rng.integers(1, 12)) * k, seed % 4 < 2)
    st = check_spgemm(oracle, bmsp, A, Bc, dtype, 0, tc, exact_expected=True)  # (small integers: exact on every path)
    a = bmsp.BmSpMatrix.from_coo(*A, dtype=dtype)
    b = bmsp.BmSpMatrix.from_coo(*Bc, transposed=True, dtype=dtype)
    new, stn = bmsp.spgemm(a, b, tc_version=tc)
    monkeypatch.setenv("BMSP_SPGEMM_ROWMERGE", "0")
    old, sto = bmsp.spgemm(a, b, tc_version=tc)
    assert sto["sort_path"] in (0, 1)
    for key in ("task_list_size", "bmp_reduction", "surviving_tasks", "c_blocks", "c_nnz"):
        assert stn[key] == sto[key] == st[key], key
    for x, y in zip(old.host_arrays(), new.host_arrays()):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("case", ["rmat", "banded", "filtered", "rect"])
def test_spgemm_single_pass_expansion(oracle, bmsp, monkeypatch, case):
    """BMSP_EXPAND_LOOKBACK: T_3 + T_4 as one decoupled look-back pass (survivors written at the running prefix of the earlier tiles);
    same task order, hence the same C bit for bit as the count + write pair."""
    from pybmsp import gen
    monkeypatch.setenv("BMSP_EXPAND_LOOKBACK", "1")
    if case == "rmat":
        n, _, r, c, v = gen.rmat(12, 8)
        A = Bc = (n, n, r, c, v)
    elif case == "banded":
        n, _, r, c, v = gen.banded(20011, 9)
        A = Bc = (n, n, r, c, v)
    elif case == "filtered":
        # A only touches column 0 of every tile, B only row 7: every candidate pair dies in the bitmap filter
        n = 2048
        ra = np.arange(n); ca = (ra // 8) * 8
        rb = (np.arange(n) // 8) * 8 + 7; cb = np.arange(n)
        A, Bc = (n, n, ra, ca, np.ones(n)), (n, n, rb, cb, np.ones(n))
    else:
        _, _, r1, c1, v1 = gen.random_coo(3001, 517, 40000, seed=21)
        _, _, r2, c2, v2 = gen.random_coo(517, 2203, 30000, seed=22)
        A, Bc = (3001, 517, r1, c1, v1), (517, 2203, r2, c2, v2)
    for dtype, tc in ((0, 5), (1, 4)):
        check_spgemm(oracle, bmsp, A, Bc, dtype, 0, tc)


def _cusp_pairs():
    g = json.load(open(os.path.join(GOLDEN, "cusp_multiply.json")))
    return [(p["left"], p["right"]) for p in g["products"]]


@pytest.mark.parametrize("pair", _cusp_pairs(), ids=lambda p: "%sx%s" % p)
def test_spgemm_cusp_known_answers(oracle, bmsp, pair):
    """cusp/testing/multiply.cu:39-128: every compatible pair of the literal / gallery matrices A..K, expected value = the dense
    product committed in tests/golden/cusp_multiply.json; fp32 and fp16, every block-MAC version, every sort mode.  Values are
    small multiples of 0.5 -- exact in fp16 -- so even the matrix-core path must match bit for bit; the oracle's structure and
    stage counters are checked alongside (check_spgemm)."""
    g = json.load(open(os.path.join(GOLDEN, "cusp_multiply.json")))
    L, R = np.asarray(g["matrices"][pair[0]]["dense"], dtype=np.float64), np.asarray(g["matrices"][pair[1]]["dense"], dtype=np.float64)
    want = np.asarray([p for p in g["products"] if (p["left"], p["right"]) == pair][0]["dense"], dtype=np.float64)
    r1, c1 = np.nonzero(L)
    r2, c2 = np.nonzero(R)
    A_coo = (L.shape[0], L.shape[1], r1.astype(np.int32), c1.astype(np.int32), L[r1, c1])
    B_coo = (R.shape[0], R.shape[1], r2.astype(np.int32), c2.astype(np.int32), R[r2, c2])
    for dtype, tcs in ((0, (5,)), (1, (5, 4, 3, 2, 1)), (2, (5,))):
        for tc in tcs:
            for mode in (0, 1, 2):
                check_spgemm(oracle, bmsp, A_coo, B_coo, dtype, mode, tc, exact_expected=True)
        A = bmsp.BmSpMatrix.from_coo(*A_coo, dtype=dtype)
        B = bmsp.BmSpMatrix.from_coo(*B_coo, transposed=True, dtype=dtype)
        Cm, _ = bmsp.spgemm(A, B, tc_version=4 if dtype == 1 else 5)
        k, b, o, v = Cm.host_arrays()
        got = np.zeros_like(want)
        for (i, j), val in util.bmsp_host_to_dok(want.shape[0], want.shape[1], k, b, o, v).items():
            got[i, j] = val
        np.testing.assert_array_equal(got, want)
        # the host CSR path of CSRMatrix::multiply on the same pair
        if dtype == 0:
            import scipy.sparse as sp
            la, ra = sp.csr_matrix(L.astype(np.float32)), sp.csr_matrix(R.astype(np.float32))
            la.sort_indices(); ra.sort_indices()
            ca = bmsp.CSRMatrix.from_arrays(L.shape[0], L.shape[1], la.indptr, la.indices, la.data)
            cb = bmsp.CSRMatrix.from_arrays(R.shape[0], R.shape[1], ra.indptr, ra.indices, ra.data)
            nr, nc, ro, cc, vv = ca.multiply(cb).arrays()
            np.testing.assert_array_equal(sp.csr_matrix((vv, cc, ro), shape=(nr, nc)).toarray().astype(np.float64), want)


@pytest.mark.parametrize("path", [p for p in util.all_fixture_mtx() if "real_general" not in p])
@pytest.mark.parametrize("dtype,tc", [(0, 5), (1, 5), (1, 4)])
def test_spgemm_fixtures_square(oracle, bmsp, path, dtype, tc):
    coo = oracle.mtx_read(path)
    if coo.num_rows != coo.num_cols:
        pytest.skip("not square")
    t = (coo.num_rows, coo.num_cols, coo.rows, coo.cols, coo.vals)
    for mode in (2, 1):
        # integer-valued fixtures: products and sums are exact in fp16/fp32 -> even the MFMA path is bit-exact
        check_spgemm(oracle, bmsp, t, t, dtype, mode, tc, exact_expected=True)


@pytest.mark.parametrize("case", ["rect", "banded", "rmat", "empty_rows", "filtered", "single_block", "wide", "long_segments", "hub_c_blocks", "hub_row"])
@pytest.mark.parametrize("dtype,tc", [(0, 5), (1, 5), (1, 4), (2, 5)])
def test_spgemm_synthetic(oracle, bmsp, case, dtype, tc):
    from pybmsp import gen
    if case == "rect":
        A = gen.random_coo(70, 45, 900, seed=1, lo=0, hi=1)
        B = gen.random_coo(45, 123, 1100, seed=2, lo=0, hi=1)
    elif case == "banded":
        A = B = gen.banded(600, 5)
    elif case == "rmat":
        A = B = gen.rmat(10, 6)
    elif case == "empty_rows":
        A = gen.random_coo(500, 500, 400, seed=3, lo=0, hi=1)   # B has many empty block-rows; fan-out 0 tasks
        B = gen.random_coo(500, 500, 150, seed=4, lo=0, hi=1)
    elif case == "wide":
        # 2^24 columns: 21 column bits + 12 position bits do not fit a 32-bit sort word -> 64-bit words in the segmented sort
        A = gen.random_coo(300, 200, 3000, seed=5, lo=0, hi=1)
        B = gen.random_coo(200, 1 << 24, 6000, seed=6, lo=0, hi=1)
    elif case == "long_segments":
        # block-rows of A with ~300 .. ~3000 tasks: both LDS sort kernels (one wave <= 1024 tasks, workgroup <= 4096)
        _, _, r1, c1, v1 = gen.random_coo(8, 400, 1500, seed=8, lo=0, hi=1)    # one heavy block-row
        _, _, r2, c2, v2 = gen.random_coo(64, 400, 1000, seed=9, lo=0, hi=1)
        rc, first = np.unique(np.concatenate([r1, r2]) * 400 + np.concatenate([c1, c2]), return_index=True)
        A = (64, 400, rc // 400, rc % 400, np.concatenate([v1, v2])[first])
        B = gen.random_coo(400, 3000, 12000, seed=10, lo=0, hi=1)
    elif case == "hub_c_blocks":
        # C(0,0) and C(0,1) collect 300 tasks each (runs that cross several 64-task waves of the bitmap pass and several
        # 64-task windows of the block-MAC), next to ordinary short runs
        ra, ca = np.meshgrid(np.arange(8), np.arange(0, 2400, 3), indexing="ij")
        rb, cb = np.meshgrid(np.arange(0, 2400, 5), np.arange(16), indexing="ij")
        _, _, r2, c2, v2 = gen.random_coo(64, 2400, 700, seed=11, lo=0, hi=1)
        rc, first = np.unique(np.concatenate([ra.ravel(), r2]) * 2400 + np.concatenate([ca.ravel(), c2]), return_index=True)
        A = (64, 2400, rc // 2400, rc % 2400, np.concatenate([np.ones(ra.size), v2])[first])
        B = (2400, 16, rb.ravel(), cb.ravel(), np.ones(rb.size) * 0.5)
    elif case == "hub_row":
        # one block-row of A with 5600 surviving tasks (> 4096: the segmented path has to hand over to the global sort, which
        # it only learns at its next read-back) next to ordinary rows
        ra = np.zeros(5600, dtype=np.int64); ca = np.arange(5600)
        _, _, r2, c2, v2 = gen.random_coo(40, 5600, 900, seed=12, lo=0, hi=1)
        rc, first = np.unique(np.concatenate([ra, r2 + 8]) * 5600 + np.concatenate([ca, c2]), return_index=True)
        A = (48, 5600, rc // 5600, rc % 5600, np.concatenate([np.ones(5600), v2])[first])
        rb, cb = np.meshgrid(np.arange(5600), np.arange(0, 64, 8), indexing="ij")
        B = (5600, 64, rb.ravel(), cb.ravel(), np.full(rb.size, 0.5))
    elif case == "filtered":
        # A uses only even k, B only odd k inside every tile: every candidate pair dies in the bitmap filter
        n = 256
        ra = np.repeat(np.arange(n), 2); ca = (ra // 8) * 8 + np.tile([0, 2], n)
        rb = np.repeat(np.arange(n), 2); rb = (rb // 8) * 8 + np.tile([1, 3], n); cb = np.repeat(np.arange(n), 2)
        A = (n, n, ra, ca, np.ones(ra.size)); B = (n, n, rb, cb, np.ones(rb.size))
    else:
        A = (8, 8, np.array([0, 3]), np.array([1, 1]), np.array([2.0, 3.0]))
        B = (8, 8, np.array([1, 1]), np.array([0, 7]), np.array([5.0, 7.0]))
    for mode in (2, 1, 0):
        st = check_spgemm(oracle, bmsp, A, B, dtype, mode, tc)
    if case == "filtered":
        assert st["surviving_tasks"] == 0 and st["c_blocks"] == 0 and st["bmp_reduction"] == st["task_list_size"] > 0


@pytest.mark.parametrize("seed", list(range(24)))
def test_spgemm_randomized_differential(oracle, bmsp, seed):
    """seeded random shapes / densities / value types against the oracle: exercises group and window boundaries of the
    block-MAC kernels (C block counts mod 4 and mod 16, task lists around 64), empty operands, single rows and columns."""
    from pybmsp import gen
    rng = np.random.default_rng(1000 + seed)
    m, k, n = (int(rng.integers(1, 260)) for _ in range(3))
    if seed % 6 == 0:
        m = int(rng.integers(1, 9))        # a single block-row
    if seed % 6 == 1:
        n = int(rng.integers(1, 9))        # a single block-column
    dens = float(rng.choice([0.002, 0.02, 0.1, 0.5]))
    nnz_a = int(m * k * dens * rng.uniform(0.3, 1.5)); nnz_b = int(k * n * dens * rng.uniform(0.3, 1.5))
    if seed == 5:
        nnz_b = 0                           # empty B
    A = gen.random_coo(m, k, nnz_a, seed=2 * seed + 1, lo=0, hi=1)
    B = gen.random_coo(k, n, nnz_b, seed=2 * seed + 2, lo=0, hi=1)
    dtype, tc = [(0, 5), (1, 5), (1, 4), (2, 5), (1, 2)][seed % 5]
    for mode in (0, 1, 2):
        check_spgemm(oracle, bmsp, A, B, dtype, mode, tc)


def test_spgemm_chain_feeds_back(oracle, bmsp):
    """C is a valid A operand (normal layout, keys ascending): (A*A)*A runs and matches the oracle chain."""
    from pybmsp import gen
    n, _, r, c, v = gen.banded(400, 3)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    At = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True)
    C1, _ = bmsp.spgemm(A, At)
    C2, _ = bmsp.spgemm(C1, At)
    oA = oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v), 0, False)
    oAt = oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v), 0, True)
    oC1, _ = oracle.spgemm(oA, oAt)
    oC2, _ = oracle.spgemm(oC1, oAt)
    k, b, o, vals = C2.host_arrays()
    np.testing.assert_array_equal(k, oC2.keys)
    np.testing.assert_array_equal(b, oC2.bmps)
    np.testing.assert_array_equal(vals.astype(np.float64), oC2.values)


def test_spgemm_argument_errors(bmsp):
    from pybmsp import gen
    n, _, r, c, v = gen.banded(64, 2)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    At = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True)
    with pytest.raises(bmsp.BmspError):
        bmsp.spgemm(A, A)       # B not transposed
    with pytest.raises(bmsp.BmspError):
        bmsp.spgemm(At, At)     # A transposed
    W = bmsp.BmSpMatrix.from_coo(32, 32, [0], [0], [1.0], transposed=True)
    with pytest.raises(bmsp.BmspError):
        bmsp.spgemm(A, W)       # shape mismatch
    with pytest.raises(bmsp.BmspError):
        bmsp.spmv(At, bmsp.DeviceArray.from_host(np.ones(n, np.float32)))


def test_spgemm_properties_large(bmsp):
    """BASELINE-scale structural properties without the oracle: C = A*I reproduces A; symbolic nnz of A*A
    equals the pattern product computed by scipy; row sums of C equal A*(A*1)."""
    from pybmsp import gen
    import scipy.sparse as sp
    n, _, r, c, v = gen.rmat(15, 6)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    It = bmsp.BmSpMatrix.from_coo(n, n, np.arange(n), np.arange(n), np.ones(n), transposed=True)
    Cm, st = bmsp.spgemm(A, It, mode=1)
    ka, ba, oa, va = A.host_arrays()
    kc, bc, oc, vc = Cm.host_arrays()
    np.testing.assert_array_equal(ka, kc); np.testing.assert_array_equal(ba, bc)
    np.testing.assert_array_equal(oa, oc); np.testing.assert_array_equal(va, vc)
    At = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True)
    for mode in (1, 2):
        C2, st2 = bmsp.spgemm(A, At, mode=mode)
        S = sp.coo_matrix((np.ones(r.size), (r, c)), shape=(n, n)).tocsr()
        pat = (S @ S)
        assert st2["c_nnz"] == pat.nnz
        rr, cc, vv = C2.to_coo()
        S32 = sp.coo_matrix((np.asarray(v, np.float32).astype(np.float64), (r, c)), shape=(n, n)).tocsr()
        ref = (S32 @ S32).tocoo()
        key_ref = ref.row.astype(np.int64) * n + ref.col
        order = np.argsort(key_ref)
        np.testing.assert_array_equal(rr.astype(np.int64) * n + cc, key_ref[order])
        np.testing.assert_allclose(vv, ref.data[order], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("case", ["2cubes_sphere_like", "cage12_like", "fem_like_27pt", "rmat16"])
def test_spgemm_bench_size_properties(bmsp, oracle, case):
    """the SpGEMM bench workloads at their full BASELINE.json sizes, through size-independent properties: the three sort
    modes give bit-identical C (V15 numerics); C's pattern is scipy's pattern product; C*1 == A*(A*1) through the SpMV; the fp16
    MFMA product agrees with the fp32 one within the stated fp16 tolerance.  `fem_like_27pt` is the very workload bench.py times for
    configs[2] (fem_like(47, "27pt"): 24.7 M surviving tasks, 14 per C tile -- the regime of the direct / strip block-MAC kernels and of
    the read-back-free register sort); for it the oracle runs once at full size too (about a minute of host time): stage counters, C
    structure and the fp32 V15 values bit for bit.  Round 4: the oracle also runs at full size for `cage12_like` (configs[3]: the row-merge
    TASK-LIST mode, which small inputs do not reach by themselves) -- fp32 V15 bit for bit, and the fp16 tc_version 4 product: structure bit
    for bit, values within the stated fp16 tolerance of the oracle's exact-product accumulation.  `rmat16` (R-MAT 2^16 x 8: the
    column-window passes, 1.46e8 candidate pairs, hub block-rows of 3000 tiles of A and 7000 of C) is compared with the pipeline (sort
    mode 2) on the same handles, every array bit for bit; the oracle pins both on smaller hub inputs (test_spgemm_rowwindow_path)."""
    import scipy.sparse as sp
    from pybmsp import gen
    n, _, r, c, v = {"2cubes_sphere_like": lambda: gen.banded(101492, 8), "cage12_like": lambda: gen.cage_like(130228, 15.6),
                     "fem_like_27pt": lambda: gen.fem_like(47, "27pt"), "rmat16": lambda: gen.rmat(16, 8)}[case]()
    with_oracle = case in ("fem_like_27pt", "cage12_like")  # (rmat16: 2.5 minutes of oracle per run -- it is compared with the pipeline, bit for bit, below)
    v = np.round(np.asarray(v) * 64) / 64          # exactly representable in fp16 and fp32
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    At = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True)
    ref, st0 = bmsp.spgemm(A, At, mode=0, tc_version=5)
    ref_arrays = ref.host_arrays()
    if case == "cage12_like":
        assert st0["sort_path"] == 2 and st0["mac_variant"] != 3, st0   # the row-merge task-list mode
    if case == "rmat16":
        assert st0["sort_path"] == 3, st0                               # the column-window passes
    if with_oracle:
        oc, ost = oracle.spgemm(oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v), oracle.F32, False),
                                oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v), oracle.F32, True))
        assert (st0["task_list_size"], st0["bmp_reduction"], st0["surviving_tasks"], st0["c_blocks"], st0["c_nnz"]) == \
               (ost["task_list_size"], ost["bmp_reduction"], ost["surviving_tasks"], ost["c_blocks"], ost["c_nnz"])
        util.assert_bmsp_equal_exact(oc, *ref_arrays, np.float32)
        del oc
    for mode in ((1, 2) if case != "rmat16" else (2,)):
        Cm, st = bmsp.spgemm(A, At, mode=mode, tc_version=5)
        assert (st["task_list_size"], st["surviving_tasks"], st["c_blocks"], st["c_nnz"]) == \
               (st0["task_list_size"], st0["surviving_tasks"], st0["c_blocks"], st0["c_nnz"])
        for x, y in zip(Cm.host_arrays(), ref_arrays):
            np.testing.assert_array_equal(x, y)
        del Cm
    S = sp.coo_matrix((np.ones(r.size), (r, c)), shape=(n, n)).tocsr()
    if case != "rmat16":  # (a 1.1e9-entry pattern product on the host: the oracle above has pinned the structure)
        assert st0["c_nnz"] == (S @ S).nnz
    ones = bmsp.DeviceArray.from_host(np.ones(n, np.float32))
    a1 = bmsp.spmv(A, ones)
    y_chain = bmsp.spmv(A, a1).to_host().astype(np.float64)
    y_prod = bmsp.spmv(ref, ones).to_host().astype(np.float64)
    Sabs = sp.coo_matrix((np.abs(v), (r, c)), shape=(n, n)).tocsr()
    mag = Sabs @ (Sabs @ np.ones(n))
    assert np.all(np.abs(y_chain - y_prod) <= 1e-5 * mag + 1e-6)
    # fp16 inputs, matrix-core block-MAC (exact products, fp32 accumulation in hardware order)
    Ah = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=bmsp.F16)
    Aht = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=bmsp.F16)
    Ch, sth = bmsp.spgemm(Ah, Aht, mode=0, tc_version=4)
    kh, bh, oh, vh = Ch.host_arrays()
    np.testing.assert_array_equal(kh, ref_arrays[0]); np.testing.assert_array_equal(bh, ref_arrays[1]); np.testing.assert_array_equal(oh, ref_arrays[2])
    y_h = bmsp.spmv(Ch, ones).to_host().astype(np.float64)
    assert np.all(np.abs(y_h - y_prod) <= 2.0 ** -10 * mag + 1e-6)
    if with_oracle and case != "fem_like_27pt":
        # the matrix-core product against the oracle's own exact-product accumulation (multiplyV11..V14's numerics, SPGEMM.cu:294-417)
        och, osth = oracle.spgemm(oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v), oracle.F16, False),
                                  oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v), oracle.F16, True), exact_products=True)
        assert (sth["task_list_size"], sth["surviving_tasks"], sth["c_blocks"], sth["c_nnz"]) == \
               (osth["task_list_size"], osth["surviving_tasks"], osth["c_blocks"], osth["c_nnz"])
        np.testing.assert_array_equal(kh, och.keys); np.testing.assert_array_equal(bh, och.bmps); np.testing.assert_array_equal(oh, och.offsets)
        np.testing.assert_allclose(vh.astype(np.float64), och.values, rtol=2e-3, atol=1e-6)


def _segsort_cuts(shape, rng, n):
    """segment starts exercising each length bin of the device sort (wave <= 256, workgroup <= 4096, radix fallback)."""
    if shape == "mixed":      # lengths 1 .. ~600, plus empty-adjacent and unit segments
        extra = [5, 6, 7, 8, 100000, 100001]
        return np.unique(np.concatenate([[0], rng.integers(0, n, 3000), extra]))
    if shape == "tiny":       # lengths 1 .. 4
        return np.unique(np.concatenate([[0], np.cumsum(rng.integers(1, 5, n))]))[:-1].clip(0, n - 1)
    if shape == "block":      # bin edges 256/257, 4095/4096 and the power-of-two paddings in between
        lens = [256, 257, 4096, 4095, 511, 512, 513, 1024, 1025, 2048, 2049, 3000, 300, 2, 1, 1, 255]
        starts = np.concatenate([[0], np.cumsum(lens)])
        return np.unique(np.concatenate([starts, np.arange(starts[-1], n, 3500)]))
    if shape == "long":       # one hub segment: whole call takes the radix fallback
        return np.unique(np.concatenate([[0, 10, 50000], rng.integers(50000, n, 500)]))
    raise ValueError(shape)


@pytest.mark.parametrize("shape", ["mixed", "tiny", "block", "long"])
@pytest.mark.parametrize("val_bytes", [0, 4, 8, 16])
def test_segsort_against_gold(oracle, bmsp, val_bytes, shape):
    rng = np.random.default_rng(val_bytes)
    n = 200000
    keys = rng.integers(0, 1 << 40, n).astype(np.uint64)
    keys[: n // 2] &= np.uint64(0xFF)  # many ties: stability matters
    keys[n // 2: n // 2 + 3000] = np.uint64(0xFFFFFFFFFFFFFFFF)  # equal to the padding sentinel of the LDS network
    cuts = np.unique(_segsort_cuts(shape, rng, n)).astype(np.int64)
    cuts = cuts[cuts < n]
    pay = np.stack([np.arange(n, dtype=np.uint64), keys ^ np.uint64(0xABCDEF)], axis=1)
    gk, gv = oracle.segsort(keys, pay, cuts)
    dk = bmsp.DeviceArray.from_host(keys)
    if val_bytes == 0:
        dv = None
    elif val_bytes == 4:
        dv = bmsp.DeviceArray.from_host(pay[:, 0].astype(np.uint32))
    elif val_bytes == 8:
        dv = bmsp.DeviceArray.from_host(pay[:, 0].copy())
    else:
        dv = bmsp.DeviceArray.from_host(pay.reshape(-1))
        dv.dtype = np.dtype([("a", np.uint64), ("b", np.uint64)]); dv.n = n
    bmsp.segsort(dk, dv, bmsp.DeviceArray.from_host(cuts.astype(np.int32)))
    np.testing.assert_array_equal(dk.to_host(), gk)
    if val_bytes == 4:
        np.testing.assert_array_equal(dv.to_host(), gv[:, 0].astype(np.uint32))
    elif val_bytes == 8:
        np.testing.assert_array_equal(dv.to_host(), gv[:, 0])
    elif val_bytes == 16:
        out = dv.to_host()
        np.testing.assert_array_equal(out["a"], gv[:, 0]); np.testing.assert_array_equal(out["b"], gv[:, 1])


def test_csr_matrix_multiply_on_gpu(oracle, bmsp):
    """CSRMatrix(std::string) + multiply (include/CSRMatrix.h:13-21) against the cusp restatement."""
    path = os.path.join(MTX, "laplacian", "9pt_10x10.mtx")
    m = bmsp.CSRMatrix.from_mtx(path)
    Cm = m.multiply(m)
    nr, nc, ro, cols, vals = Cm.arrays()
    ref, _ = oracle.csr_spgemm(oracle.csr_from_coo(oracle.mtx_read(path, strict=True)), oracle.csr_from_coo(oracle.mtx_read(path, strict=True)), 1)
    np.testing.assert_array_equal(ro, ref.row_offsets)
    for i in range(nr):  # cusp leaves columns unsorted inside a row (csr_spgemm.h:153); compare as sets
        a = sorted(zip(cols[ro[i]:ro[i + 1]].tolist(), vals[ro[i]:ro[i + 1]].tolist()))
        bb = sorted(zip(ref.cols[ref.row_offsets[i]:ref.row_offsets[i + 1]].tolist(), ref.vals[ref.row_offsets[i]:ref.row_offsets[i + 1]].tolist()))
        assert a == bb
    x = (np.arange(nc) % 10).astype(np.float32)
    np.testing.assert_allclose(m.spmv(x), oracle.csr_spmv(oracle.csr_from_coo(oracle.mtx_read(path, strict=True)), x, 1), rtol=1e-6)


def test_row_panels_concat_equals_whole(oracle, bmsp):
    """the multi-GPU decomposition on one device: panel products concatenated == the whole product."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(12, 6)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    Bt = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True)
    whole, _ = bmsp.spgemm(A, Bt)
    for parts in (1, 2, 3, 8):
        bounds = bmsp.partition_rows(A, Bt, parts)
        assert bounds[0] == 0 and bounds[-1] == (n + 7) // 8 and np.all(np.diff(bounds) >= 0)
        panels, keep = [], []
        for p in range(parts):
            view = A.row_panel(bounds[p], bounds[p + 1])
            Cp, _ = bmsp.spgemm(view, Bt)
            keep.append((view, Cp))
            panels.append(Cp.device_arrays())
        cat = bmsp.concat_panels(n, n, panels)
        for x, y in zip(cat.host_arrays(), whole.host_arrays()):
            np.testing.assert_array_equal(x, y)


def test_row_panel_views_fp16_mfma_and_spmv(oracle, bmsp):
    """a panel view keeps absolute offsets into its parent's value array: the buffer-addressed kernels (MFMA block-MAC, sweep
    SpMV) must size their descriptors by the parent's extent, not by the panel's nnz."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(11, 6)
    v = np.round(v * 4) / 4  # fp16-exact inputs
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=bmsp.F16)
    Bt = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=bmsp.F16)
    for tc in (4, 3):
        whole, _ = bmsp.spgemm(A, Bt, tc_version=tc)
        bounds = bmsp.partition_rows(A, Bt, 3)
        panels, keep = [], []
        for p in range(3):
            view = A.row_panel(bounds[p], bounds[p + 1])
            Cp, _ = bmsp.spgemm(view, Bt, tc_version=tc)
            keep.append((view, Cp))
            panels.append(Cp.device_arrays())
        cat = bmsp.concat_panels(n, n, panels)
        for x, y in zip(cat.host_arrays(), whole.host_arrays()):
            np.testing.assert_array_equal(x, y)
    Af = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    x = bmsp.DeviceArray.from_host(gen.spmv_x(n, "cusp"))
    y = bmsp.spmv(Af, x).to_host()
    nbr = (n + 7) // 8
    lo, hi = nbr // 3, 2 * nbr // 3
    view = Af.row_panel(lo, hi)
    yp = bmsp.spmv(view, x).to_host()
    np.testing.assert_array_equal(yp[lo * 8: hi * 8], y[lo * 8: hi * 8])
    assert not yp[: lo * 8].any() and not yp[hi * 8:].any()


def test_spmv_launch_info_counts_the_launched_layout(bmsp):
    """bmsp_spmv_launch_info: the kernel the launcher picks and the bytes that kernel's layout must move, counted from the plan.  Sparse
    tiles -> value-stream kernel over the caches: 32 B per item + 4 B per tile (+ 2 B per tile of multi-batch items) + 6 B per fp32 value
    + x + y, well below the 24-B-per-tile format figure; dense tiles -> row-group kernel: the format figure itself."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(14, 2)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    info, li = A.info(), bmsp.spmv_launch_info(A)
    assert li["kernel"].startswith("spmv_vstream_kernel<kCached")
    nbr = (n + 7) // 8
    assert li["format_bytes"] == 24 * info["block_num"] + 4 * info["nnz"] + 4 * (nbr + 1) + 8 * n
    lo = 4 * info["block_num"] + 6 * info["nnz"] + 8 * n                   # slot words + entries + values + x + y
    assert lo < li["compulsory_bytes"] <= lo + 2 * info["block_num"] + 40 * info["block_num"] // 64 + 4096, li   # + value ends, items, carry slots
    assert li["compulsory_bytes"] < li["format_bytes"]
    y = bmsp.spmv(A, bmsp.DeviceArray.from_host(np.ones(n, np.float32))).to_host()     # the plan it counted is the plan that runs
    assert abs(float(y.sum()) - float(np.asarray(v, np.float32).astype(np.float64).sum())) <= 1e-3 * abs(float(v.sum()))
    n2, _, r2, c2, v2 = gen.banded(4096, 16)
    D = bmsp.BmSpMatrix.from_coo(n2, n2, r2, c2, v2)
    ld = bmsp.spmv_launch_info(D)
    assert ld["kernel"] == "spmv_rowgroup_kernel" and ld["compulsory_bytes"] == ld["format_bytes"]
    assert bmsp.spmv_launch_info(D, variant=1)["kernel"].startswith("spmv_blockrow_kernel")


def test_sharded_operators_through_c_abi_one_rank(oracle, bmsp):
    """bmsp_comm_init / bmsp_spgemm_sharded / bmsp_spmv_sharded over a real RCCL communicator of one rank (all a one-GPU box
    allows: RCCL refuses two ranks on one device): the unique-id rendezvous, the size all-gather, the broadcast of the panel into
    its final slice, offset re-basing and the in-place y exchange all run; results equal the unsharded operators bit for bit."""
    from pybmsp import gen
    comm = bmsp.Comm(bmsp.Comm.unique_id(), 1, 0)
    n, _, r, c, v = gen.rmat(12, 6)
    for dtype, tc in ((0, 5), (1, 4)):
        A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
        Bt = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=dtype)
        whole, st0 = bmsp.spgemm(A, Bt, tc_version=tc)
        Cs, st, sh = bmsp.spgemm_sharded(comm, A, Bt, tc_version=tc)
        assert (sh["world"], sh["rank"], sh["panel_block_row_begin"], sh["panel_block_row_end"]) == (1, 0, 0, (n + 7) // 8)
        assert sh["panel_tasks"] == st0["surviving_tasks"] and sh["exchange_bytes"] == 24 * whole.block_num + 4 * whole.nnz
        for x, y in zip(Cs.host_arrays(), whole.host_arrays()):
            np.testing.assert_array_equal(x, y)
    Af = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    x = bmsp.DeviceArray.from_host(gen.spmv_x(n, "cusp"))
    y0 = bmsp.spmv(Af, x).to_host()
    for _ in range(2):  # second call reuses the cached panel view
        y1, sh = bmsp.spmv_sharded(comm, Af, x)
        np.testing.assert_array_equal(y1.to_host(), y0)
    assert sh["exchange_bytes"] == 4 * n
    comm.free()


def _hub_matrix(gen, kind):
    """two skewed inputs for the sharded operators.  "rmat": R-MAT with a quarter of the block-rows empty and a hub row on top (panel
    sizes differ by far more than 2x).  "diag": a diagonal plus one hub row that alone carries half of the candidate tasks, so that an
    8-way split by candidate-task count leaves panels with NO block-row at all (bounds repeat) -- the empty-panel legs."""
    if kind == "rmat":
        n, _, r, c, v = gen.rmat(11, 6, seed=7)
        keep = (r < n // 4) | (r >= n // 2)
        r, c, v = r[keep], c[keep], v[keep]
    else:
        n = 2048
        r = np.arange(n, dtype=np.int32); c = r.copy(); v = 0.5 + (r % 7) / 8.0
    hub_c = np.arange(0, n, 3, dtype=r.dtype)
    r = np.concatenate([r, np.full(hub_c.size, 5, dtype=r.dtype)])
    c = np.concatenate([c, hub_c])
    v = np.concatenate([v, np.linspace(0.25, 1.0, hub_c.size)])
    key = r.astype(np.int64) * n + c
    _, first = np.unique(key, return_index=True)
    return n, r[first], c[first], np.round(v[first] * 64) / 64


@pytest.mark.parametrize("kind", ["rmat", "diag"])
@pytest.mark.parametrize("P", [2, 3, 8])
def test_sharded_operators_loopback(bmsp, P, kind):
    """every P > 1 branch of csrc/comm.hip on one device: the loopback transport computes the P panel products one after another and
    moves each panel into its final slice with a device copy; size gather -> slice starts (shard_layout) -> AddU64 re-basing for
    r > 0 -> terminal offset are the RCCL path's own lines.  Results must equal the unsharded operators bit for bit -- fp32 V15 and
    fp16 MFMA, skewed panels, empty panels ("diag", P = 8), a ragged last block-row, a rectangular product, and the SpMV's slice-only
    sweeps into NaN-poisoned vectors."""
    from pybmsp import gen
    comm = bmsp.Comm.loopback(P)
    n, r, c, v = _hub_matrix(gen, kind)
    n_rows = n - 5                                 # ragged last block-row
    keep = r < n_rows
    r, c, v = r[keep], c[keep], v[keep]
    for dtype, tc in ((0, 5), (1, 4)):
        A = bmsp.BmSpMatrix.from_coo(n_rows, n, r, c, v, dtype=dtype)
        Bt = bmsp.BmSpMatrix.from_coo(n, n_rows, c, r, v, transposed=True, dtype=dtype)   # B = A^T: a rectangular product
        whole, st0 = bmsp.spgemm(A, Bt, tc_version=tc)
        bounds = bmsp.partition_rows(A, Bt, P)
        if kind == "diag" and P == 8:
            assert np.any(np.diff(bounds) == 0), bounds     # the construction's point: panels without a single block-row
        # rounds = 0: the library's choice (4 rounds of P panels; a round's broadcasts run on a second stream while the next round multiplies);
        # 1: one panel per rank, exchanged at the end (round 3's form); 3: an odd count; gather = False: owner keeps -- the loopback
        # communicator owns every panel and returns them concatenated, without an exchange
        for rounds, gather in ((0, True), (1, True), (3, True), (0, False)):
            Cs, st, sh = bmsp.spgemm_sharded(comm, A, Bt, tc_version=tc, rounds=rounds, gather=gather)
            assert sh["world"] == P and st["surviving_tasks"] == st0["surviving_tasks"] and st["c_blocks"] == st0["c_blocks"]
            if gather:
                assert sh["exchange_bytes"] == 24 * whole.block_num + 4 * whole.nnz and sh["gathered"] == 1
                assert sh["rounds"] == (rounds or 4) and 0.0 <= sh["exchange_hidden_frac"] <= 1.0
                assert sh["exchange_exposed_us"] <= sh["exchange_us"] + 1e-6
                if rounds == 1:
                    assert sh["exchange_hidden_frac"] == 0.0  # nothing left to hide behind
            else:
                assert sh["gathered"] == 0 and sh["exchange_bytes"] == 0
            for x, y in zip(Cs.host_arrays(), whole.host_arrays()):
                np.testing.assert_array_equal(x, y)
    Af = bmsp.BmSpMatrix.from_coo(n_rows, n, r, c, v)
    x = bmsp.DeviceArray.from_host(gen.spmv_x(n, "cusp"))
    for variant in (0, 1):
        y1 = bmsp.DeviceArray.from_host(np.full(n_rows, np.nan, np.float32))
        y1, sh = bmsp.spmv_sharded(comm, Af, x, y1, variant=variant)
        ref = bmsp.spmv(Af, x, batched=bool(variant)).to_host()
        np.testing.assert_array_equal(y1.to_host(), ref)
        assert sh["exchange_bytes"] == 4 * n_rows
    comm.free()


def test_matrix_invalidate_after_value_update(oracle, bmsp):
    """the dense tile copies the block-MAC kernels read are cached on the handle and the value array is writable
    (bmsp_matrix_arrays) -- multiply, overwrite the values in place, bmsp_matrix_invalidate, multiply again: the second product must be
    the oracle's product of the NEW values (fp16 tc 4 and tc 5 both read the copies), also for a handle that borrows its arrays."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(10, 6, seed=5)
    v = np.round(v * 16) / 16
    v2 = np.round((1.0 - v) * 16) / 16 + 0.0625
    for borrowed in (False, True):
        A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=bmsp.F16)
        Bt = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=bmsp.F16)
        if borrowed:
            ka, ba, oa, va = A.device_arrays()
            A = bmsp.BmSpMatrix.from_device_arrays(n, n, ka, ba, oa, va, dtype=bmsp.F16)
        for tc in (4, 5):
            bmsp.spgemm(A, Bt, tc_version=tc)      # builds and caches the copies of the OLD values
        # new values through the public arrays (same structure): A2 / B2 are built only to get the new value arrays in tile order
        A2 = bmsp.BmSpMatrix.from_coo(n, n, r, c, v2, dtype=bmsp.F16)
        B2 = bmsp.BmSpMatrix.from_coo(n, n, r, c, v2, transposed=True, dtype=bmsp.F16)
        for dst, src in ((A, A2), (Bt, B2)):
            d, s = dst.device_arrays()[3], src.device_arrays()[3]
            bmsp.check(bmsp.lib().bmsp_memcpy_d2d(d.ptr, s.ptr, d.n * 2))
            dst.invalidate()
        oa_ = oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v2), oracle.F16, False)
        ob_ = oracle.bmsp_from_coo(oracle.Coo(n, n, r, c, v2), oracle.F16, True)
        for tc in (4, 5):
            Cm, _ = bmsp.spgemm(A, Bt, tc_version=tc)
            oc, _ = oracle.spgemm(oa_, ob_, exact_products=(tc != 5))
            util.assert_bmsp_equal_exact(oc, *Cm.host_arrays(), np.float32)   # multiples of 1/256: every path is exact


@pytest.mark.parametrize("dtype", [0, 1])
def test_row_panel_views_expand_and_compare(bmsp, dtype):
    """a NON-first panel view (offsets absolute into the parent's values, outputs sized by the view's own nnz) through every
    expansion entry point: generate_coo (host + device), CSR conversion, compare (host + device)."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(11, 6)
    v = np.round(v * 8) / 8
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
    nbr = (n + 7) // 8
    for lo, hi in ((nbr // 3, 2 * nbr // 3), (nbr - 5, nbr), (7, 7)):
        view = A.row_panel(lo, hi)
        sel = (r >= lo * 8) & (r < hi * 8)
        assert view.nnz == int(sel.sum())
        pr, pc, pv = view.to_coo()
        np.testing.assert_array_equal(pr, r[sel]); np.testing.assert_array_equal(pc, c[sel]); np.testing.assert_array_equal(pv, v[sel])
        dr, dc, dv = view.to_coo_device()
        np.testing.assert_array_equal(dr.to_host(), r[sel]); np.testing.assert_array_equal(dc.to_host(), c[sel]); np.testing.assert_array_equal(dv.to_host(), v[sel])
        ro, cc, vv = view.to_csr_device()
        np.testing.assert_array_equal(ro.to_host(), np.searchsorted(r[sel], np.arange(n + 1), side="left").astype(np.int32))
        np.testing.assert_array_equal(cc.to_host(), c[sel]); np.testing.assert_array_equal(vv.to_host(), v[sel])
        assert view.compare(r, c, v) == (0.0, 0)
        assert view.compare_device(bmsp.DeviceArray.from_host(r, np.int32), bmsp.DeviceArray.from_host(c, np.int32), bmsp.DeviceArray.from_host(v, np.float64)) == (0.0, 0)


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_device_coo_csr_conversions(oracle, bmsp, dtype):
    """bmSparse -> COO / CSR on the device and back, against scipy's CSR of the same triples."""
    import scipy.sparse as sp
    from pybmsp import gen
    n, m = 1237, 911  # ragged last block row and column, empty rows in between
    rng = np.random.default_rng(7 + dtype)
    nnz = 9000
    r = rng.integers(0, n, nnz).astype(np.int32); c = rng.integers(0, m, nnz).astype(np.int32)
    r[r % 17 == 3] = 5  # a hub row, and rows == 3 (mod 17) left empty
    v = np.round(rng.standard_normal(nnz) * 8) / 8
    ref = sp.coo_matrix((v, (r, c)), shape=(n, m)).tocsr()
    ref.sum_duplicates(); ref.sort_indices()
    for transposed in (False, True):
        M = bmsp.BmSpMatrix.from_coo(n, m, r, c, v, transposed=transposed, dtype=dtype)
        ro, cc, vv = (a.to_host() for a in M.to_csr_device())
        np.testing.assert_array_equal(ro, ref.indptr)
        np.testing.assert_array_equal(cc, ref.indices)
        np.testing.assert_array_equal(vv, ref.data)
        rr, c2, v2 = (a.to_host() for a in M.to_coo_device())
        np.testing.assert_array_equal(rr, np.repeat(np.arange(n), np.diff(ref.indptr)))
        np.testing.assert_array_equal(c2, ref.indices); np.testing.assert_array_equal(v2, ref.data)
        # back: CSR on the device -> bmSparse, identical arrays
        M2 = bmsp.BmSpMatrix.from_csr_device(n, m, *M.to_csr_device(), transposed=transposed, dtype=dtype)
        for x, y in zip(M2.host_arrays(), M.host_arrays()):
            np.testing.assert_array_equal(x, y)
    E = bmsp.BmSpMatrix.from_coo(5, 5, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), dtype=dtype)
    np.testing.assert_array_equal(E.to_csr_device()[0].to_host(), np.zeros(6, np.int32))


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 3, 4, 8, 16, 17, 64, 100])
def test_spmm_against_scipy(oracle, bmsp, dtype, k):
    """Y = A X for k vectors.  The reference has no running multi-vector path (parity unpinned against it):
    checked against scipy in float64 with the SpMV tolerance, and column by column against single-vector calls."""
    import scipy.sparse as sp
    from pybmsp import gen
    np_in = bmsp.NP_DTYPE[dtype]; np_out = bmsp.OUT_DTYPE[dtype]
    cases = [gen.rmat(11, 8), gen.banded(1003, 5)]  # hub block-rows (carry path) / ragged last block, regular rows
    for (n, _, r, c, v) in cases:
        v = np.round(v * 8) / 8
        A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
        ref = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
        rng = np.random.default_rng(k)
        X = (np.round(rng.standard_normal((n, k)) * 4) / 4).astype(np_in)
        Y = bmsp.spmm(A, bmsp.DeviceArray.from_host(X.reshape(-1)), k).to_host().reshape(n, k)
        assert Y.dtype == np_out
        want = ref @ X.astype(np.float64)
        mag = abs(ref) @ abs(X.astype(np.float64))
        tol = 1e-5 if dtype != 1 else 2e-3
        assert np.all(np.abs(Y - want) <= tol * mag + 1e-6)
        # a column of the k-wide product equals the one-vector product (inputs are exact in the accumulator, so any order agrees)
        if k in (3, 17):
            for jj in (0, k - 1):
                y1 = bmsp.spmm(A, bmsp.DeviceArray.from_host(np.ascontiguousarray(X[:, jj])), 1).to_host()
                np.testing.assert_array_equal(y1, Y[:, jj])
        # strided operands
        if k == 8:
            ldx, ldy = 11, 9
            Xs = np.zeros((n, ldx), np_in); Xs[:, :k] = X
            Ys = bmsp.spmm(A, bmsp.DeviceArray.from_host(Xs.reshape(-1)), k, ldx=ldx, ldy=ldy).to_host().reshape(n, ldy)
            np.testing.assert_array_equal(Ys[:, :k], Y)


def test_spgemm_paneled_fallback(bmsp, monkeypatch):
    """products whose candidate block pairs exceed the 32-bit task range run block-row panel after panel inside bmsp_spgemm;
    forced here on a small product: identical C, summed stage counters."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(12, 6)
    for dtype, tc in ((bmsp.F32, 5), (bmsp.F16, 4)):
        A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
        Bt = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=dtype)
        whole, st0 = bmsp.spgemm(A, Bt, tc_version=tc)
        monkeypatch.setenv("BMSP_SPGEMM_FORCE_PANELS", "1")
        try:
            paneled, st1 = bmsp.spgemm(A, Bt, tc_version=tc)
        finally:
            monkeypatch.delenv("BMSP_SPGEMM_FORCE_PANELS")
        for x, y in zip(paneled.host_arrays(), whole.host_arrays()):
            np.testing.assert_array_equal(x, y)
        for key in ("task_list_size", "bmp_reduction", "surviving_tasks", "c_blocks", "c_nnz"):
            assert st1[key] == st0[key]


def test_spgemm_beyond_32bit_candidates(bmsp):
    """R-MAT scale 22, edge factor 2: 7.9 G candidate block pairs -- more than one task list can index; checked through
    C*1 == A*(A*1) with the SpMV and through the candidate count of the fan-out."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(22, 2)
    v = np.round(np.asarray(v) * 16) / 16
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=bmsp.F16)
    At = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=bmsp.F16)
    Cm, st = bmsp.spgemm(A, At, tc_version=4)
    assert st["task_list_size"] >= 1 << 32 and st["surviving_tasks"] > 0 and st["c_blocks"] == Cm.block_num
    A32 = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    ones = bmsp.DeviceArray.from_host(np.ones(n, np.float32))
    a1 = bmsp.spmv(A32, ones)
    y_chain = bmsp.spmv(A32, a1).to_host().astype(np.float64)
    y_prod = bmsp.spmv(Cm, ones).to_host().astype(np.float64)
    absv = np.abs(v)
    rs = np.bincount(r, weights=absv, minlength=n)
    mag = np.bincount(r, weights=absv * rs[c], minlength=n)
    assert np.all(np.abs(y_chain - y_prod) <= 2.0 ** -10 * mag + 1e-4)


def test_compare_on_device(bmsp):
    """bmSpMatrix::compare with a device-resident comparand: same figure as the host comparison -- zero for an
    identical matrix, the injected error for perturbed values, `missing` for entries the comparand lacks; duplicates and extra
    comparand entries behave as on the host."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(12, 6)
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v)
    At = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True)
    Cm, _ = bmsp.spgemm(A, At)
    rr, cc, vv = Cm.to_coo()
    dev = lambda a, t: bmsp.DeviceArray.from_host(np.ascontiguousarray(a, dtype=t))
    assert Cm.compare_device(dev(rr, np.int32), dev(cc, np.int32), dev(vv, np.float64)) == (0.0, 0)
    rng = np.random.default_rng(3)
    perm = rng.permutation(rr.size)                 # the comparand need not be sorted
    v2 = vv.copy(); v2[::7] *= 1.0 + 1e-3           # relative error 1e-3 on every 7th entry
    keep = np.ones(rr.size, bool); keep[5::11] = False  # entries the comparand lacks
    extra_r = np.array([n - 1, n - 1], np.int32); extra_c = np.array([0, 0], np.int32)  # entries only the comparand has (twice)
    r3 = np.concatenate([rr[perm][keep[perm]], extra_r]); c3 = np.concatenate([cc[perm][keep[perm]], extra_c])
    v3 = np.concatenate([v2[perm][keep[perm]], [9.0, 7.0]])
    host_err, host_miss = Cm.compare(r3, c3, v3)
    dev_err, dev_miss = Cm.compare_device(dev(r3, np.int32), dev(c3, np.int32), dev(v3, np.float64))
    assert dev_miss == host_miss == int((~keep).sum())
    assert abs(dev_err - host_err) <= 1e-12 * max(1.0, host_err) and host_err > 0


def test_borrowed_arrays_multiply(oracle, bmsp):
    """bmsp_matrix_from_arrays with ownership 2 (caller keeps the arrays), every MAC kernel."""
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(10, 5)
    v = np.round(v * 4) / 4
    A = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, dtype=bmsp.F16)
    Bt = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=bmsp.F16)
    A2 = bmsp.BmSpMatrix.from_device_arrays(n, n, *A.device_arrays(), dtype=bmsp.F16)
    B2 = bmsp.BmSpMatrix.from_device_arrays(n, n, *Bt.device_arrays(), dtype=bmsp.F16, transposed=True)
    for tc in (5, 4, 2):
        ref, _ = bmsp.spgemm(A, Bt, tc_version=tc)
        got, _ = bmsp.spgemm(A2, B2, tc_version=tc)
        for x, y in zip(got.host_arrays(), ref.host_arrays()):
            np.testing.assert_array_equal(x, y)


# ---------------------------------------------------------------------------------------------------------
# drop-in executables and batch scripts (boundary: argv + stdout contract)
# ---------------------------------------------------------------------------------------------------------
def test_cpp_api_surface(tmp_path, bmsp):
    """the reference's C++ surface end to end on data/real (known answers of tests/golden/ragusa16_known.json)."""
    import subprocess
    import test_abi
    exe = str(tmp_path / "cpp_api_check")
    test_abi._build_cpp_api_check(exe)
    folder = os.path.join(MTX, "real")
    out = subprocess.run([exe, os.path.join(folder, "A_matrix.mtx"), os.path.join(folder, "B_matrix.mtx")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = {l.split()[1]: l.split()[2:] for l in out.stdout.splitlines() if l.startswith("CHECK")}
    A = bmsp.BmSpMatrix.from_mtx(os.path.join(folder, "A_matrix.mtx"))
    assert [int(x) for x in lines["A"]] == [24, 24, A.nnz, 9]
    assert [int(x) for x in lines["adopted"]] == [A.nnz, 9, 0, 0]        # the vectors were adopted (left empty)
    assert lines["compare"] == ["Final:", "0"]
    assert float(lines["spmv"][0]) == 109.0                              # sum of y = A*1
    assert float(lines["spmm"][0]) == 3 * 109.0
    assert lines["mult"] == ["9", "255", "1070", "27"]
    Cm = bmsp.CSRMatrix.from_mtx(os.path.join(folder, "A_matrix.mtx")).multiply(bmsp.CSRMatrix.from_mtx(os.path.join(folder, "B_matrix.mtx")))
    nr, nc, ro, cols, vals = Cm.arrays()
    assert [int(lines["csr"][0]), int(lines["csr"][1])] == [nr, len(cols)] and float(lines["csr"][2]) == float(np.sum(vals.astype(np.float64)))
    # CSRMatrix's host path (cusp::multiply on the host container): same product, no GPU call; y = A*1 sums to 109
    assert [int(lines["csr_host"][0]), int(lines["csr_host"][1])] == [nr, len(cols)] and float(lines["csr_host"][2]) == 1070.0 and float(lines["csr_host"][3]) == 109.0


def test_cli_executables_and_batch_scripts(tmp_path):
    import shutil
    import subprocess
    from conftest import REPO
    folder = os.path.join(MTX, "real")
    env = dict(os.environ, BMSP_PRINT_CHECKSUM="1")
    out = subprocess.run([os.path.join(REPO, "bmsparse_spmv_float"), folder, "A_matrix", "A_matrix", "0"], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == "A matrix: %s/A_matrix" % folder
    assert lines[1].startswith("Parsing mtx files / Loading matrices from disk BMSP: ") and lines[1].endswith(" μs")
    assert lines[2].strip() == "Running SpMV"
    assert lines[3].startswith("Parsing mtx files / Loading matrix and vectors: ")
    assert lines[4].startswith("bmSparse SpMV execution: ") and lines[4].endswith(" μs")
    assert lines[5] == "u checksum: 109"  # sum of y = A*1
    for seg, tc in (("0", "5"), ("1", "4")):
        out = subprocess.run([os.path.join(REPO, "bmsparse_spgemm_float"), folder, "A_matrix", "B_matrix", seg, tc, "1"], capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stderr
        text = out.stdout
        labels = ["A matrix: ", "B matrix: ", "Parsing mtx files / Loading matrices from disk BMSP: ", "T_1: ", "T_2: ", "Task list size: 27",
                  "T_3: ", "Bmp reduction: 0", "T_4: ", "T_5: ", "T_6: ", "T_9: ", "T_7: ", "Toda F: ", "bmSparse execution: ", "C blocks: 9", "C nnz: 255",
                  "C checksum: 1070"]
        pos = -1
        for lab in labels:  # same labels, same order as the reference
            nxt = text.find(lab, pos + 1)
            assert nxt > pos, (lab, text)
            pos = nxt
        if seg == "1":
            assert "Segmented sort: " in text
    # the sharded form of the same executable (one rank: BMSP_WORLD=1 needs no rendezvous file): same answer through RCCL
    out = subprocess.run([os.path.join(REPO, "bmsparse_spgemm_float"), folder, "A_matrix", "B_matrix", "0", "4", "0"], capture_output=True, text=True,
                         env=dict(env, BMSP_WORLD="1", BMSP_RANK="0"))
    assert out.returncode == 0, out.stderr
    assert "C blocks: 9" in out.stdout and "C nnz: 255" in out.stdout and "C checksum: 1070" in out.stdout and "rank 0 of 1: block-rows [0, 3), 27 tasks" in out.stdout
    # usage errors: exit code 1 and the reference's usage line
    out = subprocess.run([os.path.join(REPO, "bmsparse_spgemm_float"), folder], capture_output=True, text=True)
    assert out.returncode == 1 and "./main MatrixFolder A_Matrix B_Matrix" in out.stdout
    out = subprocess.run([os.path.join(REPO, "bmsparse_spmv_float"), folder, "missing", "missing", "0"], capture_output=True, text=True)
    assert out.returncode == 2 and "cannot open" in out.stderr
    # batch drivers: list file -> one run per line, output appended to *_out.txt
    lst = tmp_path / "lista9.txt"
    lst.write_text("A_matrix\nB_matrix\n")
    for script, outfile, needle in (("spmv_run_batch.sh", "spmv_out.txt", "bmSparse SpMV execution"), ("spgemm_run_batch.sh", "spgemm_out.txt", "C nnz: 255")):
        r = subprocess.run(["bash", os.path.join(REPO, script)], cwd=tmp_path, capture_output=True, text=True,
                           env=dict(os.environ, folder=folder, list=str(lst)))
        assert r.returncode == 0, r.stderr
        assert r.stdout.count("Working on") == 2
        assert (tmp_path / outfile).read_text().count(needle) == 2


def test_binary_cache_roundtrip(oracle, bmsp, tmp_path):
    from pybmsp import gen
    n, _, r, c, v = gen.rmat(10, 5)
    for dtype in (0, 1, 2):
        for transposed in (False, True):
            m = bmsp.BmSpMatrix.from_coo(n, n, r, c, v, transposed=transposed, dtype=dtype)
            path = str(tmp_path / ("m_%d_%d.bmsp" % (dtype, transposed)))
            m.save(path)
            m2 = bmsp.BmSpMatrix.load(path)
            assert m.info() == m2.info()
            for x, y in zip(m.host_arrays(), m2.host_arrays()):
                np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
            np.testing.assert_array_equal(m.block_row_ptr(), m2.block_row_ptr())
    with pytest.raises(bmsp.BmspError):
        bmsp.BmSpMatrix.load(os.path.join(MTX, "real", "A_matrix.mtx"))  # not a cache file
    open(str(tmp_path / "trunc.bmsp"), "wb").write(open(path, "rb").read()[:100])
    with pytest.raises(bmsp.BmspError):
        bmsp.BmSpMatrix.load(str(tmp_path / "trunc.bmsp"))
