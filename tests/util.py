"""shared helpers for the parity tests."""
import glob
import os
import numpy as np
from conftest import MTX


def all_fixture_mtx(include_pattern=False):
    files = sorted(glob.glob(os.path.join(MTX, "*", "*.mtx")))
    if not include_pattern:
        files = [f for f in files if "pattern" not in f and "complex" not in f]
    return files


def dok_from_coo(rows, cols, vals):
    d = {}
    for r, c, v in zip(rows.tolist(), cols.tolist(), vals.tolist()):
        d[(r, c)] = d.get((r, c), 0.0) + v
    return d


def scipy_csr(num_rows, num_cols, rows, cols, vals, dtype=np.float64):
    import scipy.sparse as sp
    return sp.coo_matrix((np.asarray(vals, dtype=dtype), (rows, cols)), shape=(num_rows, num_cols)).tocsr()


def bmsp_host_to_dok(num_rows, num_cols, keys, bmps, offsets, values, transposed=False):
    """expand the four bmSparse arrays to {(r,c): v} straight from the format definition."""
    out = {}
    for b in range(len(keys)):
        brow, bcol = int(keys[b]) >> 32, int(keys[b]) & 0xFFFFFFFF
        bmp, off, k = int(bmps[b]), int(offsets[b]), 0
        for p in range(64):
            if bmp >> (63 - p) & 1:
                hi, lo = p // 8, p % 8
                r, c = (brow * 8 + lo, bcol * 8 + hi) if transposed else (brow * 8 + hi, bcol * 8 + lo)
                out[(r, c)] = float(values[off + k])
                k += 1
    return out


def assert_bmsp_equal_exact(oracle_m, keys, bmps, offsets, values, np_dtype):
    """bit-exact comparison of the product's four arrays with the oracle's matrix."""
    nb = oracle_m.block_num
    assert len(keys) == nb, (len(keys), nb)
    np.testing.assert_array_equal(np.asarray(keys, dtype=np.uint64), oracle_m.keys)
    np.testing.assert_array_equal(np.asarray(bmps, dtype=np.uint64)[:nb], oracle_m.bmps)
    np.testing.assert_array_equal(np.asarray(offsets, dtype=np.uint64)[:nb + 1], oracle_m.offsets)
    ref = oracle_m.values.astype(np_dtype)
    got = np.asarray(values)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got.view(np.uint8), ref.view(np.uint8))
