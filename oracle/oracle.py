"""ctypes loader for the CPU parity oracle (oracle/libbmsp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (bmsparse-spgemm-spmv_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32, F16, F64 = 0, 1, 2
_NP = {F32: np.float32, F16: np.float16, F64: np.float64}


class _Bmsp(C.Structure):
    _fields_ = [("num_rows", C.c_int), ("num_cols", C.c_int), ("nnz", C.c_int64), ("block_num", C.c_int64),
                ("dtype", C.c_int), ("transposed", C.c_int),
                ("keys", C.POINTER(C.c_uint64)), ("bmps", C.POINTER(C.c_uint64)), ("offsets", C.POINTER(C.c_uint64)),
                ("values", C.POINTER(C.c_double))]


class _Coo(C.Structure):
    _fields_ = [("num_rows", C.c_int), ("num_cols", C.c_int), ("nnz", C.c_int64),
                ("rows", C.POINTER(C.c_int)), ("cols", C.POINTER(C.c_int)), ("vals", C.POINTER(C.c_double))]


class _Csr(C.Structure):
    _fields_ = [("num_rows", C.c_int), ("num_cols", C.c_int), ("nnz", C.c_int64),
                ("row_offsets", C.POINTER(C.c_int)), ("cols", C.POINTER(C.c_int)), ("vals", C.POINTER(C.c_float))]


class _Stats(C.Structure):
    _fields_ = [("task_list_size", C.c_int64), ("bmp_reduction", C.c_int64), ("surviving_tasks", C.c_int64),
                ("c_blocks", C.c_int64), ("c_nnz", C.c_int64), ("scalar_products", C.c_int64)]


def build(force=False):
    so = os.path.join(_HERE, "libbmsp_oracle.so")
    src = os.path.join(_HERE, "bmsp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libbmsp_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_f64_to_f16_bits.restype = C.c_uint16
        L.orc_f64_to_f16_bits.argtypes = [C.c_double]
        L.orc_f16_bits_to_f64.restype = C.c_double
        L.orc_f16_bits_to_f64.argtypes = [C.c_uint16]
        L.orc_round_to_dtype.restype = C.c_double
        L.orc_round_to_dtype.argtypes = [C.c_double, C.c_int]
        L.orc_bmp_product.restype = C.c_uint64
        L.orc_bmp_product.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_bmp_product_empty.restype = C.c_int
        L.orc_bmp_product_empty.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_bmsp_compare.restype = C.c_double
        L.orc_spmv_f32.argtypes = [C.POINTER(_Bmsp), C.c_void_p, C.c_void_p]
        L.orc_csr_spmv.argtypes = [C.POINTER(_Csr), C.c_void_p, C.c_void_p, C.c_int]
        L.orc_segsort_u64_kv.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    return _LIB


class Coo:
    """host COO (0-based)."""

    def __init__(self, num_rows, num_cols, rows, cols, vals):
        self.num_rows, self.num_cols = int(num_rows), int(num_cols)
        self.rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.cols = np.ascontiguousarray(cols, dtype=np.int32)
        self.vals = np.ascontiguousarray(vals, dtype=np.float64)
        assert self.rows.shape == self.cols.shape == self.vals.shape

    @property
    def nnz(self):
        return int(self.rows.shape[0])

    def _c(self):
        return _Coo(self.num_rows, self.num_cols, self.nnz,
                    self.rows.ctypes.data_as(C.POINTER(C.c_int)), self.cols.ctypes.data_as(C.POINTER(C.c_int)),
                    self.vals.ctypes.data_as(C.POINTER(C.c_double)))

    @staticmethod
    def _from_c(c):
        n = c.nnz
        out = Coo(c.num_rows, c.num_cols,
                  np.ctypeslib.as_array(c.rows, (n,)).copy() if n else np.zeros(0, np.int32),
                  np.ctypeslib.as_array(c.cols, (n,)).copy() if n else np.zeros(0, np.int32),
                  np.ctypeslib.as_array(c.vals, (n,)).copy() if n else np.zeros(0, np.float64))
        return out


class Bmsp:
    """host bmSparse matrix as the oracle builds it (values held as float64, exactly representable in dtype)."""

    def __init__(self, num_rows, num_cols, dtype, transposed, keys, bmps, offsets, values):
        self.num_rows, self.num_cols, self.dtype, self.transposed = int(num_rows), int(num_cols), dtype, int(transposed)
        self.keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self.bmps = np.ascontiguousarray(bmps, dtype=np.uint64)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)  # block_num + 1 entries
        self.values = np.ascontiguousarray(values, dtype=np.float64)

    @property
    def block_num(self):
        return int(self.keys.shape[0])

    @property
    def nnz(self):
        return int(self.values.shape[0])

    def values_as_dtype(self):
        return self.values.astype(_NP[self.dtype])

    def _c(self):
        return _Bmsp(self.num_rows, self.num_cols, self.nnz, self.block_num, self.dtype, self.transposed,
                     self.keys.ctypes.data_as(C.POINTER(C.c_uint64)), self.bmps.ctypes.data_as(C.POINTER(C.c_uint64)),
                     self.offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                     self.values.ctypes.data_as(C.POINTER(C.c_double)))

    @staticmethod
    def _from_c(c):
        nb, nz = c.block_num, c.nnz
        arr = lambda p, n, dt: (np.ctypeslib.as_array(p, (n,)).copy() if n else np.zeros(0, dt))
        return Bmsp(c.num_rows, c.num_cols, c.dtype, c.transposed, arr(c.keys, nb, np.uint64), arr(c.bmps, nb, np.uint64),
                    np.ctypeslib.as_array(c.offsets, (nb + 1,)).copy(), arr(c.values, nz, np.float64))


def mtx_read(path, strict=False):
    c = _Coo()
    rc = lib().orc_mtx_read(path.encode(), int(strict), C.byref(c))
    if rc != 0:
        raise IOError("orc_mtx_read(%s) failed: %d" % (path, rc))
    out = Coo._from_c(c)
    lib().orc_coo_free(C.byref(c))
    return out


def bmsp_from_coo(coo, dtype=F32, transposed=False):
    c = _Bmsp()
    cc = coo._c()
    rc = lib().orc_bmsp_from_coo(C.byref(cc), dtype, int(transposed), C.byref(c))
    assert rc == 0
    out = Bmsp._from_c(c)
    lib().orc_bmsp_free(C.byref(c))
    return out


def bmsp_to_coo(m):
    c = _Coo()
    mc = m._c()
    rc = lib().orc_bmsp_to_coo(C.byref(mc), C.byref(c))
    assert rc == 0
    out = Coo._from_c(c)
    lib().orc_coo_free(C.byref(c))
    return out


def bmsp_compare(m, coo):
    mc, cc = m._c(), coo._c()
    return float(lib().orc_bmsp_compare(C.byref(mc), C.byref(cc)))


def spmv_f32(A, v):
    v = np.ascontiguousarray(v, dtype=np.float32)
    assert v.shape[0] >= A.num_cols
    u = np.zeros(A.num_rows, dtype=np.float32)
    ac = A._c()
    rc = lib().orc_spmv_f32(C.byref(ac), v.ctypes.data, u.ctypes.data)
    assert rc == 0
    return u


def spgemm(A, B, exact_products=False):
    """returns (C: Bmsp with fp32 values, stats dict)."""
    c = _Bmsp()
    st = _Stats()
    ac, bc = A._c(), B._c()
    rc = lib().orc_spgemm(C.byref(ac), C.byref(bc), int(exact_products), C.byref(c), C.byref(st))
    if rc != 0:
        raise ValueError("orc_spgemm failed: %d" % rc)
    out = Bmsp._from_c(c)
    lib().orc_bmsp_free(C.byref(c))
    return out, {k: int(getattr(st, k)) for k, _ in _Stats._fields_}


def segsort(keys, vals2, segs):
    keys = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    vals2 = np.ascontiguousarray(vals2, dtype=np.uint64).copy()
    segs = np.ascontiguousarray(segs, dtype=np.int64)
    rc = lib().orc_segsort_u64_kv(keys.ctypes.data, vals2.ctypes.data, keys.shape[0], segs.ctypes.data, segs.shape[0])
    assert rc == 0
    return keys, vals2


class Csr:
    def __init__(self, num_rows, num_cols, row_offsets, cols, vals):
        self.num_rows, self.num_cols = int(num_rows), int(num_cols)
        self.row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int32)
        self.cols = np.ascontiguousarray(cols, dtype=np.int32)
        self.vals = np.ascontiguousarray(vals, dtype=np.float32)

    @property
    def nnz(self):
        return int(self.cols.shape[0])

    def _c(self):
        return _Csr(self.num_rows, self.num_cols, self.nnz, self.row_offsets.ctypes.data_as(C.POINTER(C.c_int)),
                    self.cols.ctypes.data_as(C.POINTER(C.c_int)), self.vals.ctypes.data_as(C.POINTER(C.c_float)))


def csr_from_coo(coo):
    """coo must be sorted by (row, col) (what the CUSP reader hands over)."""
    c = _Csr()
    cc = coo._c()
    rc = lib().orc_csr_from_coo(C.byref(cc), C.byref(c))
    assert rc == 0, "COO not row-sorted"
    n = c.nnz
    out = Csr(c.num_rows, c.num_cols, np.ctypeslib.as_array(c.row_offsets, (c.num_rows + 1,)).copy(),
              np.ctypeslib.as_array(c.cols, (n,)).copy() if n else np.zeros(0, np.int32),
              np.ctypeslib.as_array(c.vals, (n,)).copy() if n else np.zeros(0, np.float32))
    lib().orc_csr_free(C.byref(c))
    return out


def csr_spmv(A, x, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros(A.num_rows, dtype=np.float32)
    ac = A._c()
    lib().orc_csr_spmv(C.byref(ac), x.ctypes.data, y.ctypes.data, int(threads))
    return y


def csr_spgemm(A, B, threads=1):
    c = _Csr()
    prods = C.c_int64(0)
    ac, bc = A._c(), B._c()
    rc = lib().orc_csr_spgemm(C.byref(ac), C.byref(bc), C.byref(c), int(threads), C.byref(prods))
    assert rc == 0
    n = c.nnz
    out = Csr(c.num_rows, c.num_cols, np.ctypeslib.as_array(c.row_offsets, (c.num_rows + 1,)).copy(),
              np.ctypeslib.as_array(c.cols, (n,)).copy() if n else np.zeros(0, np.int32),
              np.ctypeslib.as_array(c.vals, (n,)).copy() if n else np.zeros(0, np.float32))
    lib().orc_csr_free(C.byref(c))
    return out, int(prods.value)


def max_threads():
    return int(lib().orc_max_threads())
