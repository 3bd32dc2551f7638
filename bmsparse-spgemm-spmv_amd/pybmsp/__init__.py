"""pybmsp -- thin ctypes binding of libbmsp.so (include/bmsp.h), used by tests/ and bench.py.

Plumbing only: every operation is a call through the C ABI into the HIP kernels.  There is no CPU fallback;
if the shared library is missing or a call fails, an exception is raised.

If PyTorch is used in the same process, import torch BEFORE this module so that both share one HIP runtime
(torch/lib/libamdhip64.so and /opt/rocm/lib/libamdhip64.so carry the same SONAME).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BMSP_LIB_PATH") or os.path.join(_PKG, "lib", "libbmsp.so")  # (the override: A/B runs of two builds in one GPU session)

F32, F16, F64 = 0, 1, 2
NP_DTYPE = {F32: np.float32, F16: np.float16, F64: np.float64}
OUT_DTYPE = {F32: np.float32, F16: np.float32, F64: np.float64}
SORT_AUTO, SORT_SEGMENTED, SORT_GLOBAL = 0, 1, 2
SPMV_DEFAULT, SPMV_BATCHED = 0, 1

# every symbol include/bmsp.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "bmsp_last_error", "bmsp_version", "bmsp_device_count", "bmsp_set_device", "bmsp_malloc", "bmsp_free",
    "bmsp_memcpy_h2d", "bmsp_memcpy_d2h", "bmsp_memcpy_d2d", "bmsp_memset", "bmsp_synchronize", "bmsp_trim_pool",
    "bmsp_event_create", "bmsp_event_record", "bmsp_event_elapsed_ms", "bmsp_event_destroy",
    "bmsp_matrix_from_mtx", "bmsp_matrix_from_coo", "bmsp_matrix_from_coo_device", "bmsp_matrix_from_arrays",
    "bmsp_matrix_save", "bmsp_matrix_load", "bmsp_matrix_free", "bmsp_matrix_prepare", "bmsp_matrix_invalidate", "bmsp_matrix_info", "bmsp_matrix_arrays", "bmsp_matrix_block_row_ptr",
    "bmsp_matrix_to_coo_host", "bmsp_matrix_to_coo_device", "bmsp_matrix_to_csr_device", "bmsp_matrix_from_csr_device", "bmsp_matrix_compare", "bmsp_matrix_compare_device", "bmsp_spmv", "bmsp_spmv_launch_info", "bmsp_spmm", "bmsp_spgemm", "bmsp_spgemm_symbolic", "bmsp_spgemm_numeric", "bmsp_selftest_mfma_layout", "bmsp_selftest_mfma_f32_chain", "bmsp_selftest_tile_product", "bmsp_selftest_mfma_f32_cancel", "bmsp_segsort_u64",
    "bmsp_partition_rows", "bmsp_matrix_row_panel", "bmsp_matrix_concat_panels",
    "bmsp_comm_unique_id", "bmsp_comm_init", "bmsp_comm_init_from_env", "bmsp_comm_init_loopback", "bmsp_shard_layout", "bmsp_shard_row_slices", "bmsp_comm_info", "bmsp_comm_free", "bmsp_spgemm_sharded", "bmsp_spgemm_sharded_ex", "bmsp_spmv_sharded",
    "bmsp_csr_from_mtx", "bmsp_csr_from_arrays", "bmsp_csr_info", "bmsp_csr_arrays", "bmsp_csr_multiply",
    "bmsp_csr_spmv", "bmsp_csr_multiply_host", "bmsp_csr_spmv_host", "bmsp_csr_free",
]


class BmspError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("bmsp status %d: %s" % (status, msg))
        self.status = status


class SpgemmStats(C.Structure):
    _fields_ = [("task_list_size", C.c_int64), ("bmp_reduction", C.c_int64), ("surviving_tasks", C.c_int64),
                ("c_blocks", C.c_int64), ("c_nnz", C.c_int64), ("t_us", C.c_double * 10),
                ("sort_path", C.c_int), ("mac_kernel", C.c_int), ("mac_variant", C.c_int), ("sort_long", C.c_int)]

    def as_dict(self):
        d = {k: getattr(self, k) for k in ("task_list_size", "bmp_reduction", "surviving_tasks", "c_blocks", "c_nnz",
                                           "sort_path", "mac_kernel", "mac_variant", "sort_long")}
        d["t_us"] = list(self.t_us)
        return d


class ShardStats(C.Structure):
    _fields_ = [("world", C.c_int), ("rank", C.c_int), ("panel_block_row_begin", C.c_int64), ("panel_block_row_end", C.c_int64),
                ("panel_tasks", C.c_int64), ("exchange_bytes", C.c_int64), ("exchange_us", C.c_double), ("exchange_exposed_us", C.c_double),
                ("exchange_hidden_frac", C.c_double), ("rounds", C.c_int), ("gathered", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def build_library():
    subprocess.check_call(["make", "-s", "-C", _PKG, "lib"])
    return LIB_PATH


def lib():
    """loads libbmsp.so; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbmsp.so not built: run `make -C %s lib` (or __graft_entry__.build())" % _PKG)
        L = C.CDLL(LIB_PATH)
        L.bmsp_last_error.restype = C.c_char_p
        L.bmsp_version.restype = C.c_char_p
        p, i, i64, vp = C.POINTER, C.c_int, C.c_int64, C.c_void_p
        L.bmsp_malloc.argtypes = [p(vp), C.c_size_t]
        L.bmsp_free.argtypes = [vp]
        L.bmsp_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.bmsp_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.bmsp_memcpy_d2d.argtypes = [vp, vp, C.c_size_t]
        L.bmsp_memset.argtypes = [vp, i, C.c_size_t]
        L.bmsp_event_create.argtypes = [p(vp)]
        L.bmsp_event_record.argtypes = [vp, vp]
        L.bmsp_event_elapsed_ms.argtypes = [vp, vp, p(C.c_float)]
        L.bmsp_event_destroy.argtypes = [vp]
        L.bmsp_matrix_from_mtx.argtypes = [C.c_char_p, i, i, p(vp)]
        L.bmsp_matrix_from_coo.argtypes = [i, i, i64, vp, vp, vp, i, i, p(vp)]
        L.bmsp_matrix_from_coo_device.argtypes = [i, i, i64, vp, vp, vp, i, i, vp, p(vp)]
        L.bmsp_matrix_from_arrays.argtypes = [i, i, i64, i64, vp, vp, vp, vp, i, i, i, p(vp)]
        L.bmsp_matrix_save.argtypes = [vp, C.c_char_p]
        L.bmsp_matrix_load.argtypes = [C.c_char_p, p(vp)]
        L.bmsp_matrix_free.argtypes = [vp]
        L.bmsp_matrix_prepare.argtypes = [vp, i, vp]
        L.bmsp_matrix_info.argtypes = [vp, p(i), p(i), p(i64), p(i64), p(i), p(i)]
        L.bmsp_matrix_arrays.argtypes = [vp, p(vp), p(vp), p(vp), p(vp)]
        L.bmsp_matrix_block_row_ptr.argtypes = [vp, p(vp), p(i64)]
        L.bmsp_matrix_to_coo_host.argtypes = [vp, vp, vp, vp]
        L.bmsp_matrix_to_coo_device.argtypes = [vp, vp, vp, vp, vp]
        L.bmsp_matrix_to_csr_device.argtypes = [vp, vp, vp, vp, vp]
        L.bmsp_matrix_from_csr_device.argtypes = [i, i, i64, vp, vp, vp, i, i, vp, p(vp)]
        L.bmsp_matrix_compare_device.argtypes = [vp, i64, vp, vp, vp, p(C.c_double), p(i64), vp]
        L.bmsp_matrix_compare.argtypes = [vp, i64, vp, vp, vp, p(C.c_double), p(i64)]
        L.bmsp_spmv.argtypes = [vp, vp, vp, i, vp]
        L.bmsp_spmm.argtypes = [vp, vp, i64, vp, i64, i, vp]
        L.bmsp_spmv_launch_info.argtypes = [vp, i, C.c_char_p, C.c_size_t, p(i64), p(i64)]
        L.bmsp_spgemm.argtypes = [vp, vp, p(vp), i, i, i, vp, p(SpgemmStats)]
        L.bmsp_spgemm_symbolic.argtypes = [vp, vp, p(vp), i, i, vp, p(SpgemmStats)]
        L.bmsp_spgemm_numeric.argtypes = [vp, vp, vp, i, vp, p(SpgemmStats)]
        L.bmsp_selftest_mfma_layout.argtypes = [p(i)]
        L.bmsp_selftest_mfma_f32_chain.argtypes = [p(i)]
        L.bmsp_selftest_tile_product.argtypes = [p(i)]
        L.bmsp_selftest_mfma_f32_cancel.argtypes = [p(i), p(i)]
        L.bmsp_segsort_u64.argtypes = [vp, vp, i, i64, vp, i64, vp]
        L.bmsp_partition_rows.argtypes = [vp, vp, i, vp]
        L.bmsp_matrix_row_panel.argtypes = [vp, i64, i64, p(vp)]
        L.bmsp_matrix_concat_panels.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, i, p(vp)]
        L.bmsp_comm_unique_id.argtypes = [vp]
        L.bmsp_comm_init.argtypes = [vp, i, i, p(vp)]
        L.bmsp_comm_init_from_env.argtypes = [p(vp)]
        L.bmsp_comm_init_loopback.argtypes = [i, p(vp)]
        L.bmsp_shard_layout.argtypes = [i, vp, vp, vp, vp]
        L.bmsp_shard_row_slices.argtypes = [i, i, vp, vp, vp]
        L.bmsp_matrix_invalidate.argtypes = [vp, i]
        L.bmsp_comm_info.argtypes = [vp, p(i), p(i)]
        L.bmsp_comm_free.argtypes = [vp]
        L.bmsp_spgemm_sharded.argtypes = [vp, vp, vp, p(vp), i, i, i, vp, p(SpgemmStats), p(ShardStats)]
        L.bmsp_spgemm_sharded_ex.argtypes = [vp, vp, vp, p(vp), i, i, i, vp, p(SpgemmStats), p(ShardStats), i, i]
        L.bmsp_spmv_sharded.argtypes = [vp, vp, vp, vp, i, vp, p(ShardStats)]
        L.bmsp_csr_from_mtx.argtypes = [C.c_char_p, p(vp)]
        L.bmsp_csr_from_arrays.argtypes = [i, i, i64, vp, vp, vp, p(vp)]
        L.bmsp_csr_info.argtypes = [vp, p(i), p(i), p(i64)]
        L.bmsp_csr_arrays.argtypes = [vp, p(vp), p(vp), p(vp)]
        L.bmsp_csr_multiply.argtypes = [vp, vp, p(vp)]
        L.bmsp_csr_spmv.argtypes = [vp, vp, vp]
        L.bmsp_csr_multiply_host.argtypes = [vp, vp, p(vp), i]
        L.bmsp_csr_spmv_host.argtypes = [vp, vp, vp, i]
        L.bmsp_csr_free.argtypes = [vp]
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise BmspError(status, lib().bmsp_last_error().decode(errors="replace"))


# ---- device buffers ---------------------------------------------------------------------------------------
class DeviceArray:
    """a typed device buffer from the library's pool."""

    def __init__(self, n, dtype, ptr=None, owner=None):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self._owned = ptr is None
        self._owner = owner  # keeps a parent (matrix) alive for borrowed pointers
        if ptr is None:
            q = C.c_void_p()
            check(lib().bmsp_malloc(C.byref(q), max(1, self.n) * self.dtype.itemsize))
            self.ptr = q.value
        else:
            self.ptr = int(ptr) if ptr else 0

    @staticmethod
    def from_host(arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        d = DeviceArray(a.size, a.dtype)
        if a.size:
            check(lib().bmsp_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def to_host(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            check(lib().bmsp_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self):
        if self.n:
            check(lib().bmsp_memset(self.ptr, 0, self.n * self.dtype.itemsize))

    def release(self):
        """gives the pointer away (ownership moves to a matrix that adopts it)."""
        self._owned = False
        return self.ptr

    def __del__(self):
        try:
            if self._owned and self.ptr:
                lib().bmsp_free(self.ptr)
        except Exception:
            pass


class Event:
    """hipEvent on the operators' stream (device-side timing)."""

    def __init__(self):
        e = C.c_void_p()
        check(lib().bmsp_event_create(C.byref(e)))
        self.e = e.value

    def record(self, stream=None):
        check(lib().bmsp_event_record(self.e, stream))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(lib().bmsp_event_elapsed_ms(self.e, stop.e, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().bmsp_event_destroy(self.e)
        except Exception:
            pass


def synchronize():
    check(lib().bmsp_synchronize())


def device_count():
    n = C.c_int()
    check(lib().bmsp_device_count(C.byref(n)))
    return n.value


def set_device(d):
    check(lib().bmsp_set_device(int(d)))


# ---- the container ------------------------------------------------------------------------------------------
class BmSpMatrix:
    """mirror of the reference's bmSpMatrix<T> (include/bmSpMatrix.h:20-40) over the C ABI."""

    def __init__(self, handle, parent=None):
        self.h = handle
        self._parent = parent

    # bmSpMatrix(std::string path, bool transpose)
    @staticmethod
    def from_mtx(path, transposed=False, dtype=F32):
        h = C.c_void_p()
        check(lib().bmsp_matrix_from_mtx(os.fsencode(path), int(bool(transposed)), dtype, C.byref(h)))
        return BmSpMatrix(h.value)

    @staticmethod
    def from_coo(num_rows, num_cols, rows, cols, vals, transposed=False, dtype=F32):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        assert rows.shape == cols.shape == vals.shape
        h = C.c_void_p()
        check(lib().bmsp_matrix_from_coo(int(num_rows), int(num_cols), rows.size, rows.ctypes.data, cols.ctypes.data,
                                         vals.ctypes.data, int(bool(transposed)), dtype, C.byref(h)))
        return BmSpMatrix(h.value)

    # bmSpMatrix(int num_rows, int num_cols, int block_num, keys&, bmps&, offsets&, values&)
    @staticmethod
    def from_arrays(num_rows, num_cols, keys, bmps, offsets, values, dtype=F32, transposed=False):
        """host arrays are copied to the device and adopted by the new matrix."""
        k = DeviceArray.from_host(keys, np.uint64)
        b = DeviceArray.from_host(bmps, np.uint64)
        o = DeviceArray.from_host(offsets, np.uint64)
        v = DeviceArray.from_host(values, NP_DTYPE[dtype])
        h = C.c_void_p()
        check(lib().bmsp_matrix_from_arrays(int(num_rows), int(num_cols), k.n, v.n, k.ptr, b.ptr, o.ptr, v.ptr, dtype,
                                            int(bool(transposed)), 1, C.byref(h)))
        for d in (k, b, o, v):
            d.release()
        return BmSpMatrix(h.value)

    @staticmethod
    def from_device_arrays(num_rows, num_cols, keys, bmps, offsets, values, dtype=F32, transposed=False):
        """borrows DeviceArrays (ownership 2): `offsets` must hold block_num+1 entries; the arrays must outlive the matrix."""
        h = C.c_void_p()
        check(lib().bmsp_matrix_from_arrays(int(num_rows), int(num_cols), keys.n, values.n, keys.ptr, bmps.ptr, offsets.ptr, values.ptr, dtype,
                                            int(bool(transposed)), 2, C.byref(h)))
        return BmSpMatrix(h.value, parent=(keys, bmps, offsets, values))

    def prepare(self, what=3, stream=None):
        """builds the cached sweep plan (1) / block-MAC operand records (2) ahead of the first product."""
        check(lib().bmsp_matrix_prepare(self.h, int(what), stream))
        return self

    def invalidate(self, structure_changed=False):
        """after writing the arrays in place: drops the cached derived structures (bmsp_matrix_invalidate)."""
        check(lib().bmsp_matrix_invalidate(self.h, int(bool(structure_changed))))
        return self

    def clone(self):
        """an independent device copy of the four arrays (bmsp_matrix_from_arrays, ownership 0)."""
        i = self.info()
        k, b, o, v = self.device_arrays()
        h = C.c_void_p()
        check(lib().bmsp_matrix_from_arrays(i["num_rows"], i["num_cols"], i["block_num"], i["nnz"], k.ptr, b.ptr, o.ptr, v.ptr, i["dtype"],
                                            i["transposed"], 0, C.byref(h)))
        return BmSpMatrix(h.value)

    def save(self, path):
        check(lib().bmsp_matrix_save(self.h, os.fsencode(path)))

    @staticmethod
    def load(path):
        h = C.c_void_p()
        check(lib().bmsp_matrix_load(os.fsencode(path), C.byref(h)))
        return BmSpMatrix(h.value)

    def info(self):
        nr, nc, tr, dt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        nnz, nb = C.c_int64(), C.c_int64()
        check(lib().bmsp_matrix_info(self.h, C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(nb), C.byref(dt), C.byref(tr)))
        return dict(num_rows=nr.value, num_cols=nc.value, nnz=nnz.value, block_num=nb.value, dtype=dt.value,
                    transposed=tr.value)

    num_rows = property(lambda s: s.info()["num_rows"])
    num_cols = property(lambda s: s.info()["num_cols"])
    nnz = property(lambda s: s.info()["nnz"])
    block_num = property(lambda s: s.info()["block_num"])
    dtype = property(lambda s: s.info()["dtype"])

    def device_arrays(self):
        """(keys, bmps, offsets, values) as borrowed DeviceArrays; offsets has block_num+1 entries."""
        k, b, o, v = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().bmsp_matrix_arrays(self.h, C.byref(k), C.byref(b), C.byref(o), C.byref(v)))
        i = self.info()
        return (DeviceArray(i["block_num"], np.uint64, k.value or 0, self), DeviceArray(i["block_num"], np.uint64, b.value or 0, self),
                DeviceArray(i["block_num"] + 1, np.uint64, o.value or 0, self),
                DeviceArray(i["nnz"], NP_DTYPE[i["dtype"]], v.value or 0, self))

    def host_arrays(self):
        return tuple(a.to_host() for a in self.device_arrays())

    def block_row_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        check(lib().bmsp_matrix_block_row_ptr(self.h, C.byref(p), C.byref(n)))
        return DeviceArray(n.value + 1, np.uint32, p.value, self).to_host()

    # generate_coo()
    def to_coo(self):
        n = self.nnz
        r, c, v = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        check(lib().bmsp_matrix_to_coo_host(self.h, r.ctypes.data, c.ctypes.data, v.ctypes.data))
        return r, c, v

    def to_coo_device(self, stream=None):
        """(rows int32, cols int32, vals float64) DeviceArrays, sorted by (row, col)."""
        n = self.nnz
        r, c, v = DeviceArray(n, np.int32), DeviceArray(n, np.int32), DeviceArray(n, np.float64)
        check(lib().bmsp_matrix_to_coo_device(self.h, r.ptr, c.ptr, v.ptr, stream))
        return r, c, v

    def to_csr_device(self, stream=None):
        """(row_offsets int32 [num_rows+1], cols int32, vals float64) DeviceArrays."""
        i = self.info()
        ro, c, v = DeviceArray(i["num_rows"] + 1, np.int32), DeviceArray(i["nnz"], np.int32), DeviceArray(i["nnz"], np.float64)
        check(lib().bmsp_matrix_to_csr_device(self.h, ro.ptr, c.ptr, v.ptr, stream))
        return ro, c, v

    @staticmethod
    def from_csr_device(num_rows, num_cols, row_offsets, cols, vals, transposed=False, dtype=F32, stream=None):
        h = C.c_void_p()
        check(lib().bmsp_matrix_from_csr_device(int(num_rows), int(num_cols), cols.n, row_offsets.ptr, cols.ptr, vals.ptr,
                                                int(bool(transposed)), dtype, stream, C.byref(h)))
        return BmSpMatrix(h.value)

    # compare(coo)
    def compare(self, rows, cols, vals):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        err, miss = C.c_double(), C.c_int64()
        check(lib().bmsp_matrix_compare(self.h, rows.size, rows.ctypes.data, cols.ctypes.data, vals.ctypes.data,
                                        C.byref(err), C.byref(miss)))
        return err.value, miss.value

    def compare_device(self, rows, cols, vals, stream=None):
        """rows / cols int32, vals float64 DeviceArrays."""
        err, miss = C.c_double(), C.c_int64()
        check(lib().bmsp_matrix_compare_device(self.h, rows.n, rows.ptr, cols.ptr, vals.ptr, C.byref(err), C.byref(miss), stream))
        return err.value, miss.value

    def row_panel(self, brow_begin, brow_end):
        h = C.c_void_p()
        check(lib().bmsp_matrix_row_panel(self.h, int(brow_begin), int(brow_end), C.byref(h)))
        return BmSpMatrix(h.value, parent=self)

    def free(self):
        if self.h:
            lib().bmsp_matrix_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# bmSparse_SpMV(A, v, u, batched)
def spmv(A, v, u=None, batched=False, stream=None):
    """v: DeviceArray (A's dtype); returns u as a DeviceArray (float32, float64 for F64)."""
    i = A.info()
    if u is None:
        u = DeviceArray(i["num_rows"], OUT_DTYPE[i["dtype"]])
    check(lib().bmsp_spmv(A.h, v.ptr, u.ptr, SPMV_BATCHED if batched else SPMV_DEFAULT, stream))
    return u


def spmv_launch_info(A, variant=0):
    """{"kernel", "compulsory_bytes", "format_bytes"} of the launch bmsp_spmv would make for (A, variant)."""
    name = C.create_string_buffer(128)
    cb, fb = C.c_int64(), C.c_int64()
    check(lib().bmsp_spmv_launch_info(A.h, int(variant), name, 128, C.byref(cb), C.byref(fb)))
    return {"kernel": name.value.decode(), "compulsory_bytes": cb.value, "format_bytes": fb.value}


def spmm(A, X, k, Y=None, ldx=None, ldy=None, stream=None):
    """X: DeviceArray holding num_cols x k row-major (A's dtype); returns Y (num_rows x k, float32 / float64 for F64)."""
    i = A.info()
    ldx = k if ldx is None else ldx
    ldy = k if ldy is None else ldy
    if Y is None:
        Y = DeviceArray(i["num_rows"] * ldy, OUT_DTYPE[i["dtype"]])
    check(lib().bmsp_spmm(A.h, X.ptr, int(ldx), Y.ptr, int(ldy), int(k), stream))
    return Y


# bmSparse_mult(A, B, C, mode, VERBOSE, tc_version)
def spgemm(A, B, mode=SORT_AUTO, tc_version=5, verbose=False, stream=None):
    h = C.c_void_p()
    st = SpgemmStats()
    check(lib().bmsp_spgemm(A.h, B.h, C.byref(h), int(mode), int(tc_version), int(bool(verbose)), stream, C.byref(st)))
    return BmSpMatrix(h.value), st.as_dict()


def spgemm_symbolic(A, B, mode=SORT_AUTO, tc_version=5, stream=None):
    """C's structure only (values allocated, zero): bmsp_spgemm_symbolic"""
    h = C.c_void_p()
    st = SpgemmStats()
    check(lib().bmsp_spgemm_symbolic(A.h, B.h, C.byref(h), int(mode), int(tc_version), stream, C.byref(st)))
    return BmSpMatrix(h.value), st.as_dict()


def spgemm_numeric(A, B, Cm, tc_version=5, stream=None):
    """the values of A x B into a C that already holds the product's structure: bmsp_spgemm_numeric"""
    st = SpgemmStats()
    check(lib().bmsp_spgemm_numeric(A.h, B.h, Cm.h, int(tc_version), stream, C.byref(st)))
    return st.as_dict()


# bb_segsort(keys, vals, n, segs, length)
def segsort(keys, vals, segs, stream=None):
    """keys: DeviceArray uint64; vals: DeviceArray of 4/8/16-byte items or None; segs: DeviceArray int32."""
    vb = 0 if vals is None else vals.dtype.itemsize
    check(lib().bmsp_segsort_u64(keys.ptr, None if vals is None else vals.ptr, vb, keys.n, segs.ptr, segs.n, stream))


def partition_rows(A, B, parts):
    bounds = np.zeros(parts + 1, dtype=np.int64)
    check(lib().bmsp_partition_rows(A.h, B.h, int(parts), bounds.ctypes.data))
    return bounds


def concat_panels(num_rows, num_cols, panels, dtype=F32):
    """panels: list of (keys, bmps, offsets, values) DeviceArrays (offsets with block_num+1 entries)."""
    P = len(panels)
    bn = np.array([p[0].n for p in panels], dtype=np.int64)
    nz = np.array([p[3].n for p in panels], dtype=np.int64)
    arr = lambda j: (C.c_void_p * P)(*[p[j].ptr for p in panels])
    k, b, o, v = arr(0), arr(1), arr(2), arr(3)
    h = C.c_void_p()
    check(lib().bmsp_matrix_concat_panels(int(num_rows), int(num_cols), P, bn.ctypes.data, nz.ctypes.data, k, b, o, v, dtype,
                                          C.byref(h)))
    return BmSpMatrix(h.value)


class Comm:
    """one RCCL communicator behind the C ABI (bmsp_comm_t): rank 0 makes the id, every rank calls Comm(id, world, rank)."""

    def __init__(self, id_bytes, world, rank):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        check(lib().bmsp_comm_init(buf, int(world), int(rank), C.byref(h)))
        self.h, self.world, self.rank = h.value, int(world), int(rank)

    @classmethod
    def loopback(cls, world):
        """`world` panels computed one after another on the current device, panels moved by device copies (no RCCL)."""
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib().bmsp_comm_init_loopback(int(world), C.byref(h)))
        self.h, self.world, self.rank = h.value, int(world), 0
        return self

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().bmsp_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def from_torch(dist, torch):
        """rendezvous over an existing torch.distributed group (the id travels as 128 bytes in a broadcast)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return Comm(box[0], world, rank)

    def free(self):
        if self.h:
            lib().bmsp_comm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def spgemm_sharded(comm, A, B, mode=SORT_AUTO, tc_version=5, verbose=False, stream=None, gather=True, rounds=0):
    """gather=False: owner keeps (no exchange, the rank's panel of C); rounds: panels per rank whose exchange overlaps the next product (0: library's choice)"""
    h = C.c_void_p()
    st, sh = SpgemmStats(), ShardStats()
    check(lib().bmsp_spgemm_sharded_ex(comm.h, A.h, B.h, C.byref(h), int(mode), int(tc_version), int(bool(verbose)), stream, C.byref(st), C.byref(sh),
                                       int(bool(gather)), int(rounds)))
    return BmSpMatrix(h.value), st.as_dict(), sh.as_dict()


def spmv_sharded(comm, A, v, u=None, variant=0, stream=None):
    i = A.info()
    if u is None:
        u = DeviceArray(i["num_rows"], OUT_DTYPE[i["dtype"]])
    sh = ShardStats()
    check(lib().bmsp_spmv_sharded(comm.h, A.h, v.ptr, u.ptr, int(variant), stream, C.byref(sh)))
    return u, sh.as_dict()


def shard_layout(block_nums, nnzs):
    """(block_start, value_start), parts+1 entries each: where every panel of a sharded product lands (host arithmetic of comm.hip)."""
    bn = np.ascontiguousarray(block_nums, dtype=np.int64)
    nz = np.ascontiguousarray(nnzs, dtype=np.int64)
    b0, z0 = np.zeros(bn.size + 1, np.int64), np.zeros(bn.size + 1, np.int64)
    check(lib().bmsp_shard_layout(int(bn.size), bn.ctypes.data, nz.ctypes.data, b0.ctypes.data, z0.ctypes.data))
    return b0, z0


def shard_row_slices(num_rows, bounds):
    """(row_start, row_count) per panel of a sharded SpMV with block-row bounds `bounds` (parts+1 entries)."""
    b = np.ascontiguousarray(bounds, dtype=np.int64)
    rs, rc = np.zeros(b.size - 1, np.int64), np.zeros(b.size - 1, np.int64)
    check(lib().bmsp_shard_row_slices(int(num_rows), int(b.size - 1), b.ctypes.data, rs.ctypes.data, rc.ctypes.data))
    return rs, rc


class CSRMatrix:
    """mirror of the reference's CSRMatrix (include/CSRMatrix.h:13-21)."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def from_mtx(path):
        h = C.c_void_p()
        check(lib().bmsp_csr_from_mtx(os.fsencode(path), C.byref(h)))
        return CSRMatrix(h.value)

    @staticmethod
    def from_arrays(num_rows, num_cols, row_offsets, cols, vals):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        c = np.ascontiguousarray(cols, dtype=np.int32)
        v = np.ascontiguousarray(vals, dtype=np.float32)
        h = C.c_void_p()
        check(lib().bmsp_csr_from_arrays(int(num_rows), int(num_cols), c.size, ro.ctypes.data, c.ctypes.data, v.ctypes.data, C.byref(h)))
        return CSRMatrix(h.value)

    def arrays(self):
        nr, nc, nnz = C.c_int(), C.c_int(), C.c_int64()
        check(lib().bmsp_csr_info(self.h, C.byref(nr), C.byref(nc), C.byref(nnz)))
        ro, c, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().bmsp_csr_arrays(self.h, C.byref(ro), C.byref(c), C.byref(v)))
        n = nnz.value
        f = lambda p, cnt, ct, dt: (np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (cnt,)).copy() if cnt else np.zeros(0, dt))
        return nr.value, nc.value, f(ro, nr.value + 1, C.c_int, np.int32), f(c, n, C.c_int, np.int32), f(v, n, C.c_float, np.float32)

    def multiply(self, other):
        h = C.c_void_p()
        check(lib().bmsp_csr_multiply(self.h, other.h, C.byref(h)))
        return CSRMatrix(h.value)

    def multiply_host(self, other, threads=0):
        """cusp::multiply's host path (no GPU call)."""
        h = C.c_void_p()
        check(lib().bmsp_csr_multiply_host(self.h, other.h, C.byref(h), int(threads)))
        return CSRMatrix(h.value)

    def shape(self):
        nr, nc, nnz = C.c_int(), C.c_int(), C.c_int64()
        check(lib().bmsp_csr_info(self.h, C.byref(nr), C.byref(nc), C.byref(nnz)))
        return nr.value, nc.value, nnz.value

    def spmv_host(self, x, threads=0):
        nr = self.shape()[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(nr, dtype=np.float32)
        check(lib().bmsp_csr_spmv_host(self.h, x.ctypes.data, y.ctypes.data, int(threads)))
        return y

    def spmv(self, x):
        nr = self.shape()[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(nr, dtype=np.float32)
        check(lib().bmsp_csr_spmv(self.h, x.ctypes.data, y.ctypes.data))
        return y

    def __del__(self):
        try:
            if self.h:
                lib().bmsp_csr_free(self.h)
                self.h = None
        except Exception:
            pass
