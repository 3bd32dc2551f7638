// capi.hip -- the extern "C" surface declared in include/bmsp.h.  Translates exceptions to status codes;
// no other logic lives here.
#include "matrix.h"
#include "prims.hip.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <omp.h>
#include <vector>

namespace bmsp {
const std::string &last_error();
void partition_rows(bmsp_matrix_s *A, bmsp_matrix_s *B, int parts, int64_t *bounds, hipStream_t st, uint64_t *total_out = nullptr);
void spgemm_paneled(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **C, int mode, int tc_version, int verbose, hipStream_t st,
                    bmsp_spgemm_stats *stats);
bmsp_matrix_s *row_panel(bmsp_matrix_s *m, int64_t rb, int64_t re, hipStream_t st);
bmsp_matrix_s *concat_panels(int num_rows, int num_cols, int parts, const int64_t *block_nums, const int64_t *nnzs,
                             uint64_t *const *d_keys, uint64_t *const *d_bmps, uint64_t *const *d_offsets, void *const *d_values,
                             bmsp_dtype dtype, hipStream_t st);
}  // namespace bmsp

using namespace bmsp;

// host CSR container (class CSRMatrix, include/CSRMatrix.h:13-21)
struct bmsp_csr_s {
    int num_rows = 0, num_cols = 0;
    std::vector<int> row_offsets, cols;
    std::vector<float> vals;
    // device forms, built on first use and cached
    bmsp_matrix_s *dev_normal = nullptr, *dev_transposed = nullptr;
    ~bmsp_csr_s()
    {
        free_matrix(dev_normal);
        free_matrix(dev_transposed);
    }
};

#define BMSP_API_BEGIN try {
#define BMSP_API_END                                         \
    return BMSP_OK;                                          \
    }                                                        \
    catch (const bmsp::Error &e)                             \
    {                                                        \
        bmsp::set_last_error(e.what());                      \
        return e.status;                                     \
    }                                                        \
    catch (const std::bad_alloc &)                           \
    {                                                        \
        bmsp::set_last_error("host allocation failed");      \
        return BMSP_ERR_NOMEM;                               \
    }                                                        \
    catch (const std::exception &e)                          \
    {                                                        \
        bmsp::set_last_error(e.what());                      \
        return BMSP_ERR_INVALID;                             \
    }

static void need(const void *p, const char *what)
{
    if (!p) fail(BMSP_ERR_INVALID, "%s is null", what);
}

static bmsp_matrix_s *build_from_host_coo(int num_rows, int num_cols, int64_t nnz, const int *rows, const int *cols, const double *vals,
                                          int transposed, bmsp_dtype dtype)
{
    if (dtype != BMSP_F32 && dtype != BMSP_F16 && dtype != BMSP_F64) fail(BMSP_ERR_INVALID, "unknown dtype %d", (int)dtype);
    for (int64_t i = 0; i < nnz; i++)
        if (rows[i] < 0 || rows[i] >= num_rows || cols[i] < 0 || cols[i] >= num_cols)
            fail(BMSP_ERR_INVALID, "entry %lld has index (%d,%d) outside %dx%d", (long long)i, rows[i], cols[i], num_rows, num_cols);
    size_t n = (size_t)nnz;
    DevBuf<int> dr(n), dc(n);
    DevBuf<double> dv(n);
    if (n) {
        copy_h2d_staged(dr.p, rows, sizeof(int) * n);
        copy_h2d_staged(dc.p, cols, sizeof(int) * n);
        copy_h2d_staged(dv.p, vals, sizeof(double) * n);
    }
    return build_from_device_coo(num_rows, num_cols, nnz, dr.p, dc.p, dv.p, transposed, dtype, nullptr);
}

extern "C" {

const char *bmsp_last_error(void) { return bmsp::last_error().c_str(); }
const char *bmsp_version(void) { return "bmsparse-mi355x 0.1 (gfx950)"; }

int bmsp_device_count(int *count)
{
    BMSP_API_BEGIN
    need(count, "count");
    BMSP_HIP(hipGetDeviceCount(count));
    BMSP_API_END
}
int bmsp_set_device(int device)
{
    BMSP_API_BEGIN
    BMSP_HIP(hipSetDevice(device));
    BMSP_API_END
}
int bmsp_malloc(void **dptr, size_t bytes)
{
    BMSP_API_BEGIN
    need(dptr, "dptr");
    *dptr = pool_alloc(bytes);
    BMSP_API_END
}
int bmsp_free(void *dptr)
{
    BMSP_API_BEGIN
    pool_free(dptr);
    BMSP_API_END
}
int bmsp_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    BMSP_API_BEGIN
    copy_h2d_staged(dst, src, bytes);
    BMSP_API_END
}
int bmsp_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    BMSP_API_BEGIN
    copy_d2h_staged(dst, src, bytes);
    BMSP_API_END
}
int bmsp_memcpy_d2d(void *dst, const void *src, size_t bytes)
{
    BMSP_API_BEGIN
    if (bytes) BMSP_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    BMSP_API_END
}
int bmsp_memset(void *dptr, int value, size_t bytes)
{
    BMSP_API_BEGIN
    if (bytes) BMSP_HIP(hipMemset(dptr, value, bytes));
    BMSP_API_END
}
int bmsp_synchronize(void)
{
    BMSP_API_BEGIN
    BMSP_HIP(hipDeviceSynchronize());
    BMSP_API_END
}
int bmsp_trim_pool(void)
{
    BMSP_API_BEGIN
    pool_trim();
    BMSP_API_END
}

int bmsp_event_create(void **event)
{
    BMSP_API_BEGIN
    need(event, "event");
    hipEvent_t e;
    BMSP_HIP(hipEventCreate(&e));
    *event = (void *)e;
    BMSP_API_END
}
int bmsp_event_record(void *event, void *stream)
{
    BMSP_API_BEGIN
    BMSP_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    BMSP_API_END
}
int bmsp_event_elapsed_ms(void *start, void *stop, float *ms)
{
    BMSP_API_BEGIN
    need(ms, "ms");
    BMSP_HIP(hipEventSynchronize((hipEvent_t)stop));
    BMSP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    BMSP_API_END
}
int bmsp_event_destroy(void *event)
{
    BMSP_API_BEGIN
    if (event) BMSP_HIP(hipEventDestroy((hipEvent_t)event));
    BMSP_API_END
}

int bmsp_matrix_from_mtx(const char *path, int transposed, bmsp_dtype dtype, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    load_kernels();
    need(path, "path");
    need(out, "out");
    HostCoo coo;
    read_matrix_market(path, coo);
    *out = build_from_host_coo(coo.num_rows, coo.num_cols, (int64_t)coo.rows.size(), coo.rows.data(), coo.cols.data(), coo.vals.data(),
                               transposed, dtype);
    BMSP_API_END
}

int bmsp_matrix_from_coo(int num_rows, int num_cols, int64_t nnz, const int *rows, const int *cols, const double *vals, int transposed,
                         bmsp_dtype dtype, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    load_kernels();
    need(out, "out");
    if (nnz > 0) { need(rows, "rows"); need(cols, "cols"); need(vals, "vals"); }
    *out = build_from_host_coo(num_rows, num_cols, nnz, rows, cols, vals, transposed, dtype);
    BMSP_API_END
}

int bmsp_matrix_from_coo_device(int num_rows, int num_cols, int64_t nnz, const int *d_rows, const int *d_cols, const double *d_vals,
                                int transposed, bmsp_dtype dtype, void *stream, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    load_kernels();
    need(out, "out");
    *out = build_from_device_coo(num_rows, num_cols, nnz, d_rows, d_cols, d_vals, transposed, dtype, as_stream(stream));
    BMSP_API_END
}

int bmsp_matrix_from_arrays(int num_rows, int num_cols, int64_t block_num, int64_t nnz, uint64_t *d_keys, uint64_t *d_bmps,
                            uint64_t *d_offsets, void *d_values, bmsp_dtype dtype, int transposed, int ownership, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    load_kernels();
    need(out, "out");
    if (block_num < 0 || nnz < 0 || num_rows < 0 || num_cols < 0) fail(BMSP_ERR_INVALID, "negative size");
    if (block_num > 0) { need(d_keys, "keys"); need(d_bmps, "bmps"); need(d_offsets, "offsets"); }
    if (nnz > 0) need(d_values, "values");
    if (ownership < 0 || ownership > 2) fail(BMSP_ERR_INVALID, "ownership must be 0, 1 or 2");
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> m(new bmsp_matrix_s(), free_matrix);
    m->num_rows = num_rows; m->num_cols = num_cols; m->block_num = block_num; m->nnz = nnz;
    m->dtype = dtype; m->transposed = transposed ? 1 : 0;
    const size_t nb = (size_t)block_num, es = dtype_size(dtype);
    // offsets always carry block_num+1 entries internally; the caller's array may hold only block_num
    // (the reference builder's, src/bmSpMatrix.cu:194), so the terminal entry is written here.
    uint64_t term = (uint64_t)nnz;
    if (ownership == 0 || ownership == 1) {
        m->ownership = 1;
        m->offsets = (uint64_t *)pool_alloc(8 * (nb + 1));
        if (nb) BMSP_HIP(hipMemcpy(m->offsets, d_offsets, 8 * nb, hipMemcpyDeviceToDevice));
        BMSP_HIP(hipMemcpy(m->offsets + nb, &term, 8, hipMemcpyHostToDevice));
        if (ownership == 1) {
            if ((nb && (!pool_owns(d_keys) || !pool_owns(d_bmps))) || (nnz && !pool_owns(d_values)))
                fail(BMSP_ERR_INVALID, "ownership=1 needs arrays allocated with bmsp_malloc");
            m->keys = d_keys; m->bmps = d_bmps; m->values = d_values;
            pool_free(d_offsets);  // adopted and replaced by the block_num+1 copy
        } else {
            m->keys = (uint64_t *)pool_alloc(8 * (nb ? nb : 1));
            m->bmps = (uint64_t *)pool_alloc(8 * (nb ? nb : 1));
            m->values = pool_alloc(es * (size_t)(nnz ? nnz : 1));
            if (nb) {
                BMSP_HIP(hipMemcpy(m->keys, d_keys, 8 * nb, hipMemcpyDeviceToDevice));
                BMSP_HIP(hipMemcpy(m->bmps, d_bmps, 8 * nb, hipMemcpyDeviceToDevice));
            }
            if (nnz) BMSP_HIP(hipMemcpy(m->values, d_values, es * (size_t)nnz, hipMemcpyDeviceToDevice));
        }
    } else {
        // borrowed: the caller's offsets must already hold block_num+1 entries
        m->ownership = 2;
        m->keys = d_keys; m->bmps = d_bmps; m->offsets = d_offsets; m->values = d_values;
    }
    *out = m.release();
    BMSP_API_END
}

namespace {
struct CacheHeader {
    char magic[8];
    int32_t num_rows, num_cols, dtype, transposed;
    int64_t block_num, nnz;
};
const char kCacheMagic[8] = {'B', 'M', 'S', 'P', 'v', '1', 0, 0};
}  // namespace

int bmsp_matrix_save(bmsp_matrix_t m, const char *path)
{
    BMSP_API_BEGIN
    need(m, "matrix"); need(path, "path");
    if (m->ownership == 2 && m->view_block_begin) fail(BMSP_ERR_UNSUPPORTED, "row-panel views cannot be saved; save the parent");
    FILE *f = fopen(path, "wb");
    if (!f) fail(BMSP_ERR_IO, "cannot create '%s'", path);
    std::unique_ptr<FILE, int (*)(FILE *)> guard(f, fclose);
    CacheHeader h;
    memcpy(h.magic, kCacheMagic, 8);
    h.num_rows = m->num_rows; h.num_cols = m->num_cols; h.dtype = (int32_t)m->dtype; h.transposed = m->transposed;
    h.block_num = m->block_num; h.nnz = m->nnz;
    const size_t nb = (size_t)m->block_num, es = dtype_size(m->dtype);
    std::vector<char> buf;
    auto dump = [&](const void *dptr, size_t bytes) {
        buf.resize(bytes);
        copy_d2h_staged(buf.data(), dptr, bytes);
        if (bytes && fwrite(buf.data(), 1, bytes, f) != bytes) fail(BMSP_ERR_IO, "short write to '%s'", path);
    };
    if (fwrite(&h, sizeof h, 1, f) != 1) fail(BMSP_ERR_IO, "short write to '%s'", path);
    dump(m->keys, 8 * nb); dump(m->bmps, 8 * nb); dump(m->offsets, 8 * (nb + 1)); dump(m->values, es * (size_t)m->nnz);
    BMSP_API_END
}

int bmsp_matrix_load(const char *path, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    load_kernels();
    need(path, "path"); need(out, "out");
    FILE *f = fopen(path, "rb");
    if (!f) fail(BMSP_ERR_IO, "cannot open '%s'", path);
    std::unique_ptr<FILE, int (*)(FILE *)> guard(f, fclose);
    CacheHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kCacheMagic, 8) != 0) fail(BMSP_ERR_IO, "'%s' is not a bmSparse cache file", path);
    if (h.block_num < 0 || h.nnz < 0 || h.num_rows < 0 || h.num_cols < 0 || h.dtype < 0 || h.dtype > 2) fail(BMSP_ERR_IO, "'%s': corrupt header", path);
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> m(new bmsp_matrix_s(), free_matrix);
    m->num_rows = h.num_rows; m->num_cols = h.num_cols; m->dtype = (bmsp_dtype)h.dtype; m->transposed = h.transposed ? 1 : 0;
    m->block_num = h.block_num; m->nnz = h.nnz;
    const size_t nb = (size_t)h.block_num, es = dtype_size(m->dtype);
    std::vector<char> buf;
    auto slurp = [&](void **dptr, size_t bytes) {
        *dptr = pool_alloc(bytes ? bytes : 8);
        buf.resize(bytes);
        if (bytes && fread(buf.data(), 1, bytes, f) != bytes) fail(BMSP_ERR_IO, "'%s' is truncated", path);
        copy_h2d_staged(*dptr, buf.data(), bytes);
    };
    slurp((void **)&m->keys, 8 * nb); slurp((void **)&m->bmps, 8 * nb); slurp((void **)&m->offsets, 8 * (nb + 1));
    slurp(&m->values, es * (size_t)h.nnz);
    ensure_rowptr(m.get(), nullptr);
    BMSP_HIP(hipStreamSynchronize(nullptr));
    *out = m.release();
    BMSP_API_END
}

int bmsp_matrix_free(bmsp_matrix_t m)
{
    BMSP_API_BEGIN
    free_matrix(m);
    BMSP_API_END
}

int bmsp_matrix_invalidate(bmsp_matrix_t m, int structure_changed)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    invalidate_matrix(m, structure_changed);
    BMSP_API_END
}

int bmsp_matrix_info(bmsp_matrix_t m, int *num_rows, int *num_cols, int64_t *nnz, int64_t *block_num, bmsp_dtype *dtype, int *transposed)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    if (num_rows) *num_rows = m->num_rows;
    if (num_cols) *num_cols = m->num_cols;
    if (nnz) *nnz = m->nnz;
    if (block_num) *block_num = m->block_num;
    if (dtype) *dtype = m->dtype;
    if (transposed) *transposed = m->transposed;
    BMSP_API_END
}

int bmsp_matrix_arrays(bmsp_matrix_t m, uint64_t **d_keys, uint64_t **d_bmps, uint64_t **d_offsets, void **d_values)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    if (d_keys) *d_keys = m->keys;
    if (d_bmps) *d_bmps = m->bmps;
    if (d_offsets) *d_offsets = m->offsets;
    if (d_values) *d_values = m->values;
    BMSP_API_END
}

int bmsp_matrix_block_row_ptr(bmsp_matrix_t m, const uint32_t **d_rowptr, int64_t *num_block_rows)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    ensure_rowptr(m, nullptr);
    BMSP_HIP(hipStreamSynchronize(nullptr));
    if (d_rowptr) *d_rowptr = m->rowptr;
    if (num_block_rows) *num_block_rows = m->num_block_rows();
    BMSP_API_END
}

int bmsp_matrix_prepare(bmsp_matrix_t m, int what, void *stream)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    if (what & ~(BMSP_PREPARE_SPMV | BMSP_PREPARE_SPGEMM)) fail(BMSP_ERR_INVALID, "unknown prepare flags %d", what);
    ensure_rowptr(m, as_stream(stream));
    if (what & BMSP_PREPARE_SPMV) prepare_spmv(m, as_stream(stream));
    if (what & BMSP_PREPARE_SPGEMM) prepare_spgemm_operand(m, as_stream(stream));
    BMSP_API_END
}

int bmsp_matrix_to_coo_host(bmsp_matrix_t m, int *rows, int *cols, double *vals)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    if (m->nnz) { need(rows, "rows"); need(cols, "cols"); need(vals, "vals"); }
    matrix_to_coo_host(m, rows, cols, vals, nullptr);
    BMSP_API_END
}

int bmsp_matrix_to_coo_device(bmsp_matrix_t m, int *d_rows, int *d_cols, double *d_vals, void *stream)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    if (m->nnz) { need(d_rows, "rows"); need(d_cols, "cols"); need(d_vals, "vals"); }
    matrix_to_coo_device_split(m, d_rows, d_cols, d_vals, as_stream(stream));
    BMSP_API_END
}

int bmsp_matrix_to_csr_device(bmsp_matrix_t m, int *d_row_offsets, int *d_cols, double *d_vals, void *stream)
{
    BMSP_API_BEGIN
    need(m, "matrix"); need(d_row_offsets, "row_offsets");
    if (m->nnz) { need(d_cols, "cols"); need(d_vals, "vals"); }
    matrix_to_csr_device(m, d_row_offsets, d_cols, d_vals, as_stream(stream));
    BMSP_API_END
}

int bmsp_matrix_from_csr_device(int num_rows, int num_cols, int64_t nnz, const int *d_row_offsets, const int *d_cols, const double *d_vals,
                                int transposed, bmsp_dtype dtype, void *stream, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    load_kernels();
    need(out, "out"); need(d_row_offsets, "row_offsets");
    if (num_rows < 0 || num_cols < 0 || nnz < 0) fail(BMSP_ERR_INVALID, "negative size");
    if (nnz) { need(d_cols, "cols"); need(d_vals, "vals"); }
    *out = build_from_device_csr(num_rows, num_cols, nnz, d_row_offsets, d_cols, d_vals, transposed, dtype, as_stream(stream));
    BMSP_API_END
}

int bmsp_matrix_compare_device(bmsp_matrix_t m, int64_t nnz, const int *d_rows, const int *d_cols, const double *d_vals, double *mean_rel_err,
                               int64_t *missing, void *stream)
{
    BMSP_API_BEGIN
    need(m, "matrix"); need(mean_rel_err, "mean_rel_err");
    if (nnz < 0) fail(BMSP_ERR_INVALID, "negative size");
    if (nnz) { need(d_rows, "rows"); need(d_cols, "cols"); need(d_vals, "vals"); }
    matrix_compare_device(m, nnz, d_rows, d_cols, d_vals, mean_rel_err, missing, as_stream(stream));
    BMSP_API_END
}

int bmsp_matrix_compare(bmsp_matrix_t m, int64_t nnz, const int *rows, const int *cols, const double *vals, double *mean_rel_err,
                        int64_t *missing)
{
    BMSP_API_BEGIN
    need(m, "matrix");
    need(mean_rel_err, "mean_rel_err");
    size_t n = (size_t)m->nnz;
    std::vector<int> mr(n), mc(n);
    std::vector<double> mv(n);
    matrix_to_coo_host(m, mr.data(), mc.data(), mv.data(), nullptr);
    // comparand sorted by (row, col) (src/bmSpMatrix.cu:394)
    std::vector<int64_t> perm((size_t)nnz);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
        return rows[a] != rows[b] ? rows[a] < rows[b] : cols[a] < cols[b];
    });
    const double eps = 1e-8;  // :403
    double count = 0;
    int64_t miss = 0;
    size_t j = 0;
    for (size_t i = 0; i < n; i++) {
        // skip comparand entries that m does not hold (:405-408)
        while (j < (size_t)nnz && (rows[perm[j]] < mr[i] || (rows[perm[j]] == mr[i] && cols[perm[j]] < mc[i]))) j++;
        if (j >= (size_t)nnz || rows[perm[j]] != mr[i] || cols[perm[j]] != mc[i]) { miss++; continue; }
        double e = std::fabs(vals[perm[j]]) < eps ? 0.0 : vals[perm[j]];
        double r = std::fabs(mv[i]) < eps ? 0.0 : mv[i];
        count += std::fabs(e - r) / std::max(std::fabs(e), eps);  // :418
        j++;
    }
    *mean_rel_err = n ? count / (double)n : 0.0;  // "Final:" (:429)
    if (missing) *missing = miss;
    BMSP_API_END
}

int bmsp_spmv(bmsp_matrix_t A, const void *d_v, void *d_u, int variant, void *stream)
{
    BMSP_API_BEGIN
    need(A, "A");
    spmv(A, d_v, d_u, variant, as_stream(stream));
    BMSP_API_END
}

int bmsp_spmv_launch_info(bmsp_matrix_t A, int variant, char *kernel_name, size_t kernel_name_cap, int64_t *compulsory_bytes, int64_t *format_bytes)
{
    BMSP_API_BEGIN
    need(A, "A");
    spmv_launch_info(A, variant, nullptr, kernel_name, kernel_name_cap, compulsory_bytes, format_bytes);
    BMSP_API_END
}

int bmsp_comm_unique_id(void *id_bytes)
{
    BMSP_API_BEGIN
    need(id_bytes, "id_bytes");
    comm_unique_id(id_bytes);
    BMSP_API_END
}

int bmsp_comm_init(const void *id_bytes, int world, int rank, bmsp_comm_t *out)
{
    BMSP_API_BEGIN
    need(id_bytes, "id_bytes"); need(out, "out");
    *out = comm_init(id_bytes, world, rank);
    BMSP_API_END
}

int bmsp_comm_init_loopback(int world, bmsp_comm_t *out)
{
    BMSP_API_BEGIN
    need(out, "out");
    *out = comm_init_loopback(world);
    BMSP_API_END
}

int bmsp_shard_layout(int parts, const int64_t *block_nums, const int64_t *nnzs, int64_t *block_start, int64_t *value_start)
{
    BMSP_API_BEGIN
    need(block_nums, "block_nums"); need(nnzs, "nnzs"); need(block_start, "block_start"); need(value_start, "value_start");
    shard_layout(parts, block_nums, nnzs, block_start, value_start);
    BMSP_API_END
}

int bmsp_shard_row_slices(int num_rows, int parts, const int64_t *bounds, int64_t *row_start, int64_t *row_count)
{
    BMSP_API_BEGIN
    need(bounds, "bounds"); need(row_start, "row_start"); need(row_count, "row_count");
    shard_row_slices(num_rows, parts, bounds, row_start, row_count);
    BMSP_API_END
}

int bmsp_comm_init_from_env(bmsp_comm_t *out)
{
    BMSP_API_BEGIN
    need(out, "out");
    *out = comm_init_from_env();
    BMSP_API_END
}

int bmsp_comm_info(bmsp_comm_t c, int *rank, int *world)
{
    BMSP_API_BEGIN
    need(c, "comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    BMSP_API_END
}

int bmsp_comm_free(bmsp_comm_t c)
{
    BMSP_API_BEGIN
    comm_free(c);
    BMSP_API_END
}

int bmsp_spgemm_sharded(bmsp_comm_t c, bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, int verbose, void *stream,
                        bmsp_spgemm_stats *stats, bmsp_shard_stats *shard)
{
    BMSP_API_BEGIN
    need(c, "comm"); need(A, "A"); need(B, "B"); need(C, "C");
    spgemm_sharded(c, A, B, C, mode, tc_version, verbose, as_stream(stream), stats, shard);
    BMSP_API_END
}

int bmsp_spgemm_sharded_ex(bmsp_comm_t c, bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, int verbose, void *stream,
                           bmsp_spgemm_stats *stats, bmsp_shard_stats *shard, int gather, int rounds)
{
    BMSP_API_BEGIN
    need(c, "comm"); need(A, "A"); need(B, "B"); need(C, "C");
    spgemm_sharded(c, A, B, C, mode, tc_version, verbose, as_stream(stream), stats, shard, gather ? 1 : 0, rounds);
    BMSP_API_END
}

int bmsp_spmv_sharded(bmsp_comm_t c, bmsp_matrix_t A, const void *d_v, void *d_u, int variant, void *stream, bmsp_shard_stats *shard)
{
    BMSP_API_BEGIN
    need(c, "comm"); need(A, "A");
    spmv_sharded(c, A, d_v, d_u, variant, as_stream(stream), shard);
    BMSP_API_END
}

int bmsp_selftest_mfma_layout(int *mismatches)
{
    BMSP_API_BEGIN
    need(mismatches, "mismatches");
    *mismatches = mfma32_selftest(nullptr);
    BMSP_API_END
}

int bmsp_selftest_mfma_f32_cancel(int *mismatches, int *exp_floor)
{
    BMSP_API_BEGIN
    need(mismatches, "mismatches");
    *mismatches = mfma_f32_selftest(nullptr, true);
    if (exp_floor) *exp_floor = mac_f32_exp_floor(nullptr);
    BMSP_API_END
}

int bmsp_selftest_tile_product(int *mismatches)
{
    BMSP_API_BEGIN
    need(mismatches, "mismatches");
    *mismatches = tile_product_selftest(nullptr);
    BMSP_API_END
}

int bmsp_selftest_mfma_f32_chain(int *mismatches)
{
    BMSP_API_BEGIN
    need(mismatches, "mismatches");
    *mismatches = mfma_f32_selftest(nullptr);
    BMSP_API_END
}

int bmsp_spmm(bmsp_matrix_t A, const void *d_X, int64_t ldx, void *d_Y, int64_t ldy, int k, void *stream)
{
    BMSP_API_BEGIN
    need(A, "A");
    if (A->num_cols) need(d_X, "X");
    if (A->num_rows) need(d_Y, "Y");
    spmm(A, d_X, ldx, d_Y, ldy, k, as_stream(stream));
    BMSP_API_END
}

int bmsp_spgemm(bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, int verbose, void *stream,
                bmsp_spgemm_stats *stats)
{
    BMSP_API_BEGIN
    try {
        if (getenv("BMSP_SPGEMM_FORCE_PANELS")) throw TaskRangeExceeded(BMSP_ERR_LIMIT, "forced (test hook for the paneled path)");
        spgemm(A, B, C, mode, tc_version, verbose, as_stream(stream), stats);
    } catch (const TaskRangeExceeded &) {
        // more candidate block pairs than one task list can index: the same product, block-row panel after panel
        spgemm_paneled(A, B, C, mode, tc_version, verbose, as_stream(stream), stats);
    }
    BMSP_API_END
}

int bmsp_spgemm_symbolic(bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, void *stream, bmsp_spgemm_stats *stats)
{
    BMSP_API_BEGIN
    spgemm(A, B, C, mode, tc_version, 0, as_stream(stream), stats, true);
    BMSP_API_END
}

int bmsp_spgemm_numeric(bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t C, int tc_version, void *stream, bmsp_spgemm_stats *stats)
{
    BMSP_API_BEGIN
    spgemm_numeric(A, B, C, tc_version, as_stream(stream), stats);
    BMSP_API_END
}

int bmsp_segsort_u64(uint64_t *d_keys, void *d_vals, int val_bytes, int64_t n, const int *d_segs, int64_t num_segs, void *stream)
{
    BMSP_API_BEGIN
    segsort_u64(d_keys, d_vals, val_bytes, n, d_segs, num_segs, as_stream(stream));
    BMSP_API_END
}

int bmsp_partition_rows(bmsp_matrix_t A, bmsp_matrix_t B, int parts, int64_t *bounds)
{
    BMSP_API_BEGIN
    need(A, "A"); need(B, "B"); need(bounds, "bounds");
    partition_rows(A, B, parts, bounds, nullptr);
    BMSP_API_END
}

int bmsp_matrix_row_panel(bmsp_matrix_t m, int64_t brow_begin, int64_t brow_end, bmsp_matrix_t *view)
{
    BMSP_API_BEGIN
    need(m, "matrix"); need(view, "view");
    *view = row_panel(m, brow_begin, brow_end, nullptr);
    BMSP_API_END
}

int bmsp_matrix_concat_panels(int num_rows, int num_cols, int parts, const int64_t *block_nums, const int64_t *nnzs,
                              uint64_t *const *d_keys, uint64_t *const *d_bmps, uint64_t *const *d_offsets, void *const *d_values,
                              bmsp_dtype dtype, bmsp_matrix_t *out)
{
    BMSP_API_BEGIN
    need(out, "out");
    if (parts < 1) fail(BMSP_ERR_INVALID, "parts must be >= 1");
    *out = concat_panels(num_rows, num_cols, parts, block_nums, nnzs, d_keys, d_bmps, d_offsets, d_values, dtype, nullptr);
    BMSP_API_END
}

// ---- host CSR --------------------------------------------------------------------------------------------
// CSRMatrix is a host container in the reference (cusp::csr_matrix<int,float,host_memory>) whose multiply is
// cusp::multiply.  Here the container stays on the host and multiply / spmv run on the GPU through the bmSparse
// operators (CSR -> bmSparse build, product, expansion back to CSR with columns ascending and numeric zeros
// dropped as cusp's host SpGEMM does, csr_spgemm.h:135).

static bmsp_matrix_s *csr_device_form(bmsp_csr_s *m, int transposed)
{
    bmsp_matrix_s *&slot = transposed ? m->dev_transposed : m->dev_normal;
    if (slot) return slot;
    size_t n = m->cols.size();
    std::vector<int> rows(n);
    std::vector<double> vals(n);
    for (int r = 0; r < m->num_rows; r++)
        for (int k = m->row_offsets[r]; k < m->row_offsets[r + 1]; k++) rows[(size_t)k] = r;
    for (size_t i = 0; i < n; i++) vals[i] = (double)m->vals[i];
    slot = build_from_host_coo(m->num_rows, m->num_cols, (int64_t)n, rows.data(), m->cols.data(), vals.data(), transposed, BMSP_F32);
    return slot;
}

int bmsp_csr_from_mtx(const char *path, bmsp_csr_t *out)
{
    BMSP_API_BEGIN
    need(path, "path"); need(out, "out");
    HostCoo coo;
    read_matrix_market(path, coo);
    size_t n = coo.rows.size();
    std::unique_ptr<bmsp_csr_s> m(new bmsp_csr_s());
    m->num_rows = coo.num_rows; m->num_cols = coo.num_cols;
    m->row_offsets.assign((size_t)coo.num_rows + 1, 0);
    m->cols.resize(n); m->vals.resize(n);
    // CUSP's reader sorts by (row, column) (cusp/io/detail/matrix_market.inl:295): stable counting sort by row, then a
    // stable sort by column inside each row
    for (size_t i = 0; i < n; i++) m->row_offsets[(size_t)coo.rows[i] + 1]++;
    for (int r = 0; r < coo.num_rows; r++) m->row_offsets[(size_t)r + 1] += m->row_offsets[(size_t)r];
    {
        std::vector<int> fill(m->row_offsets.begin(), m->row_offsets.end() - 1);
        std::vector<std::pair<int, float>> tmp(n);
        for (size_t i = 0; i < n; i++) tmp[(size_t)fill[(size_t)coo.rows[i]]++] = std::make_pair(coo.cols[i], (float)coo.vals[i]);
        for (int r = 0; r < coo.num_rows; r++)
            std::stable_sort(tmp.begin() + m->row_offsets[(size_t)r], tmp.begin() + m->row_offsets[(size_t)r + 1],
                             [](const std::pair<int, float> &a, const std::pair<int, float> &b) { return a.first < b.first; });
        for (size_t i = 0; i < n; i++) { m->cols[i] = tmp[i].first; m->vals[i] = tmp[i].second; }
    }
    *out = m.release();
    BMSP_API_END
}

int bmsp_csr_from_arrays(int num_rows, int num_cols, int64_t nnz, const int *row_offsets, const int *cols, const float *vals,
                         bmsp_csr_t *out)
{
    BMSP_API_BEGIN
    need(out, "out"); need(row_offsets, "row_offsets");
    if (num_rows < 0 || num_cols < 0 || nnz < 0) fail(BMSP_ERR_INVALID, "negative size");
    if (row_offsets[0] != 0 || row_offsets[num_rows] != nnz) fail(BMSP_ERR_INVALID, "row_offsets do not span [0,nnz]");
    std::unique_ptr<bmsp_csr_s> m(new bmsp_csr_s());
    m->num_rows = num_rows; m->num_cols = num_cols;
    m->row_offsets.assign(row_offsets, row_offsets + num_rows + 1);
    m->cols.assign(cols, cols + nnz);
    m->vals.assign(vals, vals + nnz);
    for (int64_t i = 0; i < nnz; i++)
        if (cols[i] < 0 || cols[i] >= num_cols) fail(BMSP_ERR_INVALID, "column index %d outside [0,%d)", cols[i], num_cols);
    *out = m.release();
    BMSP_API_END
}

int bmsp_csr_info(bmsp_csr_t m, int *num_rows, int *num_cols, int64_t *nnz)
{
    BMSP_API_BEGIN
    need(m, "csr");
    if (num_rows) *num_rows = m->num_rows;
    if (num_cols) *num_cols = m->num_cols;
    if (nnz) *nnz = (int64_t)m->cols.size();
    BMSP_API_END
}

int bmsp_csr_arrays(bmsp_csr_t m, const int **row_offsets, const int **cols, const float **vals)
{
    BMSP_API_BEGIN
    need(m, "csr");
    if (row_offsets) *row_offsets = m->row_offsets.data();
    if (cols) *cols = m->cols.data();
    if (vals) *vals = m->vals.data();
    BMSP_API_END
}

int bmsp_csr_multiply(bmsp_csr_t A, bmsp_csr_t B, bmsp_csr_t *C)
{
    BMSP_API_BEGIN
    need(A, "A"); need(B, "B"); need(C, "C");
    if (A->num_cols != B->num_rows) fail(BMSP_ERR_INVALID, "shape mismatch");
    bmsp_matrix_s *dA = csr_device_form(A, 0), *dB = csr_device_form(B, 1), *dC = nullptr;
    spgemm(dA, dB, &dC, BMSP_SORT_AUTO, 5, 0, nullptr, nullptr);
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> guard(dC, free_matrix);
    size_t n = (size_t)dC->nnz;
    std::vector<int> r(n), c(n);
    std::vector<double> v(n);
    matrix_to_coo_host(dC, r.data(), c.data(), v.data(), nullptr);
    std::unique_ptr<bmsp_csr_s> m(new bmsp_csr_s());
    m->num_rows = A->num_rows; m->num_cols = B->num_cols;
    m->row_offsets.assign((size_t)A->num_rows + 1, 0);
    for (size_t i = 0; i < n; i++) {
        if ((float)v[i] == 0.0f) continue;  // cusp drops numeric zeros (csr_spgemm.h:135)
        m->row_offsets[(size_t)r[i] + 1]++;
        m->cols.push_back(c[i]);
        m->vals.push_back((float)v[i]);
    }
    for (int i = 0; i < A->num_rows; i++) m->row_offsets[(size_t)i + 1] += m->row_offsets[(size_t)i];
    *C = m.release();
    BMSP_API_END
}

int bmsp_csr_spmv(bmsp_csr_t A, const float *x, float *y)
{
    BMSP_API_BEGIN
    need(A, "A"); need(x, "x"); need(y, "y");
    bmsp_matrix_s *dA = csr_device_form(A, 0);
    DevBuf<float> dx((size_t)A->num_cols), dy((size_t)A->num_rows);
    copy_h2d_staged(dx.p, x, 4 * (size_t)A->num_cols);
    spmv(dA, dx.p, dy.p, BMSP_SPMV_DEFAULT, nullptr);
    copy_d2h_staged(y, dy.p, 4 * (size_t)A->num_rows);
    BMSP_API_END
}

// ---- the host path of class CSRMatrix (configs[0]: "cusp::multiply ..., host CPU path"): the reference's CSRMatrix wraps a
// cusp::csr_matrix<int,float,host_memory> (include/CSRMatrix.h:20), i.e. cusp::multiply runs on the host.  These two follow the
// host algorithms' contracts -- y[i] = sum over the row in column order, accumulator initialised to 0
// (cusp/system/detail/sequential/multiply/csr_spmv.h:56-73; rows in parallel: omp/detail/multiply/csr_spmv.h:67-85); row-wise
// Gustavson product in two passes, numeric zeros dropped by the sequential form (sequential/multiply/csr_spgemm.h:39-157) -- with
// `threads` host threads (0 = all).  No GPU call is made.
int bmsp_csr_spmv_host(bmsp_csr_t A, const float *x, float *y, int threads)
{
    BMSP_API_BEGIN
    need(A, "A"); need(x, "x"); need(y, "y");
    const int nr = A->num_rows;
    const int *ro = A->row_offsets.data(), *ci = A->cols.data();
    const float *va = A->vals.data();
    const int nt = threads > 0 ? threads : omp_get_max_threads();
    (void)nt;
    // rows in parallel, each row summed in column order (omp/detail/multiply/csr_spmv.h:67-85)
#pragma omp parallel for num_threads(nt) schedule(static, 512) if (nt > 1 && nr >= 4096)
    for (int i = 0; i < nr; i++) {
        float sum = 0.0f;
        for (int jj = ro[i]; jj < ro[i + 1]; jj++) sum += va[jj] * x[ci[jj]];
        y[i] = sum;
    }
    BMSP_API_END
}

int bmsp_csr_multiply_host(bmsp_csr_t A, bmsp_csr_t B, bmsp_csr_t *C, int threads)
{
    BMSP_API_BEGIN
    need(A, "A"); need(B, "B"); need(C, "C");
    if (A->num_cols != B->num_rows) fail(BMSP_ERR_INVALID, "shape mismatch");
    const int nr = A->num_rows, nc = B->num_cols;
    const int nt = std::max(1, std::min(threads > 0 ? threads : omp_get_max_threads(), std::max(1, nr / 256)));
    std::unique_ptr<bmsp_csr_s> m(new bmsp_csr_s());
    m->num_rows = nr; m->num_cols = nc;
    m->row_offsets.assign((size_t)nr + 1, 0);
    std::vector<std::vector<int>> t_cols((size_t)nt);
    std::vector<std::vector<float>> t_vals((size_t)nt);
    std::vector<int> bounds((size_t)nt + 1, nr);
    bounds[0] = 0;
    {
        const int64_t nnz = A->row_offsets[(size_t)nr];
        for (int t = 1; t < nt; t++)
            bounds[(size_t)t] = std::max(bounds[(size_t)t - 1], (int)(std::upper_bound(A->row_offsets.begin(), A->row_offsets.end(), (int)(nnz * t / nt)) - A->row_offsets.begin()));
        for (int t = 1; t < nt; t++) bounds[(size_t)t] = std::min(bounds[(size_t)t], nr);
    }
    auto work = [&](int t) {
        // Gustavson with a linked list of touched columns (spmm_csr_pass2, csr_spgemm.h:79-157); numeric zeros are not stored (:135)
        std::vector<int> next((size_t)nc, -1);
        std::vector<float> sums((size_t)nc, 0.0f);
        std::vector<int> &oc = t_cols[(size_t)t];
        std::vector<float> &ov = t_vals[(size_t)t];
        for (int i = bounds[(size_t)t]; i < bounds[(size_t)t + 1]; i++) {
            int head = -2, length = 0;
            for (int jj = A->row_offsets[(size_t)i]; jj < A->row_offsets[(size_t)i + 1]; jj++) {
                const int j = A->cols[(size_t)jj];
                const float v = A->vals[(size_t)jj];
                for (int kk = B->row_offsets[(size_t)j]; kk < B->row_offsets[(size_t)j + 1]; kk++) {
                    const int k = B->cols[(size_t)kk];
                    sums[(size_t)k] += v * B->vals[(size_t)kk];
                    if (next[(size_t)k] == -1) { next[(size_t)k] = head; head = k; length++; }
                }
            }
            int kept = 0;
            for (int q = 0; q < length; q++) {
                if (sums[(size_t)head] != 0.0f) { oc.push_back(head); ov.push_back(sums[(size_t)head]); kept++; }
                const int tmp = head;
                head = next[(size_t)head];
                next[(size_t)tmp] = -1;
                sums[(size_t)tmp] = 0.0f;
            }
            m->row_offsets[(size_t)i + 1] = kept;
        }
    };
#pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (int t = 0; t < nt; t++) work(t);
    for (int i = 0; i < nr; i++) m->row_offsets[(size_t)i + 1] += m->row_offsets[(size_t)i];
    m->cols.reserve((size_t)m->row_offsets[(size_t)nr]);
    m->vals.reserve((size_t)m->row_offsets[(size_t)nr]);
    for (int t = 0; t < nt; t++) {
        m->cols.insert(m->cols.end(), t_cols[(size_t)t].begin(), t_cols[(size_t)t].end());
        m->vals.insert(m->vals.end(), t_vals[(size_t)t].begin(), t_vals[(size_t)t].end());
    }
    *C = m.release();
    BMSP_API_END
}

int bmsp_csr_free(bmsp_csr_t m)
{
    BMSP_API_BEGIN
    delete m;
    BMSP_API_END
}

}  // extern "C"
