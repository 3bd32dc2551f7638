// shard.hip -- row-panel sharding of the SpGEMM across GPUs (one process per GPU).
//
// New relative to the reference (single GPU).  C's block-row i depends only on A's block-row i and on all of B
// (tasks are generated per A block and grouped by block-row, src/bmSparse_SPGEMM.cu:883-932,973-1004), so A is
// split into contiguous block-row panels balanced by candidate-task count, B is replicated, every rank runs the
// whole pipeline on its panel, and the C panels are concatenated after an allgatherv of the four arrays.
#include "matrix.h"
#include "prims.hip.h"
#include <memory>
#include <vector>
#include <vector>

namespace bmsp {
namespace {
struct PanelFanOut {
    const uint64_t *a_keys;
    const uint32_t *b_rowptr;
    uint64_t n_a;
    uint32_t b_block_rows;
    __device__ uint64_t operator()(uint64_t a) const
    {
        if (a >= n_a) return 0;
        uint32_t col = key_col(a_keys[a]);
        return col < b_block_rows ? (uint64_t)(b_rowptr[col + 1] - b_rowptr[col]) : 0ull;
    }
};
struct RowWork {
    const uint64_t *first_pos;
    const uint32_t *a_rowptr;
    uint64_t *cum;
    __device__ void operator()(uint64_t r) const { cum[r] = first_pos[a_rowptr[r]]; }
};
struct RebaseOffsets {
    const uint64_t *in;
    uint64_t *out;
    uint64_t in_base, add;
    __device__ void operator()(uint64_t i) const { out[i] = in[i] - in_base + add; }
};
}  // namespace

void partition_rows(bmsp_matrix_s *A, bmsp_matrix_s *B, int parts, int64_t *bounds, hipStream_t st, uint64_t *total_out)
{
    if (parts < 1) fail(BMSP_ERR_INVALID, "parts must be >= 1");
    ensure_rowptr(A, st);
    ensure_rowptr(B, st);
    const uint64_t n_a = (uint64_t)A->block_num;
    const int64_t nbr = A->num_block_rows();
    DevBuf<uint64_t> first_pos(n_a + 1), cum((size_t)nbr + 1);
    device_exclusive_scan<uint64_t>(PanelFanOut{A->keys, B->rowptr, n_a, (uint32_t)B->num_block_rows()}, PtrOut<uint64_t>{first_pos.p},
                                    n_a + 1, st);
    device_for_each(RowWork{first_pos.p, A->rowptr, cum.p}, (uint64_t)nbr + 1, st);
    std::vector<uint64_t> h((size_t)nbr + 1);
    BMSP_HIP(hipStreamSynchronize(st));
    copy_d2h_staged(h.data(), cum.p, 8 * ((size_t)nbr + 1));  // (through the library's pinned buffers: `h` is freed on return, see runtime.h)
    const uint64_t total = h[(size_t)nbr];
    if (total_out) *total_out = total;
    bounds[0] = 0;
    int64_t r = 0;
    for (int p = 1; p < parts; p++) {
        // first block-row whose cumulative work reaches p/parts of the total (ties broken towards equal row counts)
        unsigned __int128 target = (unsigned __int128)total * (unsigned)p / (unsigned)parts;
        while (r < nbr && (unsigned __int128)h[(size_t)r] < target) r++;
        if (total == 0) r = nbr * p / parts;
        bounds[p] = r;
    }
    bounds[parts] = nbr;
    for (int p = 1; p <= parts; p++)
        if (bounds[p] < bounds[p - 1]) bounds[p] = bounds[p - 1];
}

bmsp_matrix_s *row_panel(bmsp_matrix_s *m, int64_t rb, int64_t re, hipStream_t st)
{
    const int64_t nbr = m->num_block_rows();
    if (rb < 0 || re < rb || re > nbr) fail(BMSP_ERR_INVALID, "panel [%lld,%lld) outside [0,%lld]", (long long)rb, (long long)re, (long long)nbr);
    ensure_rowptr(m, st);
    uint32_t b0 = read_back(m->rowptr + rb, st), b1 = read_back(m->rowptr + re, st);
    uint64_t o0 = read_back(m->offsets + b0, st), o1 = read_back(m->offsets + b1, st);
    bmsp_matrix_s *v = new bmsp_matrix_s();
    v->num_rows = m->num_rows; v->num_cols = m->num_cols; v->dtype = m->dtype; v->transposed = m->transposed;
    v->block_num = (int64_t)(b1 - b0);
    v->nnz = (int64_t)(o1 - o0);
    v->keys = m->keys + b0; v->bmps = m->bmps + b0; v->offsets = m->offsets + b0;
    v->values = m->values;  // offsets stay absolute into the parent's value array
    v->ownership = 2;
    v->view_block_begin = b0;
    v->view_values_end = (int64_t)o1;
    return v;
}

bmsp_matrix_s *concat_panels(int num_rows, int num_cols, int parts, const int64_t *block_nums, const int64_t *nnzs,
                             uint64_t *const *d_keys, uint64_t *const *d_bmps, uint64_t *const *d_offsets, void *const *d_values,
                             bmsp_dtype dtype, hipStream_t st)
{
    int64_t nb = 0, nz = 0;
    for (int p = 0; p < parts; p++) {
        if (block_nums[p] < 0 || nnzs[p] < 0) fail(BMSP_ERR_INVALID, "negative panel size");
        nb += block_nums[p];
        nz += nnzs[p];
    }
    bmsp_matrix_s *m = new bmsp_matrix_s();
    m->num_rows = num_rows; m->num_cols = num_cols; m->dtype = dtype; m->transposed = 0;
    m->block_num = nb; m->nnz = nz;
    const size_t es = dtype_size(dtype);
    m->keys = (uint64_t *)pool_alloc(8 * (size_t)(nb ? nb : 1));
    m->bmps = (uint64_t *)pool_alloc(8 * (size_t)(nb ? nb : 1));
    m->offsets = (uint64_t *)pool_alloc(8 * ((size_t)nb + 1));
    m->values = pool_alloc(es * (size_t)(nz ? nz : 1));
    int64_t bb = 0, zz = 0;
    for (int p = 0; p < parts; p++) {
        size_t cb = (size_t)block_nums[p], cz = (size_t)nnzs[p];
        if (cb) {
            BMSP_HIP(hipMemcpyAsync(m->keys + bb, d_keys[p], 8 * cb, hipMemcpyDeviceToDevice, st));
            BMSP_HIP(hipMemcpyAsync(m->bmps + bb, d_bmps[p], 8 * cb, hipMemcpyDeviceToDevice, st));
            uint64_t base = read_back(d_offsets[p], st);  // a panel's offsets may start anywhere
            device_for_each(RebaseOffsets{d_offsets[p], m->offsets + bb, base, (uint64_t)zz}, cb, st);
        }
        if (cz) BMSP_HIP(hipMemcpyAsync((char *)m->values + es * (size_t)zz, d_values[p], es * cz, hipMemcpyDeviceToDevice, st));
        bb += block_nums[p];
        zz += nnzs[p];
    }
    uint64_t term = (uint64_t)nz;
    BMSP_HIP(hipMemcpyAsync(m->offsets + nb, &term, 8, hipMemcpyHostToDevice, st));
    BMSP_HIP(hipStreamSynchronize(st));
    return m;
}

// One GPU, a product whose candidate block pairs exceed the 32-bit task range: the same row-panel decomposition the multi-GPU path
// uses, run panel after panel on this device and concatenated -- the reference has no such limit to hit only because it runs out of
// memory first (16-byte tasks, unfiltered list materialised).
void spgemm_paneled(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **Cout, int mode, int tc_version, int verbose, hipStream_t st,
                    bmsp_spgemm_stats *stats)
{
    uint64_t total = 0;
    int64_t one[2];
    partition_rows(A, B, 1, one, st, &total);
    int parts = (int)(total / (1ull << 31)) + 2;
    std::vector<int64_t> bounds;
    std::vector<bmsp_matrix_s *> panels;
    bmsp_spgemm_stats acc{};
    auto cleanup = [&]() { for (bmsp_matrix_s *m : panels) free_matrix(m); panels.clear(); };
    try {
        bounds.resize((size_t)parts + 1);
        partition_rows(A, B, parts, bounds.data(), st, nullptr);
        for (int p = 0; p < parts; p++) {
            std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> view(row_panel(A, bounds[(size_t)p], bounds[(size_t)p + 1], st), free_matrix);
            rm_hint_inherit(view.get(), A);
            bmsp_matrix_s *cp = nullptr;
            bmsp_spgemm_stats ps{};
            spgemm(view.get(), B, &cp, mode, tc_version, verbose, st, &ps);  // a single hub block-row beyond the range still fails here
            rm_hint_merge(A, view.get());
            panels.push_back(cp);
            acc.task_list_size += ps.task_list_size; acc.bmp_reduction += ps.bmp_reduction; acc.surviving_tasks += ps.surviving_tasks;
            acc.c_blocks += ps.c_blocks; acc.c_nnz += ps.c_nnz;
            for (int i = 0; i < 10; i++) acc.t_us[i] += ps.t_us[i];
            acc.sort_path = ps.sort_path; acc.mac_kernel = ps.mac_kernel; acc.mac_variant = ps.mac_variant;
        }
        std::vector<int64_t> bn, nz;
        std::vector<uint64_t *> k, b, o;
        std::vector<void *> v;
        for (bmsp_matrix_s *m : panels) { bn.push_back(m->block_num); nz.push_back(m->nnz); k.push_back(m->keys); b.push_back(m->bmps); o.push_back(m->offsets); v.push_back(m->values); }
        *Cout = concat_panels(A->num_rows, B->num_cols, parts, bn.data(), nz.data(), k.data(), b.data(), o.data(), v.data(), panels[0]->dtype, st);
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
    if (stats) *stats = acc;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(shard)
