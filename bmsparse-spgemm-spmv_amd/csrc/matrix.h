// matrix.h -- the device-resident bmSparse container behind bmsp_matrix_t
// (class bmSpMatrix<T>, reference include/bmSpMatrix.h:20-40).
#ifndef BMSP_MATRIX_H_
#define BMSP_MATRIX_H_

#include "runtime.h"
#include <vector>

namespace bmsp { uint64_t next_matrix_uid(); }

struct bmsp_matrix_s {
    // identity of this matrix's STRUCTURE: a fresh number at construction and after bmsp_matrix_invalidate(m, 1).  What one matrix remembers
    // about another (the row-merge hint below) is keyed by it -- not by a device pointer, which the pool hands out again
    uint64_t uid = bmsp::next_matrix_uid();
    int num_rows = 0, num_cols = 0;
    int64_t nnz = 0, block_num = 0;
    bmsp_dtype dtype = BMSP_F32;
    int transposed = 0;
    // the four public arrays of the reference container (device memory)
    uint64_t *keys = nullptr;     // block_num
    uint64_t *bmps = nullptr;     // block_num
    uint64_t *offsets = nullptr;  // block_num + 1 (last = nnz)
    void *values = nullptr;       // nnz elements of dtype
    int ownership = 1;            // 1: arrays belong to the pool / this handle, 2: borrowed
    // derived, built lazily and cached (never part of the reference's public state)
    uint32_t *rowptr = nullptr;   // dense block-row pointer, num_block_rows()+1 entries
    int64_t rowptr_rows = 0;
    int64_t max_row_blocks = -1;  // most blocks in one block-row (ensure_row_stats; bounds the SpGEMM's task segments without a read-back)
    // sweep plan of the SpMV (see spmv.hip): chunk descriptors
    uint32_t *spmv_chunks = nullptr;
    int64_t spmv_num_chunks = 0;
    int64_t spmv_plan_long = 0;
    int64_t spmv_full_tiles = 0;  // tiles with all 64 values stored (decides the sweep variant)
    size_t spmv_plan_off_cnt = 0, spmv_plan_off_carry = 0;
    // position cache of the value-stream SpMV: one 16-bit {tile slot in its batch, position in the tile} entry per stored value
    uint16_t *spmv_pos = nullptr;
    uint32_t *spmv_tinfo = nullptr;  // same allocation: per tile {block column, block-row inside its item's window} (the value-stream kernel's slot word)
    uint16_t *spmv_eoff = nullptr;   // same allocation: per tile, end of its values relative to its item's first value
    int64_t spmv_pos_base = 0, spmv_pos_count = 0;
    int spmv_pos_tried = 0;
    // (bitmap, value offset) of every block as one 16-byte record, for kernels that gather both (block-MAC): built lazily
    uint32_t *block_meta = nullptr;  // block_num x {bmp lo, bmp hi, offset in elements, 0}
    uint32_t *sym_recs = nullptr;    // right operands: block_num x {bitmap ROW-major (lo, hi), block column, rows the tile uses}: all the column-window passes (rowwindow.hip) read per candidate pair
    // column index of the long block-rows (right operands of the column-window passes): col_index[col_index_row[k] + c / col_index_gran] = first
    // tile of block-row k whose block column is >= c; col_index_row[k] = ~0 for block-rows of at most kIdxMinLen tiles
    uint32_t *col_mass = nullptr;  // prefix sums of this matrix's tiles per granule of col_index_gran block columns (ceil(cols / gran) + 1 entries): where the column-window passes cut
    uint32_t *col_index = nullptr, *col_index_row = nullptr;
    uint32_t col_index_gran = 0;
    int col_index_tried = 0;
    // fp16 matrices: every tile expanded to 64 halves in position order (128 B per block), for the K = 32 MFMA block-MAC: built lazily
    void *dense_tiles = nullptr;
    uint32_t *csr_rowptr = nullptr, *csr_ent = nullptr;  // fp32 operands of the row-sparse block-MAC: a row-major CSR copy (row pointer of num_rows + 1 entries; {column, value bits} per stored value)
    void *lane_tiles = nullptr;   // fp32 matrices: tiles in the lane order of the fp32 MFMA block-MAC (256 B per block): built lazily
    // SpGEMM row-merge paths: the right operand (keys pointer, block count) this matrix was last multiplied with and what that product
    // turned out to need (1 strip mode, 2 task-list mode, 3 the pipeline, 4 column windows) -- the next product of the pair goes there directly
    uint64_t rm_partner_uid = 0;
    int64_t rm_partner_blocks = 0;
    int rm_partner_mode = 0;
    int64_t rm_partner_cw_hash = 0;  // column-window passes: candidate pairs per hashed window that worked for the pair (0: the default)
    // a product made by bmsp_spgemm_symbolic keeps its sorted task list for bmsp_spgemm_numeric (T_7 alone on new operand values)
    uint64_t *sp_tasks = nullptr;
    uint32_t *sp_task_begin = nullptr, *sp_c_of_wave = nullptr;
    uint64_t sp_n_tasks = 0, sp_candidates = 0;
    int64_t sp_a_blocks = 0, sp_b_blocks = 0;
    // a product remembers which operand STRUCTURES it was formed from (struct_hash of A and B; 0 = not stamped: a view, adopted arrays): the
    // fast paths of bmsp_spgemm_numeric trust C only when the stamps match the operands it is given
    uint64_t sp_a_hash = 0, sp_b_hash = 0;
    uint64_t struct_hash = 0;  // this matrix: sum over its blocks of mix(key, bitmap), + dimensions; 0 = not computed (ensure_struct_hash)
    int values_finite = -1;
    int f32_exp_min = 255, f32_exp_max = 0;  // fp32: biased exponent range of the non-zero stored values (with values_finite)       // fp16 operands of the strip block-MAC: 1 = no inf / NaN stored (-1 = not looked yet)
    // a row-panel view points into its parent
    int64_t view_block_begin = 0;
    // sharded SpMV (comm.hip): this rank's panel view, kept for its cached sweep plan
    bmsp_matrix_s *shard_view = nullptr;
    int shard_world = 0, shard_rank = 0;
    std::vector<int64_t> shard_bounds;

    int64_t view_values_end = 0;  // row-panel views: end (in elements) of the panel's values inside the parent's array
    // elements addressable from `values`: offsets of a view stay absolute into the parent's value array
    int64_t values_extent() const { return view_values_end ? view_values_end : nnz; }
    int64_t num_block_rows() const { return ((int64_t)num_rows + 7) / 8; }
    int64_t num_block_cols() const { return ((int64_t)num_cols + 7) / 8; }
};

// one RCCL communicator (bmsp_comm_t); `comm` is an ncclComm_t
struct bmsp_comm_s {
    void *comm = nullptr;
    int rank = 0, world = 1, device = 0;
    void *xstream = nullptr;  // the exchange stream of the sharded SpGEMM (a round's broadcasts run on it while the next round multiplies)
    int loopback = 0;  // 1: no RCCL -- every panel is computed by this process on this device, "broadcast" = device copy (comm.hip)
};

namespace bmsp {

// A row-panel view inherits what its parent learned about a right operand (which row-merge mode the pair needs) and hands back what it
// learned itself: views are made per call, and without this every panel product would pay for a pass that does not fit (hub block-rows).
inline void rm_hint_inherit(bmsp_matrix_s *view, const bmsp_matrix_s *parent)
{
    view->rm_partner_uid = parent->rm_partner_uid; view->rm_partner_blocks = parent->rm_partner_blocks; view->rm_partner_mode = parent->rm_partner_mode;
    view->rm_partner_cw_hash = parent->rm_partner_cw_hash;
}
inline void rm_hint_merge(bmsp_matrix_s *parent, const bmsp_matrix_s *view)
{
    if (!view->rm_partner_uid) return;
    const bool same = parent->rm_partner_uid == view->rm_partner_uid;
    const int mode = same && parent->rm_partner_mode > view->rm_partner_mode ? parent->rm_partner_mode : view->rm_partner_mode;
    parent->rm_partner_uid = view->rm_partner_uid; parent->rm_partner_blocks = view->rm_partner_blocks; parent->rm_partner_mode = mode;
    if (view->rm_partner_cw_hash) parent->rm_partner_cw_hash = view->rm_partner_cw_hash;
}

inline size_t dtype_size(bmsp_dtype t) { return t == BMSP_F16 ? 2 : (t == BMSP_F32 ? 4 : 8); }

// host COO triples produced by the MatrixMarket parser
struct HostCoo {
    int num_rows = 0, num_cols = 0;
    std::vector<int> rows, cols;
    std::vector<double> vals;
};

// MatrixMarket coordinate reader (reference: src/bmSpMatrix.cu:111-161 ; CUSP reader semantics for
// pattern / symmetric, cusp/io/detail/matrix_market.inl:171-196,245-292).
void read_matrix_market(const std::string &path, HostCoo &out);

// builder (reference: src/bmSpMatrix.cu:163-216) from COO triples resident on the device
bmsp_matrix_s *build_from_device_coo(int num_rows, int num_cols, int64_t nnz, const int *d_rows, const int *d_cols,
                                     const double *d_vals, int transposed, bmsp_dtype dtype, hipStream_t st);

void ensure_rowptr(bmsp_matrix_s *m, hipStream_t st);
void ensure_row_stats(bmsp_matrix_s *m, hipStream_t st);
uint64_t ensure_struct_hash(bmsp_matrix_s *m, hipStream_t st);
void ensure_block_meta(bmsp_matrix_s *m, hipStream_t st);
void ensure_sym_recs(bmsp_matrix_s *m, hipStream_t st);
void ensure_col_index(bmsp_matrix_s *m, uint32_t gran, hipStream_t st);
void ensure_col_mass(bmsp_matrix_s *m, uint32_t gran, hipStream_t st);
constexpr uint32_t kIdxMinLen = 16;
int tile_product_selftest(hipStream_t st);
void ensure_dense_tiles(bmsp_matrix_s *m, hipStream_t st);
void ensure_lane_tiles(bmsp_matrix_s *m, hipStream_t st);
void ensure_finite_flag(bmsp_matrix_s *m, hipStream_t st);
bool mac_f32_mfma_usable(hipStream_t st);
bool mac_mfma32_supported(const bmsp_matrix_s *A, const bmsp_matrix_s *B);
bool mac_mfma32_b_dense(const bmsp_matrix_s *B);
int launch_mac_mfma32(const uint64_t *tasks, uint64_t n_tasks, const uint32_t *task_begin, const uint32_t *c_of_wave, bmsp_matrix_s *A,
                      bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st);  // returns the variant it launched (BMSP_MAC_*)
int mfma32_selftest(hipStream_t st);
int mfma_f32_selftest(hipStream_t st, bool cancel = false);
int mac_f32_exp_floor(hipStream_t st);
bool launch_mac_f32_mfma(const uint64_t *tasks, uint64_t n_tasks, const uint32_t *task_begin, const uint32_t *c_of_wave, bmsp_matrix_s *A,
                         bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st);
bool mac_strip_eligible(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, uint64_t candidates, uint64_t n_tasks, hipStream_t st);
bool mac_strip_operands_ok(bmsp_matrix_s *A, bmsp_matrix_s *B, hipStream_t st);
bool mac_strip_fits_c(bmsp_matrix_s *C, hipStream_t st);
uint32_t mac_strip_row_cap();
int launch_mac_strip(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, int tc_version, hipStream_t st);  // returns the BMSP_MAC_* variant that ran (strip, or row-sparse for V15 numerics on nearly empty tiles)
bool mac_structure_numeric_ok(bmsp_matrix_s *A, bmsp_matrix_s *B, int tc_version, hipStream_t st);  // a numeric stage that works from C's structure alone exists for these operands
void ensure_csr32(bmsp_matrix_s *m, hipStream_t st);
bool mac_rowsparse_applies(bmsp_matrix_s *A, bmsp_matrix_s *B, int tc_version, hipStream_t st);
void launch_mac_rowsparse(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st);
bool mac_rowsparse_fits_c(bmsp_matrix_s *C, hipStream_t st);  // every block-row of C within the row-sparse kernel's table (768 tiles)
bool rowmerge_symbolic(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, uint32_t row_cap,
                       uint64_t *surviving, uint64_t *candidates, hipStream_t st);
bool rowmerge_tasklist(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, DevBuf<uint64_t> &tasks,
                       DevBuf<uint32_t> &task_begin, DevBuf<uint32_t> &c_of_wave, uint64_t *n_tasks, hipStream_t st);
bool rowmerge_windowed(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, DevBuf<uint64_t> &tasks,
                       DevBuf<uint32_t> &task_begin, DevBuf<uint32_t> &c_of_wave, uint64_t *n_tasks, hipStream_t st);
void matrix_to_coo_host(bmsp_matrix_s *m, int *rows, int *cols, double *vals, hipStream_t st);
void matrix_to_coo_device(bmsp_matrix_s *m, uint64_t *d_rc, double *d_vals, hipStream_t st);  // every stored value as ((row << 32) | col, value), sorted by (row, col)
void matrix_to_coo_device_split(bmsp_matrix_s *m, int *d_rows, int *d_cols, double *d_vals, hipStream_t st);
void matrix_to_csr_device(bmsp_matrix_s *m, int *d_row_offsets, int *d_cols, double *d_vals, hipStream_t st);
void matrix_compare_device(bmsp_matrix_s *m, int64_t nnz, const int *d_rows, const int *d_cols, const double *d_vals, double *mean_rel_err,
                           int64_t *missing, hipStream_t st);
bmsp_matrix_s *build_from_device_csr(int num_rows, int num_cols, int64_t nnz, const int *d_row_offsets, const int *d_cols, const double *d_vals,
                                     int transposed, bmsp_dtype dtype, hipStream_t st);
void free_matrix(bmsp_matrix_s *m);
// eager construction of the cached derived structures (bmsp_matrix_prepare)
void prepare_spmv(bmsp_matrix_s *m, hipStream_t st);
void prepare_spgemm_operand(bmsp_matrix_s *m, hipStream_t st);

void spmv(bmsp_matrix_s *A, const void *v, void *u, int variant, hipStream_t st, int64_t row_lo = 0, int64_t row_hi = -1);
void spmv_launch_info(bmsp_matrix_s *A, int variant, hipStream_t st, char *kernel, size_t kernel_cap, int64_t *compulsory, int64_t *format_bytes);
void spmm(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st);
void spgemm(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **C, int mode, int tc_version, int verbose, hipStream_t st,
            bmsp_spgemm_stats *stats, bool structure_only = false);
void spgemm_numeric(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, int tc_version, hipStream_t st, bmsp_spgemm_stats *stats);
template <typename T>
struct PingPong;
// stable sort of (key, task) pairs inside the runs of equal (key >> jbits): the SpGEMM's segmented path
// false = some block-row has more tasks than the LDS paths hold; nothing was modified and the caller sorts globally
bool segsort_tasks_by_column(PingPong<uint64_t> &keys, PingPong<uint64_t> &vals, uint64_t n, int jbits, hipStream_t st, uint64_t max_seg_bound = 0,
                             uint64_t seg_count_bound = 0, int *long_mode = nullptr);  // *long_mode: BMSP_SORT_LONG_* (0: no long segment)
void segsort_check_violation();
void segsort_u64(uint64_t *d_keys, void *d_vals, int val_bytes, int64_t n, const int *d_segs, int64_t num_segs,
                 hipStream_t st);

// multi-GPU (comm.hip)
void comm_unique_id(void *id128);
bmsp_comm_s *comm_init(const void *id128, int world, int rank);
bmsp_comm_s *comm_init_from_env();
bmsp_comm_s *comm_init_loopback(int world);
void shard_layout(int parts, const int64_t *block_nums, const int64_t *nnzs, int64_t *block_start, int64_t *value_start);
void shard_row_slices(int num_rows, int parts, const int64_t *bounds, int64_t *row_start, int64_t *row_count);
void invalidate_matrix(bmsp_matrix_s *m, int structure_changed);
void comm_free(bmsp_comm_s *c);
void spgemm_sharded(bmsp_comm_s *c, bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **C, int mode, int tc_version, int verbose, hipStream_t st,
                    bmsp_spgemm_stats *stats, bmsp_shard_stats *sh, int gather = 1, int rounds = 0);
void spmv_sharded(bmsp_comm_s *c, bmsp_matrix_s *A, const void *x, void *y, int variant, hipStream_t st, bmsp_shard_stats *sh);

}  // namespace bmsp
#endif
