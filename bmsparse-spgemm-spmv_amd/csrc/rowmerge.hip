// rowmerge.hip -- the symbolic stages of a product whose block-rows fit a wave's LDS: T_3 (expansion), T_4 (bitmap filter), T_5 (sort),
// T_6 (C's keys) and T_9 (C's bitmaps) in ONE kernel, row by row, without a task list.
//
// Reference: bmSparse_mult (src/bmSparse_SPGEMM.cu:849-1164) materialises every candidate block pair (:884-932), filters it by
// multiplication_checker (:742-757), sorts the survivors by C key (:963-1024), reduces the keys (:1040-1062) and ORs the boolean tile
// products (bmp_calculator, :787-810) -- five passes over task-sized arrays in HBM (FEM-like product: 40 M candidates, 1.46 ms of 2.0 ms).
// What C holds depends only on the SET of surviving pairs per block-row of A, so a wave that owns block-row i can form it in LDS:
// every candidate (A(i,k), B(k,j)) that passes the filter ORs its tile-product bitmap into a hash table keyed by j; the table is
// compacted, ranked by column (C's key order inside the row) and written once.  Same keys, same bitmaps, same value offsets as the
// pipeline's; the numeric stage that follows (blockmac_strip.hip) recomputes which tiles meet from the operands, as it always did.
//
// Limits (the caller keeps the expand-sort-compress pipeline otherwise): at most kRowCap distinct C tiles per block-row.
#include "matrix.h"
#include "prims.hip.h"
#include "bmsp_bits.h"

namespace bmsp {
namespace {

constexpr int kHashBits = 9;
constexpr int kHash = 1 << kHashBits;  // slots per block-row
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr uint32_t kRowCap = 384;      // most distinct C tiles per block-row (a batch of 64 candidates may pass it by 63 before the pass stops)

struct RowMergeArgs {
    const uint64_t *a_keys, *a_bmps;
    const uint32_t *a_rowptr;
    const uint64_t *b_keys, *b_bmps;
    const uint32_t *b_rowptr;
    uint32_t block_rows, b_block_rows;
    uint32_t row_cap;         // distinct C tiles per block-row the caller accepts (<= kRowCap)
    const uint32_t *tmp_off;  // block_rows + 1: first scratch slot of block-row i
    uint32_t *t_cols;         // scratch: C's block columns of row i, ascending
    uint64_t *t_bmps;         //          and their bitmaps
    uint32_t *cnt;            // block_rows (+ 1, the scan reads one past): C tiles of row i
    uint32_t *surv;           // block_rows: candidate pairs of row i that passed the filter
    uint32_t *overflow;
};

struct alignas(16) RowLds {
    uint32_t hk[kHash];  // block column; kEmpty = free
    uint64_t hb[kHash];  // OR of the tile-product bitmaps
    uint32_t abeg[64];   // chunk of 64 A tiles: first tile of B's block-row k
    uint32_t aend[64];   //                      ... and one past its last
    uint64_t abmp[64];
};  // 7 KB per wave: five workgroups per CU

__global__ __launch_bounds__(kThreads) void rowmerge_symbolic_kernel(RowMergeArgs g)
{
    __shared__ RowLds lds_all[4];
    const int w = wave_id(), lane = lane_id();
    RowLds &S = lds_all[w];
    // XCD-aware order: the workgroups of one XCD take a contiguous eighth of the block-rows (neighbouring rows read the same block-rows of B)
    uint32_t wg;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        wg = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    const uint32_t row = wg * 4 + (uint32_t)w;
    if (row >= g.block_rows) return;
    const uint32_t a0 = g.a_rowptr[row], a1 = g.a_rowptr[row + 1];
    // (a block-row beyond the cap makes the whole pass void: the waves that start after it was seen leave at once)
    if (a0 == a1 || __builtin_nontemporal_load(g.overflow) != 0u) {
        if (lane == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; }
        return;
    }
    for (uint32_t s = (uint32_t)lane; s < (uint32_t)kHash; s += 64) { S.hk[s] = kEmpty; S.hb[s] = 0ull; }
    uint32_t n = 0;       // distinct columns so far (wave-uniform)
    uint32_t surv = 0;    // this lane's surviving pairs
    bool over = false;
    for (uint32_t base = a0; base < a1 && !over; base += 64) {
        const uint32_t a = base + (uint32_t)lane;
        const bool on = a < a1;
        const uint32_t k = on ? key_col(g.a_keys[a]) : 0u;
        uint32_t bb = 0, be = 0;
        if (on && k < g.b_block_rows) { bb = g.b_rowptr[k]; be = g.b_rowptr[k + 1]; }
        __builtin_amdgcn_wave_barrier();
        S.abeg[lane] = bb; S.aend[lane] = be; S.abmp[lane] = on ? g.a_bmps[a] : 0ull;
        __builtin_amdgcn_wave_barrier();
        const uint32_t na = min(64u, a1 - base);
        // four A tiles at a time, 16 lanes each: the lanes of a group walk B's block-row k of their tile 16 tiles per step (the words of
        // the next step are requested before the current ones are used)
        for (uint32_t g4 = 0; g4 < na && !over; g4 += 4) {
            const uint32_t u = g4 + (uint32_t)(lane >> 4);
            const uint32_t end = u < na ? S.aend[u] : 0u;
            const uint64_t abm = S.abmp[min(u, 63u)];
            uint32_t t = (u < na ? S.abeg[u] : 0u) + (uint32_t)(lane & 15);
            uint32_t j = 0;
            uint64_t bbm = 0;
            if (t < end) { j = key_col(g.b_keys[t]); bbm = g.b_bmps[t]; }
            while (__any(t < end)) {
                if (n > g.row_cap) { over = true; break; }
                const bool live = t < end;
                const uint32_t tn = t + 16u;
                uint32_t jn = 0;
                uint64_t bn = 0;
                if (tn < end) { jn = key_col(g.b_keys[tn]); bn = g.b_bmps[tn]; }
                const bool keep = live && !tile_product_empty(abm, bbm);  // multiplication_checker (:742-757)
                surv += keep ? 1u : 0u;
                bool fresh = false;
                if (keep) {
                    // bmp_calculator (:787-810); two full tiles: a full one
                    const uint64_t prod = (abm & bbm) == ~0ull ? ~0ull : tile_product_bmp(abm, bbm);
                    uint32_t slot = (j * 0x9E3779B1u) >> (32 - kHashBits);
                    for (;;) {
                        const uint32_t old = atomicCAS(&S.hk[slot], kEmpty, j);
                        if (old == kEmpty || old == j) {
                            fresh = old == kEmpty;
                            atomicOr((unsigned long long *)&S.hb[slot], (unsigned long long)prod);
                            break;
                        }
                        slot = (slot + 1u) & (uint32_t)(kHash - 1);
                    }
                }
                n += (uint32_t)__popcll(__ballot(fresh));
                t = tn; j = jn; bbm = bn;
            }
        }
    }
    if (over || n > g.row_cap) {
        if (lane == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; atomicOr(g.overflow, 1u); }
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // compaction in place: round r reads slots [64 r, 64 r + 64) before it writes, and writes only in front of them
    uint32_t m = 0;
    for (uint32_t r = 0; r < (uint32_t)kHash; r += 64) {
        const uint32_t key = S.hk[r + (uint32_t)lane];
        const uint64_t bm = S.hb[r + (uint32_t)lane];
        const uint64_t bal = __ballot(key != kEmpty);
        __builtin_amdgcn_wave_barrier();
        if (key != kEmpty) {
            const uint32_t pos = m + (uint32_t)__popcll(bal & lanemask_lt());
            S.hk[pos] = key; S.hb[pos] = bm;
        }
        m += (uint32_t)__popcll(bal);
        __builtin_amdgcn_wave_barrier();
    }
    // (m == n <= kRowCap + 63 < kHash - 4) pad to a multiple of four for the 16-byte reads below
    if (lane < 4) S.hk[m + (uint32_t)lane] = kEmpty;
    __builtin_amdgcn_wave_barrier();
    // rank by column: n is small (a block-row of C), every lane counts the keys below its own; the reads are wave-wide broadcasts
    const uint32_t out0 = g.tmp_off[row];
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    for (uint32_t p0 = 0; p0 < m; p0 += 64) {
        const uint32_t p = p0 + (uint32_t)lane;
        const uint32_t key = p < m ? S.hk[p] : kEmpty;
        uint32_t rank = 0;
        for (uint32_t q = 0; q < m; q += 4) {
            const u32x4v v = *(const u32x4v *)&S.hk[q];
            rank += (v[0] < key ? 1u : 0u) + (v[1] < key ? 1u : 0u) + (v[2] < key ? 1u : 0u) + (v[3] < key ? 1u : 0u);
        }
        if (p < m) {
            g.t_cols[out0 + rank] = key;
            g.t_bmps[out0 + rank] = S.hb[p];
        }
    }
    surv = wave_sum(surv);
    if (lane == 0) { g.cnt[row] = m; g.surv[row] = surv; }  // (per-row results: one atomic pair per wave would serialise at the memory side)
}

// scratch slots of block-row i: its candidate pairs (T_2's scan), capped by what the kernel accepts
struct RowSlotsIn {
    const uint64_t *first_pos;
    const uint32_t *a_rowptr;
    uint64_t rows;
    uint32_t cap;
    __device__ uint32_t operator()(uint64_t i) const
    {
        if (i >= rows) return 0u;
        const uint64_t c = first_pos[a_rowptr[i + 1]] - first_pos[a_rowptr[i]];
        return (uint32_t)(c < (uint64_t)cap ? c : (uint64_t)cap);
    }
};
struct CntIn {
    const uint32_t *cnt;
    uint64_t rows;
    __device__ uint32_t operator()(uint64_t i) const { return i < rows ? cnt[i] : 0u; }
};
struct CntSurvIn {
    const uint32_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return (uint64_t)p[i]; }
};
struct PublishStats {
    const unsigned long long *acc;  // [0] surviving pairs, [1] most C tiles in a block-row
    const uint32_t *overflow;
    uint64_t *h_surviving, *h_max_over;
    __device__ void operator()(uint64_t) const
    {
        *h_surviving = (uint64_t)acc[0];
        *h_max_over = ((uint64_t)*overflow << 32) | (uint64_t)acc[1];
    }
};

// scratch -> C's own arrays: one wave per block-row
__global__ __launch_bounds__(kThreads) void rowmerge_emit_kernel(const uint32_t *__restrict__ tmp_off, const uint32_t *__restrict__ t_cols,
                                                                  const uint64_t *__restrict__ t_bmps, const uint32_t *__restrict__ c_rowptr,
                                                                  uint32_t block_rows, uint64_t *__restrict__ c_keys, uint64_t *__restrict__ c_bmps)
{
    const uint32_t row = blockIdx.x * 4 + (uint32_t)wave_id();
    if (row >= block_rows) return;
    const uint32_t c0 = c_rowptr[row], n = c_rowptr[row + 1] - c0, t0 = tmp_off[row];
    for (uint32_t p = (uint32_t)lane_id(); p < n; p += 64) {
        c_keys[c0 + p] = key_make(row, t_cols[t0 + p]);
        c_bmps[c0 + p] = t_bmps[t0 + p];
    }
}

}  // namespace

// C's structure (keys, bitmaps, block-row pointer, block count) of A x B by the row-merge pass.  first_pos = T_2's exclusive scan of the
// fan-out per A tile (n_a + 1 entries), total = its last element.  false: some block-row of C holds more tiles than the pass accepts
// (nothing of C was allocated; the caller runs the pipeline).  true: C->keys / bmps / rowptr / block_num / max_row_blocks are set,
// *surviving = candidate pairs that passed the bitmap filter.
bool rowmerge_symbolic(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, uint32_t row_cap,
                       uint64_t *surviving, hipStream_t st)
{
    row_cap = std::min(row_cap, kRowCap);
    const uint64_t rows = (uint64_t)A->num_block_rows();
    if (rows == 0 || rows >= (1ull << 31) || total == 0 || total >= (1ull << 32)) return false;
    ensure_rowptr(A, st);
    ensure_rowptr(B, st);
    const uint64_t slots = std::min<uint64_t>(total, rows * (uint64_t)row_cap);
    if (slots >= (1ull << 31)) return false;
    DevBuf<uint32_t> tmp_off(rows + 1), cnt(rows + 1), surv_row(rows), t_cols(slots);
    DevBuf<uint64_t> t_bmps(slots);
    DevBuf<unsigned long long> acc(3);  // [0] surviving pairs, [1] most C tiles in a block-row, [2] overflow flag
    BMSP_HIP(hipMemsetAsync(acc.p, 0, 24, st));
    device_exclusive_scan<uint32_t>(RowSlotsIn{first_pos, A->rowptr, rows, row_cap}, PtrOut<uint32_t>{tmp_off.p}, rows + 1, st);
    RowMergeArgs g{};
    g.a_keys = A->keys; g.a_bmps = A->bmps; g.a_rowptr = A->rowptr;
    g.b_keys = B->keys; g.b_bmps = B->bmps; g.b_rowptr = B->rowptr;
    g.block_rows = (uint32_t)rows; g.b_block_rows = (uint32_t)B->num_block_rows(); g.row_cap = row_cap;
    g.tmp_off = tmp_off.p; g.t_cols = t_cols.p; g.t_bmps = t_bmps.p; g.cnt = cnt.p;
    g.surv = surv_row.p; g.overflow = (uint32_t *)(acc.p + 2);
    hipLaunchKernelGGL(rowmerge_symbolic_kernel, dim3((uint32_t)((rows + 3) / 4)), dim3(kThreads), 0, st, g);
    BMSP_CHECK_LAUNCH();
    uint32_t *c_rowptr = (uint32_t *)pool_alloc(sizeof(uint32_t) * (size_t)(rows + 1));
    HostScalar<uint32_t> c_size_h;
    HostScalar<uint64_t> surv_h, mo_h;
    uint32_t c_size = 0;
    uint64_t mo = 0;
    try {
        device_exclusive_scan<uint32_t>(CntIn{cnt.p, rows}, PtrOutTotal<uint32_t>{c_rowptr, rows, c_size_h.dev()}, rows + 1, st);
        device_max_sum(CntSurvIn{surv_row.p}, rows, (unsigned long long *)nullptr, acc.p, st);
        device_max_sum(CntSurvIn{cnt.p}, rows, acc.p + 1, (unsigned long long *)nullptr, st);
        device_for_each(PublishStats{acc.p, g.overflow, surv_h.dev(), mo_h.dev()}, 1, st);
        c_size = c_size_h.wait(st);
        *surviving = surv_h.wait(st);
        mo = mo_h.wait(st);
    } catch (...) {
        pool_free(c_rowptr);
        throw;
    }
    if (mo >> 32) {
        pool_free(c_rowptr);
        return false;
    }
    C->block_num = c_size;
    C->rowptr = c_rowptr;
    C->rowptr_rows = (int64_t)rows;
    C->max_row_blocks = (int64_t)(uint32_t)mo;
    C->keys = (uint64_t *)pool_alloc(8 * (size_t)(c_size ? c_size : 1));
    C->bmps = (uint64_t *)pool_alloc(8 * (size_t)(c_size ? c_size : 1));
    if (c_size) {
        hipLaunchKernelGGL(rowmerge_emit_kernel, dim3((uint32_t)((rows + 3) / 4)), dim3(kThreads), 0, st, tmp_off.p, t_cols.p, t_bmps.p, c_rowptr,
                           (uint32_t)rows, C->keys, C->bmps);
        BMSP_CHECK_LAUNCH();
    }
    return true;
}

}  // namespace bmsp
