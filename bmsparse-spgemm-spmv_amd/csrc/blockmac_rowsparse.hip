// blockmac_rowsparse.hip -- T_7 with V15's numerics for fp32 operands whose tiles are nearly empty: only the scalar products that exist.
//
// Reference: multiplyV15 (src/bmSparse_SPGEMM.cu:204-291) runs, per task, sum = fmaf(A(i, kk), B(kk, j), sum) for kk = 0 .. 7 over the
// EXPANDED tiles (absent elements are zeros, :152-162), task after task in ascending A tile (:269-273).  On the FEM-like product a
// tile holds 3.7 of 64 values: 2.85 of a task's 512 multiply-adds have two stored operands; the others add an exact zero.  The strip
// kernel (blockmac_strip.hip) does all 512 on the matrix cores and moves 256 bytes per operand tile for them: 1.19 ms for 7.0e7 useful
// products.  For a C element (i, j) V15's order is: A tiles ascending, inside a tile kk ascending -- i.e. ascending k = 8 * (block column
// of the A tile) + kk over row i of A -- each term a fused multiply-add into the element's accumulator.  That is the row-wise product
//   for every stored A(i, k), k ascending:   for every stored B(k, j):   C(i, j) = fmaf(A(i, k), B(k, j), C(i, j))
// with the reference's zero terms left out.  A zero term changes a sum only when the sum is -0 (fmaf(0, b, -0) = +0), and a sum is -0
// only after a product has underflowed: the launcher takes this kernel only for operands whose exponent ranges keep every product of
// stored values a normal number (the precondition of the fp32 matrix-core kernel, mac_strip_operands_ok), and for finite values (0 x inf).
//
// One wave per block-row of C; C's structure is given (T_3 ... T_9 have fixed it).  Accumulators: one float per stored value of the
// block-row, in LDS (a window of C tiles of at most kRsAcc values at a time; a block-row that holds more is walked once per window);
// C's block columns of the block-row in an LDS hash table.  The two halves of the wave take the block-row's eight rows in turn (rows of
// different index never meet in a C element); a half walks its row of A entry by entry (32 entries and the bounds of their rows of B
// fetched together, handed round by shuffles), 32 lanes = 32 entries of B's row k: column -> C tile (hash) -> slot (rank in C's
// bitmap) -> one LDS read-modify-write.  Inside an iteration the 32 products of a half go to 32 different columns of one row: no
// conflicts, no atomics.  Operands: row-major CSR copies (row pointer, column, value) derived once per matrix, like the dense copies
// the matrix-core kernels read.
// (Two forms that walked the tiles themselves -- one A tile per step, lane = a tile of B's block-row k, the A tile's values in a scalar
// loop -- measured 1171 and 1084 us on the FEM-like product, the second with B's values parked in LDS: 96 products per step spread over
// eleven serialized read-modify-write iterations at one lane in seven.)
#include "mac_common.hip.h"

namespace bmsp {
namespace {

constexpr int kRsHashBits = 9, kRsHash = 1 << kRsHashBits;
constexpr uint32_t kRsRowCap = 256;   // C tiles per block-row (the strip kernels' limit: mac_strip_row_cap)
constexpr uint32_t kRsAcc = 2048;     // accumulators per window
constexpr uint32_t kRsEmpty = 0xffffffffu;

struct RsArgs {
    const uint32_t *a_rowptr, *a_cols;  // A row-major: CSR row pointer (num_rows + 1), column of every value
    const float *a_vals;
    const uint32_t *b_rowptr, *b_cols;
    const float *b_vals;
    uint32_t a_rows, b_rows;
    const uint64_t *c_keys, *c_bmps, *c_offs;
    const uint32_t *c_rowptr;  // per block-row
    float *c_vals;
    uint32_t block_rows;
};

struct alignas(16) RsLds {
    uint32_t hk[kRsHash];        // block column of a C tile of the block-row; kRsEmpty = free
    uint16_t cr[kRsHash];        // ... its index inside the block-row
    uint64_t cb[kRsRowCap];      // C's bitmaps
    uint32_t co[kRsRowCap + 1];  // first value of every C tile relative to the block-row's first
    float acc[kRsAcc];
};  // 14 KB: eleven one-wave workgroups per CU

__device__ __forceinline__ uint32_t rs_hash(uint32_t j) { return (j * 0x9E3779B1u) >> (32 - kRsHashBits); }

__global__ __launch_bounds__(64) void block_mac_rowsparse_kernel(RsArgs g)
{
    __shared__ RsLds S;
    const int lane = (int)threadIdx.x;
    // XCD-aware order: the workgroups of one XCD take a contiguous eighth of the block-rows (neighbours read the same rows of B)
    uint32_t brow;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        brow = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    if (brow >= g.block_rows) return;
    const uint32_t c0 = g.c_rowptr[brow], m = g.c_rowptr[brow + 1] - c0;
    if (m == 0 || m > kRsRowCap) return;  // (the launcher admits no product with a longer block-row of C)
    // ---- C's block-row: column table, bitmaps, value offsets ----
    for (uint32_t s = (uint32_t)lane; s < (uint32_t)kRsHash; s += 64) S.hk[s] = kRsEmpty;
    __builtin_amdgcn_wave_barrier();
    const uint64_t vbase = g.c_offs[c0];
    for (uint32_t r = (uint32_t)lane; r < m; r += 64) {
        const uint32_t j = key_col(g.c_keys[c0 + r]);
        uint32_t slot = rs_hash(j);
        while (atomicCAS(&S.hk[slot], kRsEmpty, j) != kRsEmpty) slot = (slot + 1u) & (uint32_t)(kRsHash - 1);  // (the columns of a block-row are distinct)
        S.cr[slot] = (uint16_t)r;
        S.cb[r] = g.c_bmps[c0 + r];
        S.co[r] = (uint32_t)(g.c_offs[c0 + r] - vbase);
    }
    if (lane == 0) S.co[m] = (uint32_t)(g.c_offs[c0 + m] - vbase);
    __builtin_amdgcn_wave_barrier();

    const int half = lane >> 5, l32 = lane & 31, hbase = lane & 32;
    for (uint32_t w0 = 0; w0 < m;) {
        // ---- the window: C tiles [w0, w1) with at most kRsAcc values (a tile holds at most 64) ----
        const uint32_t o0 = S.co[w0];
        uint32_t w1 = w0;
        for (uint32_t base = w0; base < m; base += 64) {
            const uint32_t r = base + (uint32_t)lane;
            const bool ok = r < m && S.co[r + 1] - o0 <= kRsAcc;
            const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));  // the offsets ascend: the tiles that fit are a prefix
            w1 = base + cnt;
            if (cnt < 64u) break;
        }
        const uint32_t nv = S.co[w1] - o0;
        for (uint32_t e = (uint32_t)lane; e < nv; e += 64) S.acc[e] = 0.f;
        __builtin_amdgcn_wave_barrier();
        // ---- the eight rows of the block-row, two at a time (one per half of the wave) ----
        for (int s4 = 0; s4 < 4; s4++) {
            const uint32_t i = (uint32_t)(2 * s4 + half);  // row inside the tiles
            const uint32_t row = brow * 8u + i;
            uint32_t pa0 = 0, pa1 = 0;
            if (row < g.a_rows) { pa0 = g.a_rowptr[row]; pa1 = g.a_rowptr[row + 1]; }
            // 32 entries of A's row per half at a time: column k, value, and the bounds of B's row k
            for (uint32_t pbase = pa0; __any(pbase < pa1); pbase += 32) {
                const uint32_t p = pbase + (uint32_t)l32;
                uint32_t k_l = 0, b0_l = 0, b1_l = 0;
                float a_l = 0.f;
                if (p < pa1) {
                    k_l = g.a_cols[p]; a_l = g.a_vals[p];
                    if (k_l < g.b_rows) { b0_l = g.b_rowptr[k_l]; b1_l = g.b_rowptr[k_l + 1]; }
                }
                const uint32_t nt = pbase < pa1 ? min(32u, pa1 - pbase) : 0u;  // entries of this half in the chunk
                const uint32_t nt_max = max(nt, (uint32_t)__shfl_xor((int)nt, 32, kWave));
                // entry t of the chunk: its row of B, 32 entries per half and iteration; the first 32 entries of entry t + 1 travel meanwhile
                auto fetch = [&](uint32_t t, uint32_t &e0, uint32_t &e1, float &a, uint32_t &col, float &b) {
                    const int src = hbase + (int)min(t, 31u);
                    e0 = (uint32_t)__shfl((int)b0_l, src, kWave);
                    e1 = (uint32_t)__shfl((int)b1_l, src, kWave);
                    a = __shfl(a_l, src, kWave);
                    if (t >= nt) e1 = e0;
                    col = 0; b = 0.f;
                    const uint32_t e = e0 + (uint32_t)l32;
                    if (e < e1) { col = g.b_cols[e]; b = g.b_vals[e]; }
                };
                uint32_t e0, e1, col, e0n = 0, e1n = 0, coln = 0;
                float a, b, an = 0.f, bn = 0.f;
                fetch(0u, e0, e1, a, col, b);
                for (uint32_t t = 0; t < nt_max; t++) {
                    if (t + 1 < nt_max) fetch(t + 1, e0n, e1n, an, coln, bn);
                    for (uint32_t eb = e0; __any(eb < e1); eb += 32) {
                        if (eb != e0) {  // a row of B beyond 32 entries: the later ones are fetched here
                            const uint32_t e = eb + (uint32_t)l32;
                            col = 0; b = 0.f;
                            if (e < e1) { col = g.b_cols[e]; b = g.b_vals[e]; }
                        }
                        if (eb + (uint32_t)l32 < e1) {
                            const uint32_t jb = col >> 3, j = col & 7u;
                            uint32_t slot = rs_hash(jb);
                            for (;;) {
                                const uint32_t key = S.hk[slot];
                                if (key == jb) {
                                    const uint32_t r = (uint32_t)S.cr[slot];
                                    if (r >= w0 && r < w1) {
                                        const uint64_t cbm = S.cb[r];
                                        const uint32_t pc = 8u * i + j;
                                        if ((cbm >> (63u - pc)) & 1ull) {
                                            const uint32_t ci = S.co[r] - o0 + (pc ? (uint32_t)__popcll(cbm >> (64u - pc)) : 0u);
                                            S.acc[ci] = __builtin_fmaf(a, b, S.acc[ci]);
                                        }
                                    }
                                    break;
                                }
                                if (key == kRsEmpty) break;  // (cannot happen for a product of stored values: C's structure holds its tile)
                                slot = (slot + 1u) & (uint32_t)(kRsHash - 1);
                            }
                        }
                    }
                    e0 = e0n; e1 = e1n; a = an; col = coln; b = bn;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the window's values leave in one run ----
        float *const dst = g.c_vals + vbase + o0;
        for (uint32_t e = (uint32_t)lane; e < nv; e += 64) dst[e] = S.acc[e];
        __builtin_amdgcn_wave_barrier();
        w0 = w1;
    }
}

struct RsRowPtr {  // CSR row pointer from the sorted (row << 32 | column) words: entry r = first word of a row >= r
    const uint64_t *rc;
    uint64_t n;
    uint32_t num_rows;
    uint32_t *rowptr;
    __device__ void operator()(uint64_t i) const
    {
        const uint32_t hi = i < n ? (uint32_t)(rc[i] >> 32) : num_rows;
        const uint32_t lo = i ? (uint32_t)(rc[i - 1] >> 32) + 1u : 0u;
        for (uint32_t r = lo; r <= hi; r++) rowptr[r] = (uint32_t)i;
    }
};
struct RsColsVals {
    const uint64_t *rc;
    const double *dv;
    uint32_t *cols;
    float *vals;
    __device__ void operator()(uint64_t i) const { cols[i] = (uint32_t)rc[i]; vals[i] = (float)dv[i]; }  // (the values were floats: exact)
};

}  // namespace

// row-major CSR copy of an fp32 matrix (row pointer, columns, values), whatever its tile layout: built once per matrix, like the dense
// tile copies; dropped by bmsp_matrix_invalidate
void ensure_csr32(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->csr_rowptr) return;
    const uint64_t n = (uint64_t)m->nnz;
    m->csr_rowptr = (uint32_t *)pool_alloc(4 * ((size_t)m->num_rows + 1));
    m->csr_cols = (uint32_t *)pool_alloc(4 * (size_t)(n ? n : 1));
    m->csr_vals = (float *)pool_alloc(4 * (size_t)(n ? n : 1));
    DevBuf<uint64_t> rc(n ? n : 1);
    DevBuf<double> dv(n ? n : 1);
    matrix_to_coo_device(m, rc.p, dv.p, st);
    device_for_each(RsRowPtr{rc.p, n, (uint32_t)m->num_rows, m->csr_rowptr}, n + 1, st);
    if (n) device_for_each(RsColsVals{rc.p, dv.p, m->csr_cols, m->csr_vals}, n, st);
    BMSP_HIP(hipStreamSynchronize(st));  // rc / dv go back to the pool
}

// fp32 operands of nearly empty tiles (at most 16 stored values per tile on average on both sides): the scalar products that exist are
// a few per cent of what the matrix-core kernel multiplies.  The caller has established mac_strip_operands_ok (finite values, every
// product a normal number) and a block-row of C of at most mac_strip_row_cap() tiles.  BMSP_MAC_ROWSPARSE=0/1: never / whatever the fill.
bool mac_rowsparse_applies(bmsp_matrix_s *A, bmsp_matrix_s *B)
{
    const char *e = getenv("BMSP_MAC_ROWSPARSE");
    if (e && e[0] == '0') return false;
    if (A->dtype != BMSP_F32 || B->dtype != BMSP_F32) return false;
    if ((uint64_t)A->nnz >= (1ull << 32) || (uint64_t)B->nnz >= (1ull << 32)) return false;
    if (A->view_values_end || B->view_values_end || A->ownership == 2 || B->ownership == 2) return false;  // (row-panel views: no copy of their own)
    if (e && e[0] == '1') return true;
    return A->nnz <= 16 * A->block_num && B->nnz <= 16 * B->block_num;
}

void launch_mac_rowsparse(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st)
{
    ensure_csr32(A, st);
    ensure_csr32(B, st);
    ensure_rowptr(C, st);
    RsArgs g{};
    g.a_rowptr = A->csr_rowptr; g.a_cols = A->csr_cols; g.a_vals = A->csr_vals; g.a_rows = (uint32_t)A->num_rows;
    g.b_rowptr = B->csr_rowptr; g.b_cols = B->csr_cols; g.b_vals = B->csr_vals; g.b_rows = (uint32_t)B->num_rows;
    g.c_keys = C->keys; g.c_bmps = C->bmps; g.c_offs = C->offsets; g.c_rowptr = C->rowptr; g.c_vals = (float *)C->values;
    g.block_rows = (uint32_t)A->num_block_rows();
    hipLaunchKernelGGL(block_mac_rowsparse_kernel, dim3(g.block_rows), dim3(64), 0, st, g);
    BMSP_CHECK_LAUNCH();
}

}  // namespace bmsp

BMSP_DEFINE_WARM(blockmac_rowsparse)
