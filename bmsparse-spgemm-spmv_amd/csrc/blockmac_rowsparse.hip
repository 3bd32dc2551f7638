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
// One workgroup (L / 8 waves) per block-row of C; C's structure is given (T_3 ... T_9 have fixed it).  Accumulators: one float per stored
// value of the block-row, in LDS (a window of C tiles of at most kRsAcc values at a time; a block-row that holds more is walked once per
// window); C's block columns of the block-row in an LDS hash table.  The workgroup's eight groups of L lanes take the block-row's eight
// rows at once (rows of different index never meet in a C element); a group walks its row of A entry by entry (L entries and the bounds
// of their rows of B fetched together, handed round by shuffles), its L lanes = L entries of B's row k: column -> C tile (hash) -> slot
// (rank in C's bitmap) -> one LDS read-modify-write.  Inside an iteration the products of a group go to different columns of one row:
// no conflicts, no atomics.  Operands: row-major CSR copies (row pointer, {column, value} entries) derived once per matrix, like the
// dense copies the matrix-core kernels read.
// (Two forms that walked the tiles themselves -- one A tile per step, lane = a tile of B's block-row k, the A tile's values in a scalar
// loop -- measured 1171 and 1084 us on the FEM-like product, the second with B's values parked in LDS: 96 products per step spread over
// eleven serialized read-modify-write iterations at one lane in seven.)
#include "mac_common.hip.h"
#include <algorithm>

namespace bmsp {
namespace {

// the table of a block-row's C columns: 2^HB slots.  HB = 9 for at most 256 tiles (strip mode's bound; 17.1 KB of LDS: nine workgroups per
// CU -- a 385-entry offset array made it eight, FEM-like T_7 285 -> 324 us) or 10 for at most 768 (products that come with a task list,
// cage-like block-rows of 400 - 600 tiles; 27 KB, five workgroups per CU)
constexpr uint32_t rs_row_cap(int hb) { return hb == 9 ? 256u : 768u; }
constexpr uint32_t kRsAcc = 2048;     // accumulators per window
constexpr uint32_t kRsEmpty = 0xffffffffu;

struct RsArgs {
    const uint32_t *a_rowptr;  // A row-major: CSR row pointer (num_rows + 1)
    const uint32_t *a_ent;     // ... and its entries {column, value bits}
    const uint32_t *b_rowptr, *b_ent;
    uint32_t a_rows, b_rows, a_ent_bytes, b_ent_bytes;
    const uint64_t *c_keys, *c_bmps, *c_offs;
    const uint32_t *c_rowptr;  // per block-row
    float *c_vals;
    uint32_t block_rows;
};

typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
// an entry {column, value bits} as a FLOAT vector: __builtin_bit_cast(float, v[1]) of an integer vector's element reads element 0 with this
// compiler (ROCm 7.2) -- element 1 is read as the float it is, element 0 is cast to the column
__device__ __forceinline__ f32x2s rs_entry(rsrc_t r, uint32_t byte_off) { return __builtin_bit_cast(f32x2s, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0)); }
template <int HB>
struct alignas(16) RsLds {
    u32x4s slot[1 << HB];        // {block column of a C tile of the block-row (kRsEmpty = free), its first value relative to the block-row's
                                 // first, its bitmap (lo, hi)}: ONE 16-byte read per probe gives all a product needs
    uint32_t co[rs_row_cap(HB) + 1];  // first value of every C tile relative to the block-row's first, in tile order (the windows' bounds)
    float acc[kRsAcc];
};  // HB = 9: 17.6 KB, nine one-wave workgroups per CU

template <int HB>
__device__ __forceinline__ uint32_t rs_hash(uint32_t j) { return (j * 0x9E3779B1u) >> (32 - HB); }

// L = lanes per row of C, a group's L lanes = L entries of B's row k.  A workgroup of L / 8 waves owns the block-row: its eight lane groups
// take the eight rows at once (rows of different index never meet in a C element), table and accumulators are shared, built and stored
// by all waves (one wave per block-row walked the rows 64 / L at a time: a cage-like block-row is 32 short iterations behind 30 us of
// set-up and dependent loads, at five waves per CU).  (The kernel is bound by its instruction count -- 1.4 M iterations of
// ~110 vector instructions on the FEM-like product, where a 16-byte table slot instead of four dependent LDS reads and requests three
// entries ahead instead of one changed nothing (407 -> 410 us) -- so L follows B's average row length, the bitmap arithmetic is
// 32-bit (the row is uniform per group), and an entry is one 8-byte buffer load.)
// HALF: fp16 operands with V15's numerics (tc_version 5, the reference's default configuration): the product is rounded to fp16 before
// the fp32 add (:269-273, `__half * __half`); the entries hold the fp16 values widened to fp32 (exact), their fp32 product is exact (11 x
// 11 bits), so one conversion to fp16 is that rounding.  No exponent condition there: an fp32 sum of fp16 values never underflows, and
// +0 + -0 = +0, so a sum is never -0 and a left-out zero term never shows.
template <int L, bool HALF, int HB>
__global__ __launch_bounds__(8 * L) void block_mac_rowsparse_kernel(RsArgs g)
{
    constexpr uint32_t kRsHash = 1u << HB, kRsRowCap = rs_row_cap(HB);
    constexpr uint32_t NT = 8u * L;  // threads: L / 8 waves
    __shared__ RsLds<HB> S;
    const int lane = lane_id();
    const uint32_t tid = threadIdx.x;
    // XCD-aware order: the workgroups of one XCD take a contiguous eighth of the block-rows (neighbours read the same rows of B)
    uint32_t brow;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        brow = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    if (brow >= g.block_rows) return;
    const uint32_t c0 = g.c_rowptr[brow], m = g.c_rowptr[brow + 1] - c0;
    if (m == 0 || m > kRsRowCap) return;  // (the launcher admits no product with a longer block-row of C)
    // ---- C's block-row: column table (column, value offset, bitmap per slot), value offsets in tile order ----
    for (uint32_t s = tid; s < (uint32_t)kRsHash; s += NT) S.slot[s] = u32x4s{kRsEmpty, 0u, 0u, 0u};
    __syncthreads();
    const uint64_t vbase = g.c_offs[c0];
    for (uint32_t r = tid; r < m; r += NT) {
        const uint32_t j = key_col(g.c_keys[c0 + r]);
        const uint64_t bm = g.c_bmps[c0 + r];
        const uint32_t off = (uint32_t)(g.c_offs[c0 + r] - vbase);
        uint32_t slot = rs_hash<HB>(j);
        uint32_t *const words = (uint32_t *)S.slot;
        while (atomicCAS(&words[4u * slot], kRsEmpty, j) != kRsEmpty) slot = (slot + 1u) & (uint32_t)(kRsHash - 1);  // (the columns of a block-row are distinct)
        words[4u * slot + 1u] = off; words[4u * slot + 2u] = (uint32_t)bm; words[4u * slot + 3u] = (uint32_t)(bm >> 32);
        S.co[r] = off;
    }
    if (tid == 0) S.co[m] = (uint32_t)(g.c_offs[c0 + m] - vbase);
    __syncthreads();

    constexpr int GROUPS = 64 / L;  // lane groups per wave; GROUPS * (L / 8) = 8 rows at once
    const int grp = lane / L, ll = lane % L, gbase = lane - ll;
    const rsrc_t ra = make_rsrc(g.a_ent, g.a_ent_bytes), rb = make_rsrc(g.b_ent, g.b_ent_bytes);
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
        for (uint32_t e = tid; e < nv; e += NT) S.acc[e] = 0.f;
        __syncthreads();
        {
            const uint32_t i = (uint32_t)(GROUPS * wave_id() + grp);  // row inside the tiles: bits 8 i .. 8 i + 7 of a C bitmap, MSB first
            const uint32_t row = brow * 8u + i;
            const bool top = i < 4u;                   // the row's byte lies in the bitmap's high word
            const uint32_t sh0 = 24u - 8u * (i & 3u);  // ... at this shift
            uint32_t pa0 = 0, pa1 = 0;
            if (row < g.a_rows) { pa0 = g.a_rowptr[row]; pa1 = g.a_rowptr[row + 1]; }
            // L entries of A's row per group at a time: column k, value, and the bounds of B's row k
            for (uint32_t pbase = pa0; __any(pbase < pa1); pbase += (uint32_t)L) {
                const uint32_t p = pbase + (uint32_t)ll;
                uint32_t b0_l = 0, b1_l = 0;
                float a_l = 0.f;
                if (p < pa1) {
                    const f32x2s ea = rs_entry(ra, p << 3);
                    const uint32_t k = __builtin_bit_cast(uint32_t, ea[0]);
                    a_l = ea[1];
                    if (k < g.b_rows) { b0_l = g.b_rowptr[k]; b1_l = g.b_rowptr[k + 1]; }
                }
                uint32_t nt = pbase < pa1 ? min((uint32_t)L, pa1 - pbase) : 0u;  // entries of this group in the chunk
                uint32_t nt_max = nt;
#pragma unroll
                for (int d = L; d < 64; d <<= 1) nt_max = max(nt_max, (uint32_t)__shfl_xor((int)nt_max, d, kWave));
                // entry t of the chunk: its row of B, L entries per group and iteration, requested three entries ahead
                struct Ent {
                    uint32_t e0, e1;
                    f32x2s cb;  // {column bits, value} of this lane's entry of B's row
                    float a;
                };
                auto fetch = [&](uint32_t t) {
                    Ent q;
                    const int src = gbase + (int)min(t, (uint32_t)(L - 1));
                    q.e0 = (uint32_t)__shfl((int)b0_l, src, kWave);
                    q.e1 = (uint32_t)__shfl((int)b1_l, src, kWave);
                    q.a = __shfl(a_l, src, kWave);
                    if (t >= nt) q.e1 = q.e0;
                    const uint32_t e = q.e0 + (uint32_t)ll;
                    q.cb = rs_entry(rb, e < q.e1 ? e << 3 : kOob);
                    return q;
                };
                auto product = [&](uint32_t col, float b, float a) {
                    const uint32_t jb = col >> 3, j = col & 7u;
                    uint32_t slot = rs_hash<HB>(jb);
                    for (;;) {
                        const u32x4s sl = S.slot[slot];
                        if (sl[0] == jb) {
                            if (sl[1] - o0 < nv) {  // the tile lies in the window (its values start inside the window's run)
                                const uint32_t word = top ? sl[3] : sl[2];
                                const uint32_t shb = sh0 + 7u - j;  // bit of (i, j) inside the word
                                if ((word >> shb) & 1u) {
                                    const uint32_t ci = sl[1] - o0 + (top ? 0u : (uint32_t)__builtin_popcount(sl[3])) + (uint32_t)__builtin_popcount((word >> 1) >> shb);
                                    if constexpr (HALF) S.acc[ci] = S.acc[ci] + (float)(_Float16)(a * b);
                                    else S.acc[ci] = __builtin_fmaf(a, b, S.acc[ci]);
                                }
                            }
                            break;
                        }
                        if (sl[0] == kRsEmpty) break;  // (cannot happen for a product of stored values: C's structure holds its tile)
                        slot = (slot + 1u) & (uint32_t)(kRsHash - 1);
                    }
                };
                // four entries in flight, each in registers of its own (a rotating set cost fifteen moves per iteration)
                auto step = [&](Ent &q, uint32_t t) {
                    if (t >= nt_max) return;
                    const Ent c = q;
                    if (t + 4 < nt_max) q = fetch(t + 4);
                    if (c.e0 + (uint32_t)ll < c.e1) product(__builtin_bit_cast(uint32_t, c.cb[0]), c.cb[1], c.a);
                    if (__any(c.e0 + (uint32_t)L < c.e1)) {  // a row of B beyond L entries: the later ones are fetched here
                        for (uint32_t eb = c.e0 + (uint32_t)L; __any(eb < c.e1); eb += (uint32_t)L) {
                            const uint32_t e = eb + (uint32_t)ll;
                            if (e < c.e1) {
                                const f32x2s cb = rs_entry(rb, e << 3);
                                product(__builtin_bit_cast(uint32_t, cb[0]), cb[1], c.a);
                            }
                        }
                    }
                };
                Ent q0 = fetch(0u), q1 = q0, q2 = q0, q3 = q0;
                if (1u < nt_max) q1 = fetch(1u);
                if (2u < nt_max) q2 = fetch(2u);
                if (3u < nt_max) q3 = fetch(3u);
                for (uint32_t t = 0; t < nt_max; t += 4) {
                    step(q0, t); step(q1, t + 1); step(q2, t + 2); step(q3, t + 3);
                }
            }
        }
        __syncthreads();
        // ---- the window's values leave in one run ----
        float *const dst = g.c_vals + vbase + o0;
        for (uint32_t e = tid; e < nv; e += NT) dst[e] = S.acc[e];
        __syncthreads();
        w0 = w1;
    }
}

// ---- the CSR copy, straight from the tiles (no sort: a block-row's tiles are in column order already) --------------------------------
// the stored values of the matrix rows 8 * block-row + r, r = 0 .. 7: one wave per block-row adds up the bytes' popcounts of its tiles'
// row-major bitmaps (an atomic per tile and row landed on the same eight counters: 108 us per operand on the FEM-like matrix)
__global__ __launch_bounds__(kThreads) void rs_count_kernel(const uint64_t *__restrict__ bmps, const uint32_t *__restrict__ block_rowptr, uint32_t block_rows,
                                                            int transposed, uint32_t num_rows, uint32_t *__restrict__ cnt)
{
    const uint32_t br = blockIdx.x * 4 + (uint32_t)wave_id();
    if (br >= block_rows) return;
    const int lane = lane_id();
    const uint32_t t0 = block_rowptr[br], t1 = block_rowptr[br + 1];
    uint64_t lo = 0, hi = 0;  // rows 0-3 / 4-7: four 16-bit fields each
    for (uint32_t t = t0 + (uint32_t)lane; t < t1; t += 64) {
        const uint64_t bm = transposed ? tile_transpose(bmps[t]) : bmps[t];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            lo += (uint64_t)__builtin_popcount(tile_byte(bm, r)) << (16 * r);
            hi += (uint64_t)__builtin_popcount(tile_byte(bm, 4 + r)) << (16 * r);
        }
    }
    // (a lane sees at most 2^13 / 64 tiles of 8 values per row: the fields cannot carry into each other before the sum over lanes either)
    lo = wave_sum(lo); hi = wave_sum(hi);
    if (lane < 8) {
        const uint32_t row = br * 8u + (uint32_t)lane;
        if (row < num_rows) cnt[row] = (uint32_t)(((lane < 4 ? lo : hi) >> (16 * (lane & 3))) & 0xffffull);
    }
    if (br == 0 && lane == 8) cnt[num_rows] = 0u;
}

// one wave per block-row: 64 tiles at a time (lane = tile), a packed wave scan of the tiles' eight row counts gives every tile the place
// of its values inside each of the eight rows; the lane then writes its tile's values, row by row, column ascending
template <typename T>
__global__ __launch_bounds__(kThreads) void rs_fill_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bmps, const uint64_t *__restrict__ offs,
                                                           const T *__restrict__ vals, const uint32_t *__restrict__ block_rowptr, uint32_t block_rows,
                                                           int transposed, uint32_t num_rows, const uint32_t *__restrict__ rowptr, uint32_t *__restrict__ ent)
{
    const uint32_t br = blockIdx.x * 4 + (uint32_t)wave_id();
    if (br >= block_rows) return;
    const int lane = lane_id();
    const uint32_t t0 = block_rowptr[br], t1 = block_rowptr[br + 1];
    // rows 0-3 in `lo`, rows 4-7 in `hi`: four 16-bit fields each (a block-row of 2^13 tiles of 8 values per row stays below 2^16)
    uint64_t carry_lo = 0, carry_hi = 0;
    for (uint32_t base = t0; base < t1; base += 64) {
        const uint32_t t = base + (uint32_t)lane;
        uint64_t bm = 0, stored = 0;
        if (t < t1) { stored = bmps[t]; bm = transposed ? tile_transpose(stored) : stored; }
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            lo |= (uint64_t)__builtin_popcount(tile_byte(bm, r)) << (16 * r);
            hi |= (uint64_t)__builtin_popcount(tile_byte(bm, 4 + r)) << (16 * r);
        }
        const uint64_t inc_lo = wave_inclusive_sum(lo), inc_hi = wave_inclusive_sum(hi);
        if (t < t1 && bm) {
            const uint64_t ex_lo = carry_lo + inc_lo - lo, ex_hi = carry_hi + inc_hi - hi;
            const uint32_t bcol8 = key_col(keys[t]) * 8u;
            const uint64_t off = offs[t];
#pragma unroll 1
            for (int r = 0; r < 8; r++) {
                const uint32_t row = br * 8u + (uint32_t)r;
                uint32_t byte = tile_byte(bm, r);
                if (!byte || row >= num_rows) continue;
                uint32_t dst = rowptr[row] + (uint32_t)(((r < 4 ? ex_lo : ex_hi) >> (16 * (r & 3))) & 0xffffull);
                while (byte) {
                    const int c = __builtin_clz(byte) - 24;  // column inside the tile, ascending
                    byte &= ~(0x80u >> c);
                    const int pos = transposed ? c * 8 + r : r * 8 + c;  // where the value sits in the stored order
                    ent[2 * (uint64_t)dst] = bcol8 + (uint32_t)c;
                    ent[2 * (uint64_t)dst + 1] = __builtin_bit_cast(uint32_t, (float)vals[off + (uint64_t)tile_rank(stored, pos)]);  // (fp16 -> fp32: exact)
                    dst++;
                }
            }
        }
        carry_lo += (uint64_t)__shfl((long long)inc_lo, 63, kWave);
        carry_hi += (uint64_t)__shfl((long long)inc_hi, 63, kWave);
    }
}

}  // namespace

// row-major CSR copy of an fp32 or fp16 matrix (row pointer, {column, value as fp32} entries), whatever its tile layout: built once per
// matrix, like the dense tile copies; dropped by bmsp_matrix_invalidate
void ensure_csr32(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->csr_rowptr) return;
    const uint64_t n = (uint64_t)m->nnz;
    const uint32_t rows = (uint32_t)m->num_rows;
    ensure_rowptr(m, st);
    ensure_row_stats(m, st);
    if (m->max_row_blocks >= (1 << 13)) fail(BMSP_ERR_LIMIT, "row-sparse block-MAC: a block-row of 8192 or more tiles");
    m->csr_rowptr = (uint32_t *)pool_alloc(4 * ((size_t)rows + 1));
    m->csr_ent = (uint32_t *)pool_alloc(8 * (size_t)(n ? n : 1));
    const uint32_t nbr_c = (uint32_t)m->num_block_rows();
    if (nbr_c) {
        hipLaunchKernelGGL(rs_count_kernel, dim3((nbr_c + 3) / 4), dim3(kThreads), 0, st, (const uint64_t *)m->bmps, (const uint32_t *)m->rowptr, nbr_c, m->transposed, rows,
                           m->csr_rowptr);
        BMSP_CHECK_LAUNCH();
    } else {
        BMSP_HIP(hipMemsetAsync(m->csr_rowptr, 0, 4 * ((size_t)rows + 1), st));
    }
    device_exclusive_scan<uint32_t>(PtrIn<uint32_t>{m->csr_rowptr}, PtrOut<uint32_t>{m->csr_rowptr}, (uint64_t)rows + 1, st);
    if (m->block_num) {
        const uint32_t nbr = (uint32_t)m->num_block_rows();
        const dim3 grid((nbr + 3) / 4);
        if (m->dtype == BMSP_F32)
            hipLaunchKernelGGL((rs_fill_kernel<float>), grid, dim3(kThreads), 0, st, m->keys, m->bmps, m->offsets, (const float *)m->values, m->rowptr, nbr, m->transposed, rows,
                               (const uint32_t *)m->csr_rowptr, m->csr_ent);
        else
            hipLaunchKernelGGL((rs_fill_kernel<_Float16>), grid, dim3(kThreads), 0, st, m->keys, m->bmps, m->offsets, (const _Float16 *)m->values, m->rowptr, nbr, m->transposed,
                               rows, (const uint32_t *)m->csr_rowptr, m->csr_ent);
        BMSP_CHECK_LAUNCH();
    }
}

// Operands of nearly empty tiles (at most 16 stored values per tile on average on both sides) whose product takes V15's numerics -- fp32
// operands under any tc_version, fp16 operands under tc_version 5: the scalar products that exist are a few per cent of what a
// tile-by-tile kernel multiplies.  Needs finite values and, for fp32, every product of stored values a normal number (biased exponents
// summing to >= 128); the caller bounds the block-rows of C by mac_strip_row_cap() tiles.  BMSP_MAC_ROWSPARSE=0/1: never / whatever
// the fill.
bool mac_rowsparse_applies(bmsp_matrix_s *A, bmsp_matrix_s *B, int tc_version, hipStream_t st)
{
    const char *e = getenv("BMSP_MAC_ROWSPARSE");
    if (e && e[0] == '0') return false;
    if (A->dtype != B->dtype || (A->dtype != BMSP_F32 && !(A->dtype == BMSP_F16 && tc_version == 5))) return false;
    if ((uint64_t)A->nnz >= (1ull << 29) || (uint64_t)B->nnz >= (1ull << 29)) return false;  // (8-byte entries behind 32-bit byte offsets)
    if (A->view_values_end || B->view_values_end || A->ownership == 2 || B->ownership == 2) return false;  // (row-panel views: no copy of their own)
    if (!(e && e[0] == '1') && !(A->nnz <= 16 * A->block_num && B->nnz <= 16 * B->block_num)) return false;
    ensure_finite_flag(A, st);
    ensure_finite_flag(B, st);
    if (A->values_finite != 1 || B->values_finite != 1) return false;
    return A->dtype != BMSP_F32 || A->f32_exp_min + B->f32_exp_min >= 128;
}

// every block-row of C within the larger table's capacity
bool mac_rowsparse_fits_c(bmsp_matrix_s *C, hipStream_t st)
{
    if (C->block_num >= (1ll << 31) || C->block_num == 0) return false;
    ensure_row_stats(C, st);
    return (uint64_t)C->max_row_blocks <= rs_row_cap(10);
}

void launch_mac_rowsparse(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st)
{
    ensure_csr32(A, st);
    ensure_csr32(B, st);
    ensure_rowptr(C, st);
    RsArgs g{};
    g.a_rowptr = A->csr_rowptr; g.a_ent = A->csr_ent; g.a_rows = (uint32_t)A->num_rows; g.a_ent_bytes = (uint32_t)((uint64_t)A->nnz * 8u);
    g.b_rowptr = B->csr_rowptr; g.b_ent = B->csr_ent; g.b_rows = (uint32_t)B->num_rows; g.b_ent_bytes = (uint32_t)((uint64_t)B->nnz * 8u);
    g.c_keys = C->keys; g.c_bmps = C->bmps; g.c_offs = C->offsets; g.c_rowptr = C->rowptr; g.c_vals = (float *)C->values;
    g.block_rows = (uint32_t)A->num_block_rows();
    // lanes per row of C = 8 x the waves per block-row.  Measured (T_7, us; FEM-like, rows of B of 26 / cage-like, 15.6): 8 lanes, one
    // wave 292 / 625; 16 lanes, two waves 174 / 380; 32 lanes, four waves 134 / 277 -- the waves per block-row matter more than the lanes a
    // short row of B leaves idle; only rows of B of a few entries take fewer.  BMSP_RS_LANES = 8 / 16 / 32 for A/B runs
    const uint64_t avg = B->num_rows ? (uint64_t)B->nnz / (uint64_t)B->num_rows : 0;
    int lanes = avg <= 3 ? 8 : (avg <= 6 ? 16 : 32);
    if (const char *e = getenv("BMSP_RS_LANES")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32) lanes = v; }
    const dim3 grid(g.block_rows), block(8u * (unsigned)lanes);
    const bool half = A->dtype == BMSP_F16;
    ensure_row_stats(C, st);
    const bool big = (uint64_t)C->max_row_blocks > rs_row_cap(9);  // (mac_rowsparse_fits_c has bounded it by rs_row_cap(10))
#define BMSP_RS_LAUNCH(L_, H_, B_) hipLaunchKernelGGL((block_mac_rowsparse_kernel<L_, H_, B_>), grid, block, 0, st, g)
#define BMSP_RS_PICK(L_) \
    do { \
        if (half) { if (big) BMSP_RS_LAUNCH(L_, true, 10); else BMSP_RS_LAUNCH(L_, true, 9); } \
        else { if (big) BMSP_RS_LAUNCH(L_, false, 10); else BMSP_RS_LAUNCH(L_, false, 9); } \
    } while (0)
    if (lanes == 8) BMSP_RS_PICK(8);
    else if (lanes == 16) BMSP_RS_PICK(16);
    else BMSP_RS_PICK(32);
#undef BMSP_RS_PICK
#undef BMSP_RS_LAUNCH
    BMSP_CHECK_LAUNCH();
}

}  // namespace bmsp

BMSP_DEFINE_WARM(blockmac_rowsparse)
