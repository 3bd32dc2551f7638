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
// Two forms: the strip-mode pass below (structure only: block-rows of C of at most 256 tiles, numeric stage by the strip kernels) and the
// task-list form further down (build + copy passes: block-rows of up to ~900 tiles, the sorted task list for the task-list kernels).
// Beyond those limits the caller keeps the expand-sort-compress pipeline.
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
    const uint32_t *b_recs;
    const uint32_t *b_rowptr;
    uint32_t block_rows, b_block_rows;
    uint32_t row_cap;         // distinct C tiles per block-row the caller accepts (<= kRowCap)
    const uint32_t *tmp_off;  // block_rows + 1: first scratch slot of block-row i; null: row_cap slots per block-row
    uint32_t *cand;           // when tmp_off is null (T_2 did not run): candidate pairs of block-row i
    uint32_t *t_cols;         // scratch: C's block columns of row i, ascending
    uint64_t *t_bmps;         //          and their bitmaps
    uint32_t *cnt;            // block_rows (+ 1, the scan reads one past): C tiles of row i
    uint32_t *surv;           // block_rows: candidate pairs of row i that passed the filter
    uint32_t *nnz;            // block_rows: values of C's block-row i (bits of its bitmaps)
    uint32_t *overflow;
};

// chunk of 64 A tiles of the block-row being walked
struct ChunkLds {
    uint32_t abeg[64];   // first tile of B's block-row k
    uint32_t aend[64];   // ... and one past its last
    uint64_t abmp[64];
};

struct alignas(16) RowLds {
    uint32_t hk[kHash];  // block column; kEmpty = free
    uint64_t hb[kHash];  // OR of the tile-product bitmaps
    ChunkLds ch;
};  // 7 KB per wave: five workgroups per CU

struct WalkArgs {
    const uint64_t *a_keys, *a_bmps;
    const uint32_t *b_recs;  // per B tile {bitmap ROW-major (lo, hi), block column, rows the tile uses} (matrix.h: sym_recs): one 16-byte load per candidate pair
    const uint32_t *b_rowptr;
    uint32_t b_block_rows;
};
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

// Walks the candidate pairs of the A tiles [a0, a1) (one block-row of A).  64 / LPT A tiles at a time, LPT lanes each: the lanes of a
// group walk B's block-row k of their tile LPT tiles per step (the words of the next step are requested before the current ones are
// used).  step(live, a, t, j, abm, rows, bbm) is called by the whole wave once per step -- live: this lane holds a candidate (A tile a, B
// tile t of block column j, their bitmaps) -- and returns false (wave-uniformly) to stop the walk.  LPT = 64: one A tile at a time, i.e.
// the candidates arrive in ascending A tile.
template <int LPT = 16, int DEPTH = 4, typename Step>
__device__ __forceinline__ bool walk_row(const WalkArgs &g, ChunkLds &L, uint32_t a0, uint32_t a1, int lane, Step step)
{
    constexpr uint32_t GROUPS = 64 / LPT;
    struct Cur {
        uint32_t u, t, end, j, rows;  // rows: bit 7-k = row k of the B tile holds a value
        uint64_t abm, bbm;            // bbm: the B tile ROW-major
        bool valid;  // (wave-uniform) false: past the chunk's last step
    };
    const u32x4r *const recs = (const u32x4r *)g.b_recs;
    auto load = [&](uint32_t t, uint32_t &j, uint32_t &rows, uint64_t &bm) {
        const u32x4r r = recs[t];
        j = r[2]; rows = r[3]; bm = ((uint64_t)r[1] << 32) | (uint64_t)r[0];
    };
    for (uint32_t base = a0; base < a1; base += 64) {
        const uint32_t a = base + (uint32_t)lane;
        const bool on = a < a1;
        const uint32_t k = on ? key_col(g.a_keys[a]) : 0u;
        uint32_t bb = 0, be = 0;
        if (on && k < g.b_block_rows) { bb = g.b_rowptr[k]; be = g.b_rowptr[k + 1]; }
        __builtin_amdgcn_wave_barrier();
        L.abeg[lane] = bb; L.aend[lane] = be; L.abmp[lane] = on ? g.a_bmps[a] : 0ull;
        __builtin_amdgcn_wave_barrier();
        const uint32_t na = min(64u, a1 - base);
        uint32_t g4 = 0;  // group of the LAST step requested
        // the first step of the group of A tiles that starts at g4 (its words requested here)
        auto enter = [&]() {
            Cur c;
            c.valid = true;
            c.u = g4 + (uint32_t)lane / (uint32_t)LPT;
            c.end = c.u < na ? L.aend[c.u] : 0u;
            c.abm = L.abmp[min(c.u, 63u)];
            c.t = (c.u < na ? L.abeg[c.u] : 0u) + (uint32_t)lane % (uint32_t)LPT;
            c.j = 0; c.bbm = 0; c.rows = 0;
            if (c.t < c.end) load(c.t, c.j, c.rows, c.bbm);
            return c;
        };
        // the step after c: the group's next LPT tiles, or the first step of the next group
        auto after = [&](const Cur &c) {
            Cur n = c;
            if (!c.valid) return n;
            if (__any(c.t + (uint32_t)LPT < c.end)) {
                n.t = c.t + (uint32_t)LPT;
                n.j = 0; n.bbm = 0; n.rows = 0;
                if (n.t < n.end) load(n.t, n.j, n.rows, n.bbm);
            } else {
                g4 += GROUPS;
                if (g4 < na) n = enter();
                else n.valid = false;
            }
            return n;
        };
        // DEPTH steps in flight: a step's words have DEPTH - 1 steps' time to arrive.  DEPTH 1: the next step of the SAME group only (the
        // first step of a group waits for its words) -- fewest instructions per step, and the fastest form where four A tiles share a
        // step (FEM-like strip-mode pass 298 us; two or four steps in flight across groups 352 / 363 us)
        static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "one, two or four steps in flight");
        if constexpr (DEPTH == 1) {
            for (g4 = 0; g4 < na; g4 += GROUPS) {
                Cur c = enter();
                while (__any(c.t < c.end)) {
                    const uint32_t tn = c.t + (uint32_t)LPT;
                    uint32_t jn = 0, rn = 0;
                    uint64_t bn = 0;
                    if (tn < c.end) load(tn, jn, rn, bn);
                    if (!step(c.t < c.end, base + c.u, c.t, c.j, c.abm, c.rows, c.bbm)) return false;
                    c.t = tn; c.j = jn; c.rows = rn; c.bbm = bn;
                }
            }
        } else if constexpr (DEPTH == 4) {
            Cur q0 = enter(), q1 = after(q0), q2 = after(q1), q3 = after(q2);
            while (q0.valid) {
                if (__any(q0.t < q0.end)) {
                    if (!step(q0.t < q0.end, base + q0.u, q0.t, q0.j, q0.abm, q0.rows, q0.bbm)) return false;
                }
                q0 = q1; q1 = q2; q2 = q3; q3 = after(q3);
            }
        } else {
            Cur q0 = enter(), q1 = after(q0);
            while (q0.valid) {
                if (__any(q0.t < q0.end)) {
                    if (!step(q0.t < q0.end, base + q0.u, q0.t, q0.j, q0.abm, q0.rows, q0.bbm)) return false;
                }
                q0 = q1; q1 = after(q1);
            }
        }
    }
    return true;
}

// insert block column j into the open-addressing table hk (kEmpty = free); returns the slot, fresh = this call claimed it
template <int BITS = kHashBits>
__device__ __forceinline__ uint32_t hash_insert(uint32_t *hk, uint32_t j, bool &fresh)
{
    uint32_t slot = (j * 0x9E3779B1u) >> (32 - BITS);
    for (;;) {
        const uint32_t old = atomicCAS(&hk[slot], kEmpty, j);
        if (old == kEmpty || old == j) { fresh = old == kEmpty; return slot; }
        slot = (slot + 1u) & ((1u << BITS) - 1u);
    }
}
// the same for a table that SEVERAL waves fill at once: the waves look at the column count once per step, so up to (waves x 64) inserts
// are under way when the cap is passed and the table can fill up -- an unbounded probe would then never end (found by a hang of
// test_spgemm_synthetic[0-5-wide]: 896 + 4 x 64 > 1024).  full = every slot was probed: the block-row has more columns than the table
// holds, the caller voids the pass
template <int BITS>
__device__ __forceinline__ uint32_t hash_insert_bounded(uint32_t *hk, uint32_t j, bool &fresh, bool &full)
{
    uint32_t slot = (j * 0x9E3779B1u) >> (32 - BITS);
    for (uint32_t probes = 0; probes < (1u << BITS); probes++) {
        const uint32_t old = atomicCAS(&hk[slot], kEmpty, j);
        if (old == kEmpty || old == j) { fresh = old == kEmpty; full = false; return slot; }
        slot = (slot + 1u) & ((1u << BITS) - 1u);
    }
    fresh = false; full = true;
    return 0u;
}
__device__ __forceinline__ uint32_t xcd_order(uint32_t b, uint32_t G)
{  // the workgroups of one XCD take a contiguous eighth of the block-rows (neighbouring rows read the same block-rows of B)
    const uint32_t q = G / 8, rm = G % 8, x = b % 8;
    return (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + b / 8;
}

// value offsets of a block-row's C tiles: the block-row's first value + the bits of the bitmaps in front (T_9's popcount scan, :1113-1130,
// split into a scan over the block-rows and this scan inside the block-row)
__device__ __forceinline__ void row_value_offsets(const uint64_t *bmps, uint32_t n, uint64_t base, uint64_t *offsets, int lane)
{
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < n; p0 += 64) {
        const uint32_t p = p0 + (uint32_t)lane;
        const uint32_t c = p < n ? (uint32_t)__popcll(bmps[p]) : 0u;
        const uint32_t inc = wave_inclusive_sum(c);
        if (p < n) offsets[p] = base + (uint64_t)(carry + inc - c);
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
}

template <int DEPTH>
__global__ __launch_bounds__(kThreads) void rowmerge_symbolic_kernel(RowMergeArgs g)
{
    __shared__ RowLds lds_all[4];
    const int w = wave_id(), lane = lane_id();
    RowLds &S = lds_all[w];
    const uint32_t row = xcd_order(blockIdx.x, gridDim.x) * 4 + (uint32_t)w;
    if (row >= g.block_rows) return;
    const uint32_t a0 = g.a_rowptr[row], a1 = g.a_rowptr[row + 1];
    // (a block-row beyond the cap makes the whole pass void: the waves that start after it was seen leave at once)
    if (a0 == a1 || __builtin_nontemporal_load(g.overflow) != 0u) {
        if (lane == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; g.nnz[row] = 0u; if (g.cand) g.cand[row] = 0u; }
        return;
    }
    for (uint32_t s = (uint32_t)lane; s < (uint32_t)kHash; s += 64) { S.hk[s] = kEmpty; S.hb[s] = 0ull; }
    uint32_t n = 0;       // distinct columns so far (wave-uniform)
    uint32_t surv = 0;    // this lane's surviving pairs
    const WalkArgs wa{g.a_keys, g.a_bmps, g.b_recs, g.b_rowptr, g.b_block_rows};
    const bool done = walk_row<16, DEPTH>(wa, S.ch, a0, a1, lane, [&](bool live, uint32_t, uint32_t, uint32_t j, uint64_t abm, uint32_t rows, uint64_t bbm) {
        if (n > g.row_cap) return false;
        const bool keep = live && (tile_or_bytes(abm) & rows);  // multiplication_checker (:742-757): a column of the A tile meets a row of the B tile
        surv += keep ? 1u : 0u;
        bool fresh = false;
        if (keep) {
            // bmp_calculator (:787-810) in its byte-permute form on the row-major B tile (88 instead of ~240 vector instructions: this pass was
            // 85 % VALU-bound on the FEM-like product); two full tiles: a full one.  (One A tile per step with the A side in scalar registers,
            // as in the build pass, measured SLOWER here: FEM-like T_3 202 -> 218 us at 55 of 64 lanes, the dense band 316 -> 603 us.)
            const uint64_t prod = (abm & bbm) == ~0ull ? ~0ull : tile_product_rm(abm, bbm);
            const uint32_t slot = hash_insert(S.hk, j, fresh);
            atomicOr((unsigned long long *)&S.hb[slot], (unsigned long long)prod);
        }
        n += (uint32_t)__popcll(__ballot(fresh));
        return true;
    });
    if (!done || n > g.row_cap) {
        if (lane == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; g.nnz[row] = 0u; if (g.cand) g.cand[row] = 0u; atomicOr(g.overflow, 1u); }
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
    const uint32_t out0 = g.tmp_off ? g.tmp_off[row] : row * g.row_cap;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    uint32_t nz = 0;
    for (uint32_t p0 = 0; p0 < m; p0 += 64) {
        const uint32_t p = p0 + (uint32_t)lane;
        const uint32_t key = p < m ? S.hk[p] : kEmpty;
        uint32_t rank = 0;
        for (uint32_t q = 0; q < m; q += 4) {
            const u32x4v v = *(const u32x4v *)&S.hk[q];
            rank += (v[0] < key ? 1u : 0u) + (v[1] < key ? 1u : 0u) + (v[2] < key ? 1u : 0u) + (v[3] < key ? 1u : 0u);
        }
        if (p < m) {
            const uint64_t bm = S.hb[p];
            g.t_cols[out0 + rank] = key;
            g.t_bmps[out0 + rank] = bm;
            nz += (uint32_t)__popcll(bm);
        }
    }
    surv = wave_sum(surv);
    nz = wave_sum(nz);
    if (g.cand) {  // the block-row's candidate pairs ("Task list size" without T_2's scan)
        uint32_t cd = 0;
        for (uint32_t a = a0 + (uint32_t)lane; a < a1; a += 64) {
            const uint32_t k = key_col(g.a_keys[a]);
            if (k < g.b_block_rows) cd += g.b_rowptr[k + 1] - g.b_rowptr[k];
        }
        cd = wave_sum(cd);
        if (lane == 0) g.cand[row] = cd;
    }
    if (lane == 0) { g.cnt[row] = m; g.surv[row] = surv; g.nnz[row] = nz; }  // (per-row results: one atomic pair per wave would serialise at the memory side)
}

// ---- row-merge WITH a task list: block-rows of C of up to kTlCap tiles, for the task-list block-MAC kernels ------------------------------
// Build pass, one wave per block-row of A, ONE walk over its candidate pairs: every surviving pair enters the column table (its C
// tile's task count with it) and is parked, with its tile-product bitmap, in walk order in the block-row's own stretch of scratch (the
// stretch of its candidate pairs, known from T_2's scan: no offsets to wait for).  Then, still row-local: the columns are ranked (C's
// key order), the task counts scanned in rank order, every parked pair put at (first task of its C tile + the count it found there), the
// products ORed into the tiles' bitmaps.  After two scans over the block-rows (C's block-row
// pointer, first task of every block-row) a copy pass moves keys, bitmaps, task ranges and tasks from the stretches to their places.
// (A first form -- count pass, then a fill pass that walked the candidates twice more -- measured 97 + 540 us on the cage-like product,
// against 660 us for the pipeline's five stages: three walks of 21 M candidate pairs cost what the pipeline costs.)
constexpr int kTlBits = 10;
constexpr int kTlHash = 1 << kTlBits;
constexpr uint32_t kTlCap = 896;  // distinct C tiles per block-row (+ 63)

struct TaskListArgs {
    WalkArgs w;
    const uint32_t *a_rowptr;
    const uint64_t *first_pos;  // T_2: candidate pairs in front of every A tile
    uint32_t block_rows;
    uint32_t *cnt, *surv, *nnz, *overflow;  // per block-row: C tiles, surviving pairs, values of C
    // scratch, indexed from the block-row's first candidate pair
    uint64_t *s_surv;   // surviving pairs in walk order: slot << 48 | (A tile - the row's first) << 32 | B tile
    uint64_t *s_prod;   // ... their tile products
    uint16_t *s_ord;    // ... and how many earlier pairs went to the same C tile
    uint64_t *s_tasks;  // the tasks in final order
    uint32_t *s_cols;   // by rank: C's block columns,
    uint32_t *s_begin;  //          first task of the C tile relative to the block-row's first,
    unsigned long long *s_bmps;  //  C's bitmaps
    // copy pass
    const uint32_t *c_rowptr, *row_task0;
    const uint64_t *row_val0;
    uint64_t *c_keys, *c_bmps, *c_offs;
    uint32_t *task_begin;
    uint64_t *tasks;
};

struct alignas(16) BuildLds {
    uint32_t hk[kTlHash];       // block column; kEmpty = free
    uint32_t tc[kTlHash / 2];   // two 16-bit fields per word (slot s: word s >> 1, half s & 1): tasks of the slot's C tile; once the column is ranked: its rank
    uint16_t list[kTlHash];     // the tiles' task counts in rank order, then their exclusive scan; the last 32 entries: prefix counts of the ranking's bit words
    ChunkLds ch;                // the walk's chunk of A tiles; after the walk: the ranking's bit words (8192 columns)
};  // 9 KB per wave: four workgroups per CU (13 KB -- separate rank array, 32-bit list, its own bit words -- held three)


template <int DEPTH>
__global__ __launch_bounds__(kThreads) void rowmerge_build_kernel(TaskListArgs g)
{
    __shared__ BuildLds lds_all[4];
    const int w = wave_id(), lane = lane_id();
    BuildLds &S = lds_all[w];
    const uint32_t row = xcd_order(blockIdx.x, gridDim.x) * 4 + (uint32_t)w;
    if (row >= g.block_rows) return;
    const uint32_t a0 = g.a_rowptr[row], a1 = g.a_rowptr[row + 1];
    if (a0 == a1 || __builtin_nontemporal_load(g.overflow) != 0u) {
        if (lane == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; g.nnz[row] = 0u; }
        return;
    }
    const uint64_t off = g.first_pos[a0];
    for (uint32_t s = (uint32_t)lane; s < (uint32_t)kTlHash; s += 64) S.hk[s] = kEmpty;
    for (uint32_t s = (uint32_t)lane; s < (uint32_t)kTlHash / 2; s += 64) S.tc[s] = 0u;
    __builtin_amdgcn_wave_barrier();
    // ---- the walk: table, task counts, surviving pairs and their products parked in walk order ----
    uint32_t n = 0, ns = 0;  // distinct columns, surviving pairs so far (wave-uniform)
    bool done = a1 - a0 <= 65535u;
    // (one A tile at a time: the columns of one block-row of B are distinct, so the count a pair finds at its C tile is the number of that
    // tile's tasks from smaller A tiles -- its place inside the tile, in V15's summation order (:269-273))
    if (done) done = walk_row<64, DEPTH>(g.w, S.ch, a0, a1, lane, [&](bool live, uint32_t a, uint32_t t, uint32_t j, uint64_t abm, uint32_t rows, uint64_t bbm) {
        if (n > kTlCap || ns + 64u > 65535u) return false;  // (task offsets inside a block-row are kept in 16 bits)
        // one A tile per step: its bitmap, the columns it uses and the product's byte masks are wave-uniform -- scalar registers, scalar ALU
        // (the pass spent 8.3 vector instructions per candidate pair, 70 % of the vector ALU, most of them in the 8 x 8 boolean product)
        const uint32_t ah = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(abm >> 32)), al = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)abm);
        const uint32_t cols = tile_or_bytes(((uint64_t)ah << 32) | (uint64_t)al);
        const bool keep = live && (cols & rows);  // multiplication_checker (:742-757)
        const uint64_t bal = __ballot(keep);
        bool fresh = false;
        if (keep) {
            const uint32_t slot = hash_insert<kTlBits>(S.hk, j, fresh);
            const uint32_t old = atomicAdd(&S.tc[slot >> 1], 1u << (16u * (slot & 1u)));
            const uint32_t k = ns + (uint32_t)__popcll(bal & lanemask_lt());
            g.s_surv[off + k] = ((uint64_t)slot << 48) | ((uint64_t)(a - a0) << 32) | (uint64_t)t;
            g.s_prod[off + k] = tile_product_scalar_a(ah, al, cols, (uint32_t)(bbm >> 32), (uint32_t)bbm);  // bmp_calculator (:787-810)
            g.s_ord[off + k] = (uint16_t)((old >> (16u * (slot & 1u))) & 0xffffu);
        }
        ns += (uint32_t)__popcll(bal);
        n += (uint32_t)__popcll(__ballot(fresh));
        return true;
    });
    if (!done || n > kTlCap) {
        if (lane == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; g.nnz[row] = 0u; atomicOr(g.overflow, 1u); }
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- rank of every column among the block-row's columns (= its C tile).  Counting sort by bitmap: the columns of a window of kWin
    //      block columns set their bit, the words' popcounts are scanned, rank = bits below.  (Counting the smaller columns per column,
    //      n^2 / 4 16-byte LDS reads per block-row, measured 390 us on the cage-like product; this form 55 us.) ----
    constexpr uint32_t kWin = 8192, kWinWords = kWin / 64;  // 1 KB of bit words in the chunk's place, one prefix count per four words
    static_assert(sizeof(ChunkLds) == kWinWords * 8, "the bit words take the walk's chunk");
    uint64_t *const bw = (uint64_t *)&S.ch;
    uint16_t *const bp = S.list + (kTlHash - kWinWords / 4);  // (ranks stay below kTlCap + 64 = 960 <= 992)
    uint32_t jmin = kEmpty, jmax = 0;
    for (uint32_t r = 0; r < (uint32_t)kTlHash; r += 64) {
        const uint32_t key = S.hk[r + (uint32_t)lane];
        if (key != kEmpty) { jmin = min(jmin, key); jmax = max(jmax, key); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        jmin = min(jmin, (uint32_t)__shfl_xor((int)jmin, d, kWave));
        jmax = max(jmax, (uint32_t)__shfl_xor((int)jmax, d, kWave));
    }
    uint32_t m = 0;  // columns ranked so far = columns below the current window
    for (uint32_t lo = jmin; n; lo += kWin) {
        for (uint32_t wd = (uint32_t)lane; wd < kWinWords; wd += 64) bw[wd] = 0ull;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t r = 0; r < (uint32_t)kTlHash; r += 64) {
            const uint32_t key = S.hk[r + (uint32_t)lane];
            if (key != kEmpty && key - lo < kWin) atomicOr((unsigned long long *)&bw[(key - lo) >> 6], 1ull << ((key - lo) & 63u));
        }
        __builtin_amdgcn_wave_barrier();
        {
            uint32_t c = 0;
            if (lane < (int)(kWinWords / 4))
                c = (uint32_t)(__popcll(bw[4 * lane]) + __popcll(bw[4 * lane + 1]) + __popcll(bw[4 * lane + 2]) + __popcll(bw[4 * lane + 3]));
            const uint32_t inc = wave_inclusive_sum(c);
            if (lane < (int)(kWinWords / 4)) bp[lane] = (uint16_t)(inc - c);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t r = 0; r < (uint32_t)kTlHash; r += 64) {
                const uint32_t s = r + (uint32_t)lane;
                const uint32_t key = S.hk[s];
                if (key != kEmpty && key - lo < kWin) {
                    const uint32_t wd = (key - lo) >> 6, bit = (key - lo) & 63u, w4 = wd & ~3u;
                    uint32_t rank = m + (uint32_t)bp[wd >> 2] + (uint32_t)__popcll(bw[wd] & ((1ull << bit) - 1ull));
                    if (wd > w4) rank += (uint32_t)__popcll(bw[w4]);
                    if (wd > w4 + 1) rank += (uint32_t)__popcll(bw[w4 + 1]);
                    if (wd > w4 + 2) rank += (uint32_t)__popcll(bw[w4 + 2]);
                    // the tile's task count goes to its rank's place; the slot keeps the rank instead (xor: the neighbour slot shares the word)
                    const uint32_t sh = 16u * (s & 1u), cnt_s = (S.tc[s >> 1] >> sh) & 0xffffu;
                    S.list[rank] = (uint16_t)cnt_s;
                    atomicXor(&S.tc[s >> 1], (cnt_s ^ rank) << sh);
                    g.s_cols[off + rank] = key;
                }
            }
            m += total;
        }
        __builtin_amdgcn_wave_barrier();
        if (jmax - lo < kWin) break;
    }
    // ---- exclusive scan of the task counts in rank order = first task of every C tile relative to the block-row's first ----
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < m; p0 += 64) {
        const uint32_t p = p0 + (uint32_t)lane;
        const uint32_t v = p < m ? (uint32_t)S.list[p] : 0u;
        const uint32_t inc = wave_inclusive_sum(v);
        if (p < m) {
            S.list[p] = (uint16_t)(carry + inc - v);
            g.s_begin[off + p] = carry + inc - v;
            g.s_bmps[off + p] = 0ull;
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    __builtin_amdgcn_wave_barrier();
    // the parked pairs and the zeroed bitmaps are written before other lanes of this wave read / OR into them: a workgroup-scope fence (the
    // wave's own CU: a wait for the stores; an agent-scope fence writes the XCD's L2 back -- 1.25 ms for this kernel on the cage-like product)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- the parked pairs to their places, their products into the tiles' bitmaps ----
    for (uint32_t k0 = 0; k0 < ns; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        if (k < ns) {
            const uint64_t e = g.s_surv[off + k], pr = g.s_prod[off + k];
            const uint32_t ord = (uint32_t)g.s_ord[off + k];
            const uint32_t slot = (uint32_t)(e >> 48);
            const uint32_t rank = (S.tc[slot >> 1] >> (16u * (slot & 1u))) & 0xffffu;
            g.s_tasks[off + (uint32_t)S.list[rank] + ord] = ((uint64_t)(a0 + (uint32_t)((e >> 32) & 0xffffu)) << 32) | (e & 0xffffffffull);
            atomicOr(&g.s_bmps[off + rank], (unsigned long long)pr);
        }
    }
    // values of the block-row (bits of its finished bitmaps: read past the L1, which may still hold the zeroed lines)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    uint32_t nz = 0;
    for (uint32_t p = (uint32_t)lane; p < m; p += 64)
        nz += (uint32_t)__popcll(__hip_atomic_load(g.s_bmps + off + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    nz = wave_sum(nz);
    if (lane == 0) { g.cnt[row] = m; g.surv[row] = ns; g.nnz[row] = nz; }
}

// ---- the build pass with a WORKGROUP per block-row (round 4) ---------------------------------------------------------------------------
// One wave per block-row is a serial chain -- a step's record load, table insert, parked stores, one A tile after the other: ~90 us for
// a cage-like block-row of 36 A tiles at 16 waves per CU, 372 us for the pass.  Here the four waves of a workgroup share the block-row:
// wave w walks the w-th quarter of its A tiles (a contiguous range, in order) and parks its surviving pairs in the stretch of ITS
// candidate pairs; the column table is shared (LDS atomics), the task counts are kept per wave and slot.  A task's place inside its C
// tile -- ascending A tile, V15's summation order (:269-273) -- is then (tasks of the tile from earlier waves) + (the count the pair found
// in its own wave's field on arrival): the waves' ranges ascend, inside a range the arrival order is the A-tile order, and the columns of
// one step are distinct.  Cage-like product: the pass 372 -> 340 us (T_3 520 -> 490).  (A count pass + fill pass built on the same
// ordering -- no parked pairs, the fill pass writing tasks and C tiles straight to their final places, no copy pass, no global atomic --
// was written and measured: 247 + 192 us against 340 + 92; clearing 20 KB of tables and the ranking's barriers per block-row at five
// workgroups per CU cost what the parked traffic costs.  Removed.)
constexpr int kBwWaves = 4;
struct alignas(16) BuildWgLds {
    uint32_t hk[kTlHash];                    // block column; kEmpty = free
    uint32_t tc[kBwWaves][kTlHash / 2];      // per wave: two 16-bit task counts per word; after the ranking: the tasks of the slot's tile from earlier waves
    uint16_t list[kTlHash];                  // the tiles' task counts in rank order, then their exclusive scan
    uint16_t rank_of[kTlHash];               // slot -> rank
    ChunkLds ch[kBwWaves];                   // the walks' chunks; ch[0] after the walks: the ranking's bit words
    uint16_t bp[32];
    uint32_t n_cols, n_surv, abort_flag, jmin, jmax, nz;
};  // 21 KB per workgroup: seven per CU, 28 waves

template <int DEPTH>
__global__ __launch_bounds__(64 * kBwWaves) void rowmerge_build_wg_kernel(TaskListArgs g)
{
    __shared__ BuildWgLds S;
    const int w = wave_id(), lane = lane_id();
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t NT = 64 * kBwWaves;
    const uint32_t row = xcd_order(blockIdx.x, gridDim.x);
    if (row >= g.block_rows) return;
    const uint32_t a0 = g.a_rowptr[row], a1 = g.a_rowptr[row + 1];
    if (a0 == a1 || __builtin_nontemporal_load(g.overflow) != 0u) {
        if (tid == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; g.nnz[row] = 0u; }
        return;
    }
    const uint64_t off = g.first_pos[a0];
    for (uint32_t s = tid; s < (uint32_t)kTlHash; s += NT) S.hk[s] = kEmpty;
    for (uint32_t s = tid; s < (uint32_t)(kBwWaves * kTlHash / 2); s += NT) (&S.tc[0][0])[s] = 0u;
    if (tid == 0) { S.n_cols = 0u; S.n_surv = 0u; S.abort_flag = a1 - a0 <= 65535u ? 0u : 1u; S.jmin = kEmpty; S.jmax = 0u; S.nz = 0u; }
    __syncthreads();
    // ---- the walks: wave w takes A tiles [aw0, aw1) ----
    const uint32_t na = a1 - a0, q = (na + kBwWaves - 1) / kBwWaves;
    const uint32_t aw0 = min(a0 + (uint32_t)w * q, a1), aw1 = min(aw0 + q, a1);
    const uint64_t offw = aw0 < a1 ? g.first_pos[aw0] : off;
    uint32_t ns = 0;  // this wave's surviving pairs so far (wave-uniform)
    if (aw0 < aw1 && S.abort_flag == 0u) {
        const bool done = walk_row<64, DEPTH>(g.w, S.ch[w], aw0, aw1, lane, [&](bool live, uint32_t a, uint32_t t, uint32_t j, uint64_t abm, uint32_t rows, uint64_t bbm) {
            if (ns + 64u > 65535u || *(volatile uint32_t *)&S.abort_flag) return false;
            const uint32_t ah = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(abm >> 32)), al = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)abm);
            const uint32_t cols = tile_or_bytes(((uint64_t)ah << 32) | (uint64_t)al);
            const bool keep = live && (cols & rows);  // multiplication_checker (:742-757)
            const uint64_t bal = __ballot(keep);
            bool fresh = false, full = false;
            if (keep) {
                const uint32_t slot = hash_insert_bounded<kTlBits>(S.hk, j, fresh, full);
                if (!full) {
                    const uint32_t old = atomicAdd(&S.tc[w][slot >> 1], 1u << (16u * (slot & 1u)));
                    const uint32_t k = ns + (uint32_t)__popcll(bal & lanemask_lt());
                    g.s_surv[offw + k] = ((uint64_t)slot << 48) | ((uint64_t)(a - a0) << 32) | (uint64_t)t;
                    g.s_prod[offw + k] = tile_product_scalar_a(ah, al, cols, (uint32_t)(bbm >> 32), (uint32_t)bbm);  // bmp_calculator (:787-810)
                    g.s_ord[offw + k] = (uint16_t)((old >> (16u * (slot & 1u))) & 0xffffu);
                }
            }
            if (__any(full)) {  // (the pass is void: the table is full)
                if (lane == 0) S.abort_flag = 1u;
                return false;
            }
            ns += (uint32_t)__popcll(bal);
            const uint32_t nf = (uint32_t)__popcll(__ballot(fresh));
            if (nf && lane == 0) {
                if (atomicAdd(&S.n_cols, nf) + nf > kTlCap) S.abort_flag = 1u;  // the table would overflow: every wave leaves
            }
            return true;
        });
        if (!done && lane == 0) S.abort_flag = 1u;
    }
    if (lane == 0) atomicAdd(&S.n_surv, ns);
    __syncthreads();
    const uint32_t n = S.n_cols, ns_all = S.n_surv;
    if (S.abort_flag != 0u || n > kTlCap || ns_all > 65535u) {  // (task offsets inside a block-row are kept in 16 bits)
        if (tid == 0) { g.cnt[row] = 0u; g.surv[row] = 0u; g.nnz[row] = 0u; atomicOr(g.overflow, 1u); }
        return;
    }
    // ---- rank of every column among the block-row's columns: counting sort by bitmap, a window of 8192 block columns at a time ----
    constexpr uint32_t kWin = 8192, kWinWords = kWin / 64;
    static_assert(sizeof(ChunkLds) == kWinWords * 8, "the bit words take a walk's chunk");
    uint64_t *const bw = (uint64_t *)&S.ch[0];
    {
        uint32_t mn = kEmpty, mx = 0;
        for (uint32_t s = tid; s < (uint32_t)kTlHash; s += NT) {
            const uint32_t key = S.hk[s];
            if (key != kEmpty) { mn = min(mn, key); mx = max(mx, key); }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, kWave));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, kWave));
        }
        if (lane == 0) { atomicMin(&S.jmin, mn); atomicMax(&S.jmax, mx); }
    }
    __syncthreads();
    const uint32_t jmin = S.jmin, jmax = S.jmax;
    uint32_t m = 0;  // columns ranked so far = columns below the current window
    for (uint32_t lo = jmin; n; lo += kWin) {
        for (uint32_t wd = tid; wd < kWinWords; wd += NT) bw[wd] = 0ull;
        __syncthreads();
        for (uint32_t s = tid; s < (uint32_t)kTlHash; s += NT) {
            const uint32_t key = S.hk[s];
            if (key != kEmpty && key - lo < kWin) atomicOr((unsigned long long *)&bw[(key - lo) >> 6], 1ull << ((key - lo) & 63u));
        }
        __syncthreads();
        if (w == 0) {
            uint32_t c = 0;
            if (lane < (int)(kWinWords / 4))
                c = (uint32_t)(__popcll(bw[4 * lane]) + __popcll(bw[4 * lane + 1]) + __popcll(bw[4 * lane + 2]) + __popcll(bw[4 * lane + 3]));
            const uint32_t inc = wave_inclusive_sum(c);
            if (lane < (int)(kWinWords / 4)) S.bp[lane] = (uint16_t)(inc - c);
            if (lane == 63) S.nz = inc;  // (borrowed: the window's column count)
        }
        __syncthreads();
        const uint32_t total = S.nz;
        for (uint32_t s = tid; s < (uint32_t)kTlHash; s += NT) {
            const uint32_t key = S.hk[s];
            if (key != kEmpty && key - lo < kWin) {
                const uint32_t wd = (key - lo) >> 6, bit = (key - lo) & 63u, w4 = wd & ~3u;
                uint32_t rank = m + (uint32_t)S.bp[wd >> 2] + (uint32_t)__popcll(bw[wd] & ((1ull << bit) - 1ull));
                if (wd > w4) rank += (uint32_t)__popcll(bw[w4]);
                if (wd > w4 + 1) rank += (uint32_t)__popcll(bw[w4 + 1]);
                if (wd > w4 + 2) rank += (uint32_t)__popcll(bw[w4 + 2]);
                // the tile's task count (all waves) goes to its rank's place; each wave's field becomes the count of the waves before it
                const uint32_t sh = 16u * (s & 1u);
                uint32_t run = 0;
#pragma unroll
                for (int v = 0; v < kBwWaves; v++) {
                    const uint32_t c = (S.tc[v][s >> 1] >> sh) & 0xffffu;
                    atomicXor(&S.tc[v][s >> 1], (c ^ run) << sh);  // (xor: the neighbour slot shares the word)
                    run += c;
                }
                S.list[rank] = (uint16_t)run;
                S.rank_of[s] = (uint16_t)rank;
                g.s_cols[off + rank] = key;
            }
        }
        m += total;
        __syncthreads();
        if (jmax - lo < kWin) break;
    }
    // ---- exclusive scan of the task counts in rank order = first task of every C tile relative to the block-row's first ----
    if (w == 0) {
        uint32_t carry = 0;
        for (uint32_t p0 = 0; p0 < m; p0 += 64) {
            const uint32_t p = p0 + (uint32_t)lane;
            const uint32_t v = p < m ? (uint32_t)S.list[p] : 0u;
            const uint32_t inc = wave_inclusive_sum(v);
            if (p < m) {
                S.list[p] = (uint16_t)(carry + inc - v);
                g.s_begin[off + p] = carry + inc - v;
            }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
    }
    for (uint32_t p = tid; p < m; p += NT) g.s_bmps[off + p] = 0ull;
    if (tid == 0) S.nz = 0u;
    // the parked pairs and the zeroed bitmaps are written before other lanes of this workgroup read / OR into them (the workgroup's own CU:
    // a wait for the stores; an agent-scope fence would write the XCD's L2 back)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    // ---- the parked pairs to their places, their products into the tiles' bitmaps: every wave its own ----
    for (uint32_t k0 = 0; k0 < ns; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        if (k < ns) {
            const uint64_t e = g.s_surv[offw + k], pr = g.s_prod[offw + k];
            const uint32_t ord = (uint32_t)g.s_ord[offw + k];
            const uint32_t slot = (uint32_t)(e >> 48);
            const uint32_t rank = (uint32_t)S.rank_of[slot];
            const uint32_t before = (S.tc[w][slot >> 1] >> (16u * (slot & 1u))) & 0xffffu;
            g.s_tasks[off + (uint32_t)S.list[rank] + before + ord] = ((uint64_t)(a0 + (uint32_t)((e >> 32) & 0xffffu)) << 32) | (e & 0xffffffffull);
            atomicOr(&g.s_bmps[off + rank], (unsigned long long)pr);
        }
    }
    // values of the block-row (bits of its finished bitmaps: read past the L1, which may still hold the zeroed lines)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    uint32_t nz = 0;
    for (uint32_t p = tid; p < m; p += NT) nz += (uint32_t)__popcll(__hip_atomic_load(g.s_bmps + off + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    nz = wave_sum(nz);
    if (lane == 0) atomicAdd(&S.nz, nz);
    __syncthreads();
    if (tid == 0) { g.cnt[row] = m; g.surv[row] = ns_all; g.nnz[row] = S.nz; }
}

// stretches -> C's own arrays and the task list: one wave per block-row (WG = false) or, with the workgroup build pass, a workgroup of
// four waves per block-row (the tiles' words and the tasks spread over 256 lanes, the value offsets by the first wave)
template <bool WG>
__global__ __launch_bounds__(kThreads) void rowmerge_copy_kernel(TaskListArgs g)
{
    const int lane = lane_id();
    const uint32_t row = WG ? xcd_order(blockIdx.x, gridDim.x) : xcd_order(blockIdx.x, gridDim.x) * 4 + (uint32_t)wave_id();
    if (row >= g.block_rows) return;
    const uint32_t c0 = g.c_rowptr[row], m = g.c_rowptr[row + 1] - c0;
    if (m == 0) return;
    const uint32_t t0 = g.row_task0[row], ns = g.row_task0[row + 1] - t0;
    const uint64_t off = g.first_pos[g.a_rowptr[row]];
    const uint32_t tid = WG ? threadIdx.x : (uint32_t)lane, nt = WG ? (uint32_t)kThreads : 64u;
    for (uint32_t r = tid; r < m; r += nt) {
        g.c_keys[c0 + r] = key_make(row, g.s_cols[off + r]);
        g.c_bmps[c0 + r] = (uint64_t)g.s_bmps[off + r];
        g.task_begin[c0 + r] = t0 + g.s_begin[off + r];
    }
    for (uint32_t k = tid; k < ns; k += nt) g.tasks[t0 + k] = g.s_tasks[off + k];
    if (!WG || wave_id() == 0) row_value_offsets((const uint64_t *)g.s_bmps + off, m, g.row_val0[row], g.c_offs + c0, lane);
}

// C tile of task 64 w (what the task-list block-MAC kernels index per 64 tasks)
struct COfWave {
    const uint32_t *task_begin;
    uint32_t c_size;
    uint32_t *out;
    __device__ void operator()(uint64_t w) const
    {
        const uint32_t t = (uint32_t)(w * 64);
        uint32_t lo = 0, hi = c_size - 1u;  // last c with task_begin[c] <= t
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (task_begin[mid] <= t) lo = mid;
            else hi = mid - 1u;
        }
        out[w] = lo;
    }
};

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
// C tiles and values per block-row scanned together: (tiles << 33) | values in one 64-bit scan (fewer than 2^27 tiles: values < 2^33)
struct CntNnzIn {
    const uint32_t *cnt, *nnz;
    uint64_t rows;
    __device__ uint64_t operator()(uint64_t i) const { return i < rows ? ((uint64_t)cnt[i] << 33) | (uint64_t)nnz[i] : 0ull; }
};
struct CntNnzOut {
    uint32_t *c_rowptr;
    uint64_t *row_val0;
    uint64_t rows;
    uint32_t *h_c_size;
    uint64_t *h_nnz;
    __device__ void operator()(uint64_t i, uint64_t ex) const
    {
        c_rowptr[i] = (uint32_t)(ex >> 33);
        row_val0[i] = ex & ((1ull << 33) - 1ull);
        if (i == rows) { *h_nnz = ex & ((1ull << 33) - 1ull); *h_c_size = (uint32_t)(ex >> 33); }
    }
};

// most C tiles in a block-row, surviving pairs and (when counted by the pass) candidate pairs of the whole product: one launch, one atomic
// per figure and workgroup
__global__ __launch_bounds__(kThreads) void rows_stats_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ surv, const uint32_t *__restrict__ cand,
                                                               uint32_t rows, unsigned long long *acc)
{
    __shared__ unsigned long long l[3][4];
    unsigned long long mx = 0, ss = 0, sc = 0;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < rows; i += gridDim.x * kThreads) {
        const unsigned long long c = cnt[i];
        mx = c > mx ? c : mx;
        ss += surv[i];
        if (cand) sc += cand[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(mx, d, kWave);
        mx = o > mx ? o : mx;
    }
    ss = wave_sum(ss); sc = wave_sum(sc);
    if (lane_id() == 0) { l[0][wave_id()] = mx; l[1][wave_id()] = ss; l[2][wave_id()] = sc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mx = l[0][w] > mx ? l[0][w] : mx; ss += l[1][w]; sc += l[2][w]; }
        atomicAdd(acc + 0, ss);
        atomicMax(acc + 1, mx);
        if (cand) atomicAdd(acc + 3, sc);
    }
}

struct PublishStats {
    const unsigned long long *acc;  // [0] surviving pairs, [1] most C tiles in a block-row
    const uint32_t *overflow;
    uint64_t *h_surviving, *h_max_over, *h_cand;
    __device__ void operator()(uint64_t) const
    {
        *h_surviving = (uint64_t)acc[0];
        *h_max_over = ((uint64_t)*overflow << 32) | (uint64_t)acc[1];
        *h_cand = (uint64_t)acc[3];
    }
};

// scratch -> C's own arrays: one wave per block-row
__global__ __launch_bounds__(kThreads) void rowmerge_emit_kernel(const uint32_t *__restrict__ tmp_off, const uint32_t *__restrict__ t_cols,
                                                                  const uint64_t *__restrict__ t_bmps, const uint32_t *__restrict__ c_rowptr,
                                                                  const uint64_t *__restrict__ row_val0, uint32_t block_rows, uint32_t stride,
                                                                  uint64_t *__restrict__ c_keys, uint64_t *__restrict__ c_bmps, uint64_t *__restrict__ c_offs)
{
    const uint32_t row = blockIdx.x * 4 + (uint32_t)wave_id();
    if (row >= block_rows) return;
    const uint32_t c0 = c_rowptr[row], n = c_rowptr[row + 1] - c0, t0 = tmp_off ? tmp_off[row] : row * stride;
    for (uint32_t p = (uint32_t)lane_id(); p < n; p += 64) {
        c_keys[c0 + p] = key_make(row, t_cols[t0 + p]);
        c_bmps[c0 + p] = t_bmps[t0 + p];
    }
    row_value_offsets(t_bmps + t0, n, row_val0[row], c_offs + c0, lane_id());
}

struct Cnt64In {
    const uint32_t *cnt;
    uint64_t rows;
    __device__ uint64_t operator()(uint64_t i) const { return i < rows ? (uint64_t)cnt[i] : 0ull; }
};
struct SetU64 {
    uint64_t *p;
    uint64_t v;
    __device__ void operator()(uint64_t) const { *p = v; }
};

}  // namespace

// C's structure (keys, bitmaps, block-row pointer, block count) of A x B by the row-merge pass.  first_pos = T_2's exclusive scan of the
// fan-out per A tile (n_a + 1 entries), total = its last element.  false: some block-row of C holds more tiles than the pass accepts
// (nothing of C was allocated; the caller runs the pipeline).  true: C->keys / bmps / offsets / nnz / rowptr / block_num / max_row_blocks
// are set (the value array is the caller's to allocate), *surviving = candidate pairs that passed the bitmap filter.
bool rowmerge_symbolic(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, uint32_t row_cap,
                       uint64_t *surviving, uint64_t *candidates, hipStream_t st)
{
    // first_pos == nullptr: T_2 did not run (a pair of operands known to fit this pass): every block-row gets row_cap scratch slots and the
    // pass counts the candidate pairs itself (*candidates)
    row_cap = std::min(row_cap, kRowCap);
    const uint64_t rows = (uint64_t)A->num_block_rows();
    if (rows == 0 || rows >= (1ull << 31)) return false;
    if (first_pos && (total == 0 || total >= (1ull << 32))) return false;
    ensure_rowptr(A, st);
    ensure_row_stats(B, st);
    ensure_sym_recs(B, st);
    // (a block-row of C holds at least the tiles of the longest block-row of B it meets: operands with a hub block-row are not tried)
    if (B->max_row_blocks > (int64_t)row_cap) return false;
    const uint64_t slots = first_pos ? std::min<uint64_t>(total, rows * (uint64_t)row_cap) : rows * (uint64_t)row_cap;
    if (slots >= (1ull << 31)) return false;
    // without T_2 every block-row gets row_cap slots whatever it holds: beyond 4 GB of scratch the caller runs T_2 and comes back with
    // scratch sized by the candidate pairs (50 M banded rows would have asked for 19 GB and failed the product on NOMEM)
    if (!first_pos && slots * 12 > (4ull << 30)) return false;
    DevBuf<uint32_t> tmp_off(first_pos ? rows + 1 : 1), cnt(rows + 1), surv_row(rows), nnz_row(rows + 1), cand_row(first_pos ? 1 : rows), t_cols(slots);
    DevBuf<uint64_t> row_val0(rows + 1);
    DevBuf<uint64_t> t_bmps(slots);
    DevBuf<unsigned long long> acc(4);  // [0] surviving pairs, [1] most C tiles in a block-row, [2] overflow flag, [3] candidate pairs
    BMSP_HIP(hipMemsetAsync(acc.p, 0, 32, st));
    if (first_pos) device_exclusive_scan<uint32_t>(RowSlotsIn{first_pos, A->rowptr, rows, row_cap}, PtrOut<uint32_t>{tmp_off.p}, rows + 1, st);
    RowMergeArgs g{};
    g.a_keys = A->keys; g.a_bmps = A->bmps; g.a_rowptr = A->rowptr;
    g.b_keys = B->keys; g.b_bmps = B->bmps; g.b_recs = B->sym_recs; g.b_rowptr = B->rowptr;
    g.block_rows = (uint32_t)rows; g.b_block_rows = (uint32_t)B->num_block_rows(); g.row_cap = row_cap;
    g.tmp_off = first_pos ? tmp_off.p : nullptr; g.cand = first_pos ? nullptr : cand_row.p; g.t_cols = t_cols.p; g.t_bmps = t_bmps.p; g.cnt = cnt.p;
    g.surv = surv_row.p; g.nnz = nnz_row.p; g.overflow = (uint32_t *)(acc.p + 2);
    hipLaunchKernelGGL(rowmerge_symbolic_kernel<1>, dim3((uint32_t)((rows + 3) / 4)), dim3(kThreads), 0, st, g);
    BMSP_CHECK_LAUNCH();
    uint32_t *c_rowptr = (uint32_t *)pool_alloc(sizeof(uint32_t) * (size_t)(rows + 1));
    HostScalar<uint32_t> c_size_h;
    HostScalar<uint64_t> surv_h, mo_h, nnz_h, cand_h;
    uint32_t c_size = 0;
    uint64_t mo = 0, c_nnz = 0;
    try {
        if (slots < (1ull << 27)) {
            device_exclusive_scan<uint64_t>(CntNnzIn{cnt.p, nnz_row.p, rows}, CntNnzOut{c_rowptr, row_val0.p, rows, c_size_h.dev(), nnz_h.dev()}, rows + 1, st);
        } else {
            device_exclusive_scan<uint32_t>(CntIn{cnt.p, rows}, PtrOutTotal<uint32_t>{c_rowptr, rows, c_size_h.dev()}, rows + 1, st);
            device_exclusive_scan<uint64_t>(Cnt64In{nnz_row.p, rows}, PtrOutTotal<uint64_t>{row_val0.p, rows, nnz_h.dev()}, rows + 1, st);
        }
        hipLaunchKernelGGL(rows_stats_kernel, dim3((uint32_t)std::min<uint64_t>((rows + kThreads - 1) / kThreads, 256)), dim3(kThreads), 0, st, cnt.p, surv_row.p,
                           first_pos ? (const uint32_t *)nullptr : cand_row.p, (uint32_t)rows, acc.p);
        BMSP_CHECK_LAUNCH();
        device_for_each(PublishStats{acc.p, g.overflow, surv_h.dev(), mo_h.dev(), cand_h.dev()}, 1, st);
        c_size = c_size_h.wait(st);
        c_nnz = nnz_h.wait(st);
        *surviving = surv_h.wait(st);
        mo = mo_h.wait(st);
        *candidates = first_pos ? total : cand_h.wait(st);
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
    C->offsets = (uint64_t *)pool_alloc(8 * ((size_t)c_size + 1));
    C->nnz = (int64_t)c_nnz;
    device_for_each(SetU64{C->offsets + c_size, c_nnz}, 1, st);
    if (c_size) {
        hipLaunchKernelGGL(rowmerge_emit_kernel, dim3((uint32_t)((rows + 3) / 4)), dim3(kThreads), 0, st, first_pos ? tmp_off.p : (const uint32_t *)nullptr, t_cols.p,
                           t_bmps.p, c_rowptr, row_val0.p, (uint32_t)rows, row_cap, C->keys, C->bmps, C->offsets);
        BMSP_CHECK_LAUNCH();
    }
    return true;
}

namespace {
struct PublishTaskStats {
    const unsigned long long *acc;  // [1] most C tiles in a block-row, [2] overflow flag
    uint64_t *h_max_over;
    __device__ void operator()(uint64_t) const { *h_max_over = ((uint64_t)(uint32_t)acc[2] << 32) | (uint64_t)acc[1]; }
};
struct SetU32 {
    uint32_t *p;
    uint32_t v;
    __device__ void operator()(uint64_t) const { *p = v; }
};
}  // namespace

// C's structure AND the sorted task list of A x B by the build pass and the copy pass.  first_pos = T_2's exclusive scan of the fan-out
// per A tile.  false: some block-row of C holds more tiles (or a block-row more than 65535 surviving pairs) than the pass accepts --
// nothing of C was allocated.  true: C->keys / bmps / rowptr / block_num / max_row_blocks are set; tasks = the surviving pairs
// ((A tile << 32) | B tile) grouped by C tile in C's key order, inside a tile in ascending A tile; task_begin[c] = first task of C tile
// c (c_size + 1 entries); c_of_wave[w] = C tile of task 64 w.
bool rowmerge_tasklist(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, DevBuf<uint64_t> &tasks,
                       DevBuf<uint32_t> &task_begin, DevBuf<uint32_t> &c_of_wave, uint64_t *n_tasks_out, hipStream_t st)
{
    const uint64_t rows = (uint64_t)A->num_block_rows();
    if (rows == 0 || rows >= (1ull << 31) || total == 0 || total >= (1ull << 32)) return false;
    if (total * 42 > (16ull << 30)) return false;  // scratch: 42 bytes per candidate pair
    ensure_rowptr(A, st);
    ensure_row_stats(B, st);
    ensure_sym_recs(B, st);
    // (a block-row of C holds at least the tiles of the longest block-row of B it meets: operands with a hub block-row -- power-law graphs --
    // go to the pipeline without a pass being tried, and without its scratch being allocated)
    if (B->max_row_blocks > (int64_t)(kTlCap + 63)) return false;
    DevBuf<uint32_t> cnt(rows + 1), surv_row(rows + 1), nnz_row(rows + 1), row_task0(rows + 1), s_cols(total), s_begin(total);
    DevBuf<uint64_t> row_val0(rows + 1);
    DevBuf<uint64_t> s_surv(total), s_prod(total), s_tasks(total), s_bmps(total);
    DevBuf<uint16_t> s_ord(total);
    DevBuf<unsigned long long> acc(3);  // [1] most C tiles in a block-row, [2] overflow flag
    BMSP_HIP(hipMemsetAsync(acc.p, 0, 24, st));
    TaskListArgs g{};
    g.w = WalkArgs{A->keys, A->bmps, B->sym_recs, B->rowptr, (uint32_t)B->num_block_rows()};
    g.a_rowptr = A->rowptr; g.first_pos = first_pos; g.block_rows = (uint32_t)rows;
    g.cnt = cnt.p; g.surv = surv_row.p; g.nnz = nnz_row.p; g.overflow = (uint32_t *)(acc.p + 2);
    g.s_surv = s_surv.p; g.s_prod = s_prod.p; g.s_ord = s_ord.p; g.s_tasks = s_tasks.p; g.s_cols = s_cols.p; g.s_begin = s_begin.p;
    g.s_bmps = (unsigned long long *)s_bmps.p;
    const dim3 grid((uint32_t)((rows + 3) / 4));
    // BMSP_RM_BUILD_WAVE=1: one wave per block-row (the form before round 4's workgroup per block-row; A/B runs and tests)
    if (getenv("BMSP_RM_BUILD_WAVE")) {
        if (getenv("BMSP_RM_DEPTH1")) hipLaunchKernelGGL(rowmerge_build_kernel<1>, grid, dim3(kThreads), 0, st, g);  // experiment switch
        else hipLaunchKernelGGL(rowmerge_build_kernel<2>, grid, dim3(kThreads), 0, st, g);
    } else {
        hipLaunchKernelGGL(rowmerge_build_wg_kernel<2>, dim3((uint32_t)rows), dim3(64 * kBwWaves), 0, st, g);
    }
    BMSP_CHECK_LAUNCH();
    uint32_t *c_rowptr = (uint32_t *)pool_alloc(sizeof(uint32_t) * (size_t)(rows + 1));
    HostScalar<uint32_t> c_size_h, n_tasks_h;
    HostScalar<uint64_t> mo_h, nnz_h;
    uint32_t c_size = 0, n_tasks = 0;
    uint64_t mo = 0, c_nnz = 0;
    try {
        if (total < (1ull << 27)) {  // (C tiles <= surviving pairs <= candidate pairs)
            device_exclusive_scan<uint64_t>(CntNnzIn{cnt.p, nnz_row.p, rows}, CntNnzOut{c_rowptr, row_val0.p, rows, c_size_h.dev(), nnz_h.dev()}, rows + 1, st);
        } else {
            device_exclusive_scan<uint32_t>(CntIn{cnt.p, rows}, PtrOutTotal<uint32_t>{c_rowptr, rows, c_size_h.dev()}, rows + 1, st);
            device_exclusive_scan<uint64_t>(Cnt64In{nnz_row.p, rows}, PtrOutTotal<uint64_t>{row_val0.p, rows, nnz_h.dev()}, rows + 1, st);
        }
        device_exclusive_scan<uint32_t>(CntIn{surv_row.p, rows}, PtrOutTotal<uint32_t>{row_task0.p, rows, n_tasks_h.dev()}, rows + 1, st);
        device_max_sum(CntSurvIn{cnt.p}, rows, acc.p + 1, (unsigned long long *)nullptr, st);
        device_for_each(PublishTaskStats{acc.p, mo_h.dev()}, 1, st);
        c_size = c_size_h.wait(st);
        n_tasks = n_tasks_h.wait(st);
        c_nnz = nnz_h.wait(st);
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
    C->offsets = (uint64_t *)pool_alloc(8 * ((size_t)c_size + 1));
    C->nnz = (int64_t)c_nnz;
    device_for_each(SetU64{C->offsets + c_size, c_nnz}, 1, st);
    tasks.alloc(n_tasks);
    task_begin.alloc((size_t)c_size + 1);
    c_of_wave.alloc((size_t)n_tasks / 64 + 1);
    *n_tasks_out = n_tasks;
    device_for_each(SetU32{task_begin.p + c_size, n_tasks}, 1, st);
    if (c_size) {
        g.c_rowptr = c_rowptr; g.row_task0 = row_task0.p;
        g.row_val0 = row_val0.p;
        g.c_keys = C->keys; g.c_bmps = C->bmps; g.c_offs = C->offsets; g.task_begin = task_begin.p; g.tasks = tasks.p;
        if (getenv("BMSP_RM_BUILD_WAVE")) hipLaunchKernelGGL(rowmerge_copy_kernel<false>, grid, dim3(kThreads), 0, st, g);
        else hipLaunchKernelGGL(rowmerge_copy_kernel<true>, dim3((uint32_t)rows), dim3(kThreads), 0, st, g);
        BMSP_CHECK_LAUNCH();
        device_for_each(COfWave{task_begin.p, c_size, c_of_wave.p}, ((uint64_t)n_tasks + 63) / 64, st);
    }
    return true;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(rowmerge)
