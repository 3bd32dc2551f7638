// rowwindow.hip -- C's structure AND the sorted task list of A x B for operands with hub block-rows (power-law graphs: R-MAT), where a
// block-row of C holds thousands of tiles: one WORKGROUP per (block-row of A, window of C's block columns), dense tables in LDS indexed by
// the column inside the window.
//
// Reference: bmSparse_mult (src/bmSparse_SPGEMM.cu:849-1164) expands every candidate pair (:884-932), filters (:742-757, :944-948), sorts
// the survivors by C key (:963-1024; bb_segsort's bin for segments > 2048, include/bb_segsort-master/bb_comput_l.h:1155-1284), reduces the
// keys (:1040-1062) and ORs the tile products (:787-810, :1067-1107).  rowmerge.hip forms a block-row of C in a wave's hash table and
// stops at ~900 tiles per block-row; on R-MAT 2^16 x 8 a block-row of C holds 1700 tiles on average (C is a fifth full at tile
// granularity) and every product went to the expand - sort - compress pipeline (5 ms of candidate-sized passes).  Here the SORT is replaced
// by addressing: inside a window of at most kWinSlots block columns a C tile's slot is (column - window start), so
//   * count pass: every surviving pair bumps its column's task count and ORs its tile product into the column's bitmap (LDS atomics, any
//     order); the non-empty columns, in slot order = C's key order, are the window's C tiles;
//   * fill pass: the tasks of a C tile must appear in ascending A tile (V15's summation order, :269-273).  The A tiles of the block-row
//     are taken 64 at a time; a surviving pair sets bit (A tile - first of the round) in its column's 64-bit hit mask; after a barrier
//     its place inside the C tile is (tasks of earlier rounds) + popcount(hit mask below its bit) -- no order dependence between waves.
// Long block-rows are cut into more windows than the tables need (<= kWinCand candidate pairs per window on a uniform split), so a hub
// block-row is spread over many workgroups.  The result is the pipeline's, bit for bit: keys, bitmaps, offsets, task order.
#include "matrix.h"
#include "prims.hip.h"
#include "bmsp_bits.h"
#include <algorithm>
#include <vector>

namespace bmsp {
namespace {

constexpr int kWinSlots = 2048;           // block columns per window (dense table entries)
constexpr uint32_t kWinCand = 16u << 10;  // candidate pairs per window a block-row is cut for (uniform split of its columns)
constexpr uint32_t kWinGran = 64;         // window edges are multiples of it: B's column index (builder.hip: ensure_col_index) answers them with one load, and the column mass is kept per granule
static_assert(kWinSlots % kWinGran == 0, "a full-width window ends on the grid");
constexpr int kHashSlots = 4096;          // slots of a hashed window's table
constexpr uint32_t kHashCap = 3072;       // distinct C tiles a hashed window holds (every wave checks before a step: <= 8 x 64 more may arrive)
constexpr uint32_t kWinCandHash = 2304;   // candidate pairs per window for block-rows with hashed windows (tiles <= candidate pairs <= 0.75 cap on average)
constexpr uint32_t kHashEmpty = 0xffffffffu;

struct WinUnit {
    uint32_t row, lo, hi;  // block-row of A / C, block columns [lo, hi) of C
    uint32_t scr;          // first scratch slot of the window's tile list (hi - lo slots)
    uint32_t a0, a1;       // the block-row's A tiles
    uint32_t tab, stride;  // stretch table: entry (A tile a, this window's left edge) = tab + (a - a0) * stride; the right edge follows it
};

// ---- windows of a block-row ----------------------------------------------------------------------------------------------------------
// A block-row with `cand` candidate pairs is cut into windows of (roughly) equal candidate counts: their edges sit at quantiles of B's
// COLUMN MASS (tiles of B per 64 block columns, prefix-summed: matrix.h col_mass) -- on a power-law operand the first columns are met by
// every block-row and a uniform cut gave windows of 3 x the average (measured on R-MAT 2^16).  A window of at most kWinSlots block
// columns is DENSE (slot = column - left edge; it cannot overflow); a wider one is HASHED (open addressing over kHashSlots slots) and
// holds at most kHashCap distinct C tiles -- block-rows whose windows would be that wide are cut for kWinCandHash candidate pairs per
// window instead of kWinCand.  Edges are multiples of kWinGran (the grid of B's column index and of the column mass).
struct WinPlan {
    const uint32_t *mass;  // G + 1 prefix sums of B's tiles per granule of kWinGran block columns
    uint32_t G, ncols;
    uint32_t cw_dense, cw_hash;
};
__device__ __forceinline__ uint32_t mass_lower_bound(const uint32_t *mass, uint32_t lo, uint32_t hi, uint64_t target)
{  // smallest g in [lo, hi] with mass[g] >= target (mass ascends; mass[hi] >= target is the caller's business)
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)mass[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// calls emit(q, lo, hi) for every window (block columns [lo, hi)), in order; returns their number
template <typename Emit>
__device__ __forceinline__ uint32_t row_windows(const WinPlan &P, uint64_t cand, Emit emit)
{
    if (cand == 0 || P.ncols == 0) return 0;
    // dense tables for the whole block-row when that takes few windows (narrow operands) or leaves every window enough to do; else windows
    // by candidate count alone, hashed where they come out wider than a dense table
    const uint64_t n_fit = (P.ncols + (uint32_t)kWinSlots - 1) / (uint32_t)kWinSlots;
    const bool dense_row = n_fit <= 16 || cand >= n_fit * 1024;
    uint64_t n_t = dense_row ? (cand + P.cw_dense - 1) / P.cw_dense : (cand + P.cw_hash - 1) / P.cw_hash;
    if (dense_row && n_t < n_fit) n_t = n_fit;
    const uint64_t total = P.mass[P.G];
    uint32_t g_lo = 0, count = 0;
    for (uint64_t q = 1; g_lo < P.G; q++) {
        uint32_t g_hi = P.G;
        if (q < n_t && total) g_hi = mass_lower_bound(P.mass, g_lo, P.G, (total * q + n_t - 1) / n_t);
        if (g_hi <= g_lo) g_hi = g_lo + 1;
        if (dense_row && g_hi - g_lo > (uint32_t)kWinSlots / kWinGran) g_hi = g_lo + (uint32_t)kWinSlots / kWinGran;
        if (g_hi > P.G) g_hi = P.G;
        const uint32_t lo = g_lo * kWinGran, hi = g_hi * kWinGran < P.ncols ? g_hi * kWinGran : P.ncols;
        emit(count, lo, hi);
        count++;
        g_lo = g_hi;
    }
    return count;
}

struct RowCand {
    const uint64_t *first_pos;
    const uint32_t *a_rowptr;
    __device__ uint64_t operator()(uint64_t i) const { return first_pos[a_rowptr[i + 1]] - first_pos[a_rowptr[i]]; }
};
// windows per block-row (one loop over the quantiles per block-row; the scans below read the counts)
struct CountWindows {
    RowCand rc;
    WinPlan P;
    uint32_t *n_win;
    __device__ void operator()(uint64_t i) const { n_win[i] = row_windows(P, rc(i), [](uint32_t, uint32_t, uint32_t) {}); }
};
struct NWinIn {
    const uint32_t *n_win;
    uint64_t rows;
    __device__ uint32_t operator()(uint64_t i) const { return i < rows ? n_win[i] : 0u; }
};
// entries of the stretch table of a block-row: (window edges) x (A tiles)
struct PlanTableIn {
    const uint32_t *n_win, *a_rowptr;
    uint64_t rows;
    __device__ uint64_t operator()(uint64_t i) const
    {
        if (i >= rows || !n_win[i]) return 0ull;
        return (uint64_t)(n_win[i] + 1u) * (uint64_t)(a_rowptr[i + 1] - a_rowptr[i]);
    }
};
struct EmitUnits {
    RowCand rc;
    WinPlan P;
    const uint32_t *n_win, *unit_first;
    const uint64_t *tab_first;
    WinUnit *units;
    __device__ void operator()(uint64_t i) const
    {
        const uint32_t n = n_win[i];
        if (!n) return;
        const uint32_t u0 = unit_first[i], a0 = rc.a_rowptr[i], a1 = rc.a_rowptr[i + 1], tab = (uint32_t)tab_first[i];
        WinUnit *const out = units;
        row_windows(P, rc(i), [&](uint32_t q, uint32_t lo, uint32_t hi) { out[u0 + q] = WinUnit{(uint32_t)i, lo, hi, 0u, a0, a1, tab + q, n + 1u}; });
    }
};
// scratch slots of a window's tile list: its width (dense) or the hashed table's capacity
struct UnitScrIn {
    const WinUnit *units;
    const uint32_t *u_cand;
    uint64_t n;
    __device__ uint64_t operator()(uint64_t u) const
    {
        if (u >= n) return 0ull;
        const uint32_t w = units[u].hi - units[u].lo, cap = w <= (uint32_t)kWinSlots ? w : (uint32_t)kHashSlots;
        // C tiles <= candidate pairs (counted only when the tables' sizes add up to too much); bits 40..: hashed windows (counted along)
        return (uint64_t)(u_cand && u_cand[u] < cap ? u_cand[u] : cap) | (w <= (uint32_t)kWinSlots ? 0ull : 1ull << 40);
    }
};
struct UnitScrOut {
    WinUnit *units;
    uint64_t n;
    uint64_t *total;
    __device__ void operator()(uint64_t u, uint64_t ex) const
    {
        if (u == n) { *total = ex; return; }
        units[u].scr = (uint32_t)(ex & ((1ull << 40) - 1ull));
    }
};

// The stretch table: for every A tile of a block-row and every window edge of that block-row, the first tile of B's block-row (the A
// tile's column) at or beyond the edge.  One thread per A tile; long block-rows of B answer from their column index, short ones are
// walked.  Built once per product: every window of the block-row, in both passes, then cuts B's block-rows with ONE load per A tile
// (before: join record -> column index / keys, two dependent round trips per round and pass).
struct BuildStretch {
    const uint32_t *a_rowptr, *n_win, *unit_first;
    const WinUnit *units;
    uint32_t ncols;
    const uint64_t *a_keys, *b_keys;
    const uint32_t *b_rowptr, *b_idx_row, *b_idx;
    uint32_t b_block_rows;
    const uint64_t *tab_first;
    uint32_t *tab;
    uint32_t *u_cand;  // per window: its candidate pairs (what bounds its C tiles: the size of its scratch)
    __device__ void operator()(uint64_t a) const
    {
        const uint64_t ak = a_keys[a];
        const uint32_t i = key_row(ak), k = key_col(ak);
        const uint32_t n = n_win[i];
        if (!n) return;
        const uint32_t u0 = unit_first[i];
        const WinUnit *const w = units + u0;
        uint32_t bb = 0, be = 0, off = ~0u;
        if (k < b_block_rows) {
            bb = b_rowptr[k]; be = b_rowptr[k + 1];
            if (b_idx_row) off = b_idx_row[k];
        }
        uint32_t *out = tab + tab_first[i] + (uint64_t)(a - a_rowptr[i]) * (n + 1u);
        uint32_t pos = bb, prev = bb;  // edges ascend: a short block-row is walked once
        for (uint32_t q = 0; q <= n; q++) {
            const uint32_t edge = q < n ? w[q].lo : ncols;
            uint32_t p;
            if (q == 0) p = bb;
            else if (edge >= ncols) p = be;
            else if (off != ~0u) p = b_idx[off + edge / kWinGran];
            else {
                while (pos < be && key_col(b_keys[pos]) < edge) pos++;
                p = pos;
            }
            out[q] = p;
            if (u_cand && q && p > prev) atomicAdd(&u_cand[u0 + q - 1], p - prev);
            prev = p;
        }
    }
};

struct WinArgs {
    const uint64_t *a_keys, *a_bmps;
    const uint32_t *stretch;  // the stretch table (BuildStretch)
    const uint32_t *a_rowptr;
    const uint64_t *b_keys, *b_bmps;
    const uint32_t *b_recs;   // per B tile {bitmap row-major (lo, hi), block column, rows the tile uses}: all a pass needs, one 16-byte load
    const uint32_t *b_rowptr;
    const uint32_t *b_idx_row, *b_idx;  // column index of B's long block-rows (null: binary search): b_idx[b_idx_row[k] + c / kWinGran] = first tile of block-row k with column >= c
    uint32_t b_block_rows;
    const WinUnit *units;
    uint32_t n_units;
    // tile lists of the windows (count pass -> fill pass), at WinUnit::scr: column, tasks, bitmap of every C tile in column order
    uint32_t *t_col, *t_cnt;
    uint64_t *t_bmp;
    uint32_t *u_tiles, *u_surv, *u_nnz;  // per window: C tiles (~0: a hashed window overflowed), surviving pairs, values
    uint32_t *overflow;                  // set when a hashed window overflowed
    // fill pass
    const uint32_t *tile_base, *task_base;  // per window: first C tile, first task
    const uint64_t *val_base;               //             first value
    uint64_t *c_keys, *c_bmps, *c_offs;
    uint32_t *task_begin, *c_of_wave;
    uint64_t *tasks;
    unsigned long long *prof;  // BMSP_WIN_PROF=1: per window {clocks of the count pass, of its look-ups, clocks of the fill pass, of its look-ups}
};

#ifndef BMSP_WIN_WAVES
#define BMSP_WIN_WAVES 8
#endif
constexpr int kWinWaves = BMSP_WIN_WAVES;       // waves per workgroup
constexpr int kWinThreads = 64 * kWinWaves;
constexpr int kWinBatch = 4;                    // steps whose records a wave requests together, fill pass (8-byte half records)
constexpr int kWinBatchCount = 4;               // ... count pass (16-byte records)

// A ROUND = 64 consecutive A tiles of the block-row.  Their stretches of B's block-rows inside the window (one stretch-table load per A
// tile) are cut into STEPS of 64 tiles; the round's steps are dealt to the workgroup's waves in turn (step i to wave i mod 8), so that one
// hub block-row of B among the 64 does not leave seven waves waiting at the round's barrier, and a wave requests the records of several
// steps -- of whatever A tiles -- together (the passes are bound by memory round trips: ~0.75 us each, measured).  Every wave loads the
// round's 64 table entries itself (lane v = A tile v of the round): no look-up phase to share, no barrier for it.
typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));

struct RoundLoad {  // what a lane requests for its A tile of a round
    uint32_t s0, s1;
    uint64_t abm;
};
__device__ __forceinline__ RoundLoad round_load(const WinArgs &g, const WinUnit &u, uint32_t r0, int lane)
{
    RoundLoad r{0u, 0u, 0ull};
    const uint32_t a = r0 + (uint32_t)lane;
    if (a < u.a1) {
        const uint32_t *e = g.stretch + u.tab + (size_t)(a - u.a0) * u.stride;
        r.s0 = e[0]; r.s1 = e[1];
        r.abm = g.a_bmps[a];
    }
    return r;
}

// lane v holds A tile v of the round; P = steps of the round in front of the tile
struct RoundRegs {
    uint32_t seg, len, alo, ahi, P;
    uint32_t S;  // steps of the round (uniform)
};
__device__ __forceinline__ RoundRegs round_regs(const RoundLoad &r)
{
    RoundRegs q;
    q.seg = r.s0; q.len = r.s1 - r.s0;
    q.alo = (uint32_t)r.abm; q.ahi = (uint32_t)(r.abm >> 32);
    const uint32_t steps = (q.len + 63u) >> 6;
    const uint32_t inc = wave_inclusive_sum(steps);
    q.P = inc - steps;
    q.S = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    return q;
}

// One A tile per step, 64 tiles of B's block-row: everything that depends on the A tile alone -- its bitmap, the columns it uses, the
// byte masks of bmp_calculator -- is wave-uniform (scalar registers, scalar ALU), and a lane is left with one record load, the filter
// (one AND) and its table update.  (A first form laid the stretches of 8 or 16 A tiles end to end, 64 pairs per step whatever the
// tile: every lane then ran the whole 8x8 boolean product -- 130 vector instructions per step, 411 M per product on R-MAT 2^16.)
struct StepScalars {
    uint32_t v, s0, len, x0, ah, al, cols;  // A tile of the round, its stretch, first pair of the step, its bitmap, its columns in use
};
// (no branch on "is there a step i": everything below is scalar code on values read with v_readlane, and a branch would make the compiler
// keep the results in vector registers -- measured: 157 vector instructions per step, the product's masks built with quarter-rate
// v_mul_lo_u32.  Beyond the round's last step the ballot names the last A tile and x0 lies past its stretch: no lane is live.)
__device__ __forceinline__ StepScalars step_scalars(const RoundRegs &q, uint32_t i)
{
    StepScalars t;
    // last A tile whose first step is <= i (tiles without a step share their start with the next tile: the last of a run owns it)
    t.v = (uint32_t)__popcll(__ballot(q.P <= i)) - 1u;
    t.s0 = (uint32_t)__builtin_amdgcn_readlane((int)q.seg, (int)t.v);
    t.len = (uint32_t)__builtin_amdgcn_readlane((int)q.len, (int)t.v);
    t.x0 = (i - (uint32_t)__builtin_amdgcn_readlane((int)q.P, (int)t.v)) << 6;
    t.ah = (uint32_t)__builtin_amdgcn_readlane((int)q.ahi, (int)t.v);
    t.al = (uint32_t)__builtin_amdgcn_readlane((int)q.alo, (int)t.v);
    t.cols = tile_or_bytes(((uint64_t)t.ah << 32) | (uint64_t)t.al);  // bit 7-k: column k of the A tile holds a value
    return t;
}

// exclusive scan of one value per thread across the kWinThreads-thread workgroup (lds: kWinWaves words; two barriers inside)
__device__ __forceinline__ uint32_t win_block_exclusive_sum(uint32_t v, uint32_t *lds, uint32_t &total)
{
    const uint32_t inc = wave_inclusive_sum(v);
    if (lane_id() == kWave - 1) lds[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWinWaves; w++) {
        const uint32_t s = lds[w];
        if (w < wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// ---- slots --------------------------------------------------------------------------------------------------------------------------------
// dense window: the column's distance from the left edge.  Hashed window: open addressing in `keys` (kHashEmpty = free)
template <bool HASH, int T>
__device__ __forceinline__ uint32_t slot_insert(uint32_t *keys, uint32_t col, uint32_t lo, bool &fresh)
{
    fresh = false;
    if (!HASH) return col - lo;
    uint32_t s = (col * 0x9E3779B1u) >> (32 - __builtin_ctz((unsigned)T));
    for (;;) {
        const uint32_t old = atomicCAS(&keys[s], kHashEmpty, col);
        if (old == kHashEmpty || old == col) { fresh = old == kHashEmpty; return s; }
        s = (s + 1u) & (uint32_t)(T - 1);
    }
}
template <bool HASH, int T>
__device__ __forceinline__ uint32_t slot_find(const uint32_t *keys, uint32_t col, uint32_t lo)
{
    if (!HASH) return col - lo;
    uint32_t s = (col * 0x9E3779B1u) >> (32 - __builtin_ctz((unsigned)T));
    while (keys[s] != col) s = (s + 1u) & (uint32_t)(T - 1);  // (the column is in the table: the fill pass entered every C tile of the window)
    return s;
}

// ---- count pass: C tiles of the window (columns, task counts, bitmaps) -------------------------------------------------------------------
// Windows are taken in launch order: consecutive workgroups go to different XCDs, so the hub block-rows of a power-law operand (its first
// block-rows: R-MAT's first eighth of the rows carries half of the candidate pairs) are spread over all eight.  (An XCD-contiguous
// assignment, as the row-merge passes use for banded operands, gave one XCD half of the work: 2.9 + 3.9 ms instead of 1.9 + 2.6 ms.)
// Two instantiations run over the same list of windows: <kWinSlots, false> takes the dense ones, <kHashSlots, true> the hashed ones.
template <int T, bool HASH>
__global__ __launch_bounds__(kWinThreads) void rowwin_count_kernel(WinArgs g)
{
    __shared__ uint32_t cnt[T];
    __shared__ uint64_t bmp[T];
    __shared__ uint32_t keys[HASH ? T : 1];
    __shared__ uint32_t order[HASH ? T : 1];  // hashed: (column << 12 | slot) of the occupied slots, sorted = C's key order
    __shared__ uint32_t red[kWinWaves], red2[2 * kWinWaves], n_distinct;
    static_assert(!HASH || T == 4096, "the sort word keeps the slot in 12 bits");
    constexpr int PT = T / kWinThreads;
    const uint32_t unit = blockIdx.x;
    const WinUnit u = g.units[unit];
    if ((u.hi - u.lo > (uint32_t)kWinSlots) != HASH) return;  // (block-uniform) the other instantiation's window
    const int w = wave_id(), lane = lane_id();
    for (int s = threadIdx.x; s < T; s += kWinThreads) { cnt[s] = 0u; bmp[s] = 0ull; if (HASH) keys[s] = kHashEmpty; }
    if (threadIdx.x == 0) n_distinct = 0u;
    const u32x4w *recs = (const u32x4w *)g.b_recs;
    const unsigned long long clk0 = g.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    RoundLoad nxt = round_load(g, u, u.a0, lane);
    __syncthreads();  // the tables are clear
    bool over = false;
    // (no barrier inside: the waves walk the rounds on their own, each with the next round's table entries already requested)
    for (uint32_t r0 = u.a0; r0 < u.a1 && !over; r0 += 64u) {
        const RoundRegs q = round_regs(nxt);
        if (r0 + 64u < u.a1) nxt = round_load(g, u, r0 + 64u, lane);
        for (uint32_t i0 = (uint32_t)w; i0 < q.S && !over; i0 += (uint32_t)(kWinWaves * kWinBatchCount)) {
            StepScalars t[kWinBatchCount];
            u32x4w r[kWinBatchCount];
#pragma unroll
            for (int b = 0; b < kWinBatchCount; b++) {
                t[b] = step_scalars(q, i0 + (uint32_t)(kWinWaves * b));
                r[b] = u32x4w{0u, 0u, 0u, 0u};
                if (t[b].x0 + (uint32_t)lane < t[b].len) r[b] = recs[t[b].s0 + t[b].x0 + (uint32_t)lane];
            }
#pragma unroll
            for (int b = 0; b < kWinBatchCount; b++) {
                if (HASH) {  // room for this step's new columns?  (kWinWaves x 64 may arrive between a wave's look and its inserts)
                    if (__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&n_distinct) > (int)kHashCap) over = true;
                    if (over) break;
                }
                // multiplication_checker (:742-757): a column of the A tile meets a row of the B tile (no record beyond the stretch: rows in use = 0)
                const bool keep = (t[b].cols & r[b][3]) != 0u;
                bool fresh = false;
                if (keep) {
                    const uint32_t slot = slot_insert<HASH, T>(keys, r[b][2], u.lo, fresh);
                    atomicAdd(&cnt[slot], 1u);
                    atomicOr((unsigned long long *)&bmp[slot], (unsigned long long)tile_product_scalar_a(t[b].ah, t[b].al, t[b].cols, r[b][1], r[b][0]));
                }
                if (HASH) {
                    const uint32_t nf = (uint32_t)__popcll(__ballot(fresh));
                    if (nf && lane == 0) atomicAdd(&n_distinct, nf);
                }
            }
        }
    }
    __syncthreads();
    if (HASH && n_distinct > kHashCap) {  // (block-uniform) more C tiles than the table holds: the host cuts the block-rows finer and runs the pass again
        if (threadIdx.x == 0) { g.u_tiles[unit] = ~0u; g.u_surv[unit] = 0u; g.u_nnz[unit] = 0u; atomicOr(g.overflow, 1u); }
        return;
    }
    uint32_t tiles, sv = 0, nz = 0;
    if (HASH) {
        // C's key order: the occupied slots sorted by column
        for (int s = threadIdx.x; s < T; s += kWinThreads) order[s] = keys[s] == kHashEmpty ? 0xffffffffu : (keys[s] << 12) | (uint32_t)s;
        __syncthreads();
        bitonic_words<uint32_t, kWinThreads, true>(order, (uint32_t)T, threadIdx.x);
        tiles = n_distinct;
        for (uint32_t rk = threadIdx.x; rk < tiles; rk += kWinThreads) {
            const uint32_t wd = order[rk], s = wd & 0xfffu;
            g.t_col[u.scr + rk] = wd >> 12;
            g.t_cnt[u.scr + rk] = cnt[s];
            g.t_bmp[u.scr + rk] = bmp[s];
            sv += cnt[s];
            nz += (uint32_t)__popcll(bmp[s]);
        }
    } else {
        // the non-empty columns in slot order: thread t owns slots [PT t, PT t + PT)
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = cnt[threadIdx.x * PT + i];
            mine += c ? 1u : 0u;
            sv += c;
            nz += (uint32_t)__popcll(bmp[threadIdx.x * PT + i]);
        }
        uint32_t pos = u.scr + win_block_exclusive_sum(mine, red, tiles);
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t s = threadIdx.x * PT + i, c = cnt[s];
            if (c) {
                g.t_col[pos] = u.lo + s;
                g.t_cnt[pos] = c;
                g.t_bmp[pos] = bmp[s];
                pos++;
            }
        }
    }
    sv = wave_sum(sv); nz = wave_sum(nz);
    if (lane == 0) { red2[w] = sv; red2[kWinWaves + w] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s1 = 0, s2 = 0;
        for (int b = 0; b < kWinWaves; b++) { s1 += red2[b]; s2 += red2[kWinWaves + b]; }
        g.u_tiles[unit] = tiles;
        g.u_surv[unit] = s1;
        g.u_nnz[unit] = s2;
        if (g.prof) { g.prof[4 * unit] = __builtin_amdgcn_s_memtime() - clk0; g.prof[4 * unit + 1] = 0; }
    }
}

// ---- fill pass: C's arrays for the window's tiles, and its tasks in (column, A tile) order ------------------------------------------------
// HASH: one mask table (three barriers per round instead of two: the table is cleared after the round's second half) -- the key array
// takes the second table's place in LDS
template <int T, bool HASH>
__global__ __launch_bounds__(kWinThreads) void rowwin_fill_kernel(WinArgs g)
{
    __shared__ uint32_t begin[T];    // first free task slot of the column's C tile, relative to the window's first task
    __shared__ uint64_t hit2[HASH ? 1 : 2][T];  // A tiles of the round (bit = A tile - the round's first) that meet the column; dense windows alternate between two tables
    __shared__ uint32_t keys[HASH ? T : 1];
    __shared__ uint32_t red[kWinWaves];
    constexpr int PT = T / kWinThreads;
    const uint32_t unit = blockIdx.x;
    const WinUnit u = g.units[unit];
    if ((u.hi - u.lo > (uint32_t)kWinSlots) != HASH) return;
    const uint32_t n = g.u_tiles[unit];
    if (n == 0) return;
    const int w = wave_id(), lane = lane_id();
    const uint32_t tb = g.tile_base[unit], kb = g.task_base[unit];
    const uint64_t vb = g.val_base[unit];
    for (int s = threadIdx.x; s < T; s += kWinThreads) {
        hit2[0][s] = 0ull;
        if constexpr (!HASH) hit2[1][s] = 0ull;
        if (HASH) keys[s] = kHashEmpty;
    }
    if (HASH) __syncthreads();
    // the window's tile list -> C's arrays; thread t owns tiles [PT t, PT t + PT)
    {
        uint32_t col[PT], c[PT];
        uint64_t bm[PT];
        uint32_t sc = 0, sn = 0;
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t r = threadIdx.x * PT + i;
            col[i] = 0; c[i] = 0; bm[i] = 0;
            if (r < n) { col[i] = g.t_col[u.scr + r]; c[i] = g.t_cnt[u.scr + r]; bm[i] = g.t_bmp[u.scr + r]; }
            sc += c[i];
            sn += (uint32_t)__popcll(bm[i]);
        }
        uint32_t tot;
        uint32_t ec = win_block_exclusive_sum(sc, red, tot);
        uint32_t en = win_block_exclusive_sum(sn, red, tot);
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t r = threadIdx.x * PT + i;
            if (r < n) {
                g.c_keys[tb + r] = key_make(u.row, col[i]);
                g.c_bmps[tb + r] = bm[i];
                g.c_offs[tb + r] = vb + (uint64_t)en;   // T_9's popcount scan (:1113-1130)
                g.task_begin[tb + r] = kb + ec;          // T_6's task ranges (:1040-1062)
                // C tile of task 64 m, for every such task in this tile's range (what the task-list block-MAC kernels index per 64 tasks)
                for (uint32_t m = (kb + ec + 63u) >> 6; (m << 6) < kb + ec + c[i]; m++) g.c_of_wave[m] = tb + r;
                bool fresh;
                begin[slot_insert<HASH, T>(keys, col[i], u.lo, fresh)] = ec;
            }
            ec += c[i];
            en += (uint32_t)__popcll(bm[i]);
        }
    }
    const u32x2w *recs = (const u32x2w *)g.b_recs;  // {block column, rows the tile uses} = the upper half of a record
    const unsigned long long clk0 = g.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    RoundLoad nxt = round_load(g, u, u.a0, lane);
    for (uint32_t r0 = u.a0, par = 0; r0 < u.a1; r0 += 64u, par ^= (HASH ? 0u : 1u)) {
        const RoundRegs q = round_regs(nxt);
        if (r0 + 64u < u.a1) nxt = round_load(g, u, r0 + 64u, lane);
        __syncthreads();  // the previous round's second half is over (first round: C's arrays, `begin` and the keys are written, the masks are zero)
        uint64_t *const hit = hit2[par];
        // the wave's first kWinBatch steps: their records stay in registers for the second half
        StepScalars t0[kWinBatch];
        u32x2w f[kWinBatch];
#pragma unroll
        for (int b = 0; b < kWinBatch; b++) {
            t0[b] = step_scalars(q, (uint32_t)w + (uint32_t)(kWinWaves * b));
            f[b] = u32x2w{0u, 0u};
            if (t0[b].x0 + (uint32_t)lane < t0[b].len) f[b] = recs[2u * (t0[b].s0 + t0[b].x0 + (uint32_t)lane) + 1u];
        }
        // the previous round's marks become task slots taken, while this round's first records travel
        if (r0 != u.a0) {
            uint64_t *const prev = hit2[HASH ? 0u : (par ^ 1u)];
            for (int s = threadIdx.x; s < T; s += kWinThreads) {
                const uint64_t h = prev[s];
                if (h) { begin[s] += (uint32_t)__popcll(h); prev[s] = 0ull; }
            }
            if (HASH) __syncthreads();  // one table: it is clear before this round marks it
        }
        // first half: the round's A tiles mark the columns they reach
#pragma unroll
        for (int b = 0; b < kWinBatch; b++)
            if (t0[b].cols & f[b][1]) atomicOr((unsigned long long *)&hit[slot_find<HASH, T>(keys, f[b][0], u.lo)], 1ull << t0[b].v);
        for (uint32_t i0 = (uint32_t)w + (uint32_t)(kWinWaves * kWinBatch); i0 < q.S; i0 += (uint32_t)(kWinWaves * kWinBatch)) {
            StepScalars t[kWinBatch];
            u32x2w r[kWinBatch];
#pragma unroll
            for (int b = 0; b < kWinBatch; b++) {
                t[b] = step_scalars(q, i0 + (uint32_t)(kWinWaves * b));
                r[b] = u32x2w{0u, 0u};
                if (t[b].x0 + (uint32_t)lane < t[b].len) r[b] = recs[2u * (t[b].s0 + t[b].x0 + (uint32_t)lane) + 1u];
            }
#pragma unroll
            for (int b = 0; b < kWinBatch; b++)
                if (t[b].cols & r[b][1]) atomicOr((unsigned long long *)&hit[slot_find<HASH, T>(keys, r[b][0], u.lo)], 1ull << t[b].v);
        }
        __syncthreads();
        // second half: a pair's place inside its C tile = tasks of earlier rounds + marks of smaller A tiles of this round
        auto place = [&](const StepScalars &t, const u32x2w &r) {
            if (t.cols & r[1]) {
                const uint32_t slot = slot_find<HASH, T>(keys, r[0], u.lo);
                g.tasks[kb + begin[slot] + (uint32_t)__popcll(hit[slot] & ((1ull << t.v) - 1ull))] =
                    ((uint64_t)(r0 + t.v) << 32) | (uint64_t)(t.s0 + t.x0 + (uint32_t)lane);
            }
        };
#pragma unroll
        for (int b = 0; b < kWinBatch; b++) place(step_scalars(q, (uint32_t)w + (uint32_t)(kWinWaves * b)), f[b]);
        for (uint32_t i0 = (uint32_t)w + (uint32_t)(kWinWaves * kWinBatch); i0 < q.S; i0 += (uint32_t)(kWinWaves * kWinBatch)) {
            StepScalars t[kWinBatch];
            u32x2w r[kWinBatch];
#pragma unroll
            for (int b = 0; b < kWinBatch; b++) {
                t[b] = step_scalars(q, i0 + (uint32_t)(kWinWaves * b));
                r[b] = u32x2w{0u, 0u};
                if (t[b].x0 + (uint32_t)lane < t[b].len) r[b] = recs[2u * (t[b].s0 + t[b].x0 + (uint32_t)lane) + 1u];
            }
#pragma unroll
            for (int b = 0; b < kWinBatch; b++) place(t[b], r[b]);
        }
    }
    if (g.prof && threadIdx.x == 0) { g.prof[4 * unit + 2] = __builtin_amdgcn_s_memtime() - clk0; g.prof[4 * unit + 3] = 0; }
}

struct TilesSurvIn {
    const uint32_t *tiles, *surv;
    uint64_t n;
    __device__ uint64_t operator()(uint64_t i) const { return i < n ? ((uint64_t)tiles[i] << 32) | (uint64_t)surv[i] : 0ull; }
};
struct TilesSurvOut {
    uint32_t *tile_base, *task_base;
    uint64_t n;
    uint32_t *h_tiles, *h_tasks;
    __device__ void operator()(uint64_t i, uint64_t ex) const
    {
        tile_base[i] = (uint32_t)(ex >> 32);
        task_base[i] = (uint32_t)ex;
        if (i == n) { *h_tiles = (uint32_t)(ex >> 32); *h_tasks = (uint32_t)ex; }
    }
};
struct CntU32In {
    const uint32_t *p;
    uint64_t n;
    __device__ uint32_t operator()(uint64_t i) const { return i < n ? p[i] : 0u; }
};
struct CntU64In {
    const uint32_t *p;
    uint64_t n;
    __device__ uint64_t operator()(uint64_t i) const { return i < n ? (uint64_t)p[i] : 0ull; }
};
struct RowPtrOfUnits {
    const uint32_t *unit_first, *tile_base;
    uint32_t *c_rowptr;
    __device__ void operator()(uint64_t i) const { c_rowptr[i] = tile_base[unit_first[i]]; }
};
struct SetOne64 {
    uint64_t *p;
    uint64_t v;
    __device__ void operator()(uint64_t) const { *p = v; }
};
struct SetOne32 {
    uint32_t *p;
    uint32_t v;
    __device__ void operator()(uint64_t) const { *p = v; }
};

}  // namespace

// true: C->keys / bmps / offsets / nnz / rowptr / block_num are set; tasks = the surviving pairs ((A tile << 32) | B tile) grouped by C
// tile in C's key order, inside a tile in ascending A tile; task_begin[c] = first task of C tile c (c_size + 1 entries); c_of_wave[w] =
// C tile of task 64 w -- what rowmerge_tasklist and the pipeline's T_3 .. T_9 produce.  false: not applicable (B wider than
// kWinMaxPerRow windows, or too much scratch); nothing of C was allocated.
bool rowmerge_windowed(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *first_pos, uint64_t total, DevBuf<uint64_t> &tasks,
                       DevBuf<uint32_t> &task_begin, DevBuf<uint32_t> &c_of_wave, uint64_t *n_tasks_out, hipStream_t st)
{
    const uint64_t rows = (uint64_t)A->num_block_rows();
    const uint64_t ncols64 = (uint64_t)B->num_block_cols();
    if (rows == 0 || rows >= (1ull << 31) || total == 0 || total >= (1ull << 32)) return false;
    if (ncols64 == 0 || ncols64 >= (1ull << 20)) return false;  // (a hashed window sorts (column << 12 | slot) words)
    const uint32_t ncols = (uint32_t)ncols64;
    const char *ce = getenv("BMSP_WIN_CAND");  // experiment / test switches: candidate pairs per window a block-row is cut for
    const char *he = getenv("BMSP_WIN_CAND_HASH");
    ensure_rowptr(A, st);
    ensure_rowptr(B, st);
    ensure_sym_recs(B, st);
    ensure_col_index(B, kWinGran, st);
    ensure_col_mass(B, kWinGran, st);
    const RowCand rc{first_pos, A->rowptr};
    WinPlan P{B->col_mass, (uint32_t)((ncols + kWinGran - 1) / kWinGran), ncols, ce ? (uint32_t)std::max(1, atoi(ce)) : kWinCand,
              he ? (uint32_t)std::max(1, atoi(he)) : kWinCandHash};
    // a pair of operands whose hashed windows overflowed once is cut finer from the start the next time
    const bool same_pair = A->rm_partner_uid == B->uid;
    if (same_pair && A->rm_partner_cw_hash && !he) P.cw_hash = (uint32_t)A->rm_partner_cw_hash;
    DevBuf<uint32_t> n_win(rows + 1), unit_first(rows + 1), overflow(1);
    DevBuf<uint64_t> tab_first(rows + 1);
    DevBuf<WinUnit> units;
    DevBuf<uint32_t> stretch, t_col, t_cnt, u_tiles, u_surv, u_nnz, u_cand;
    DevBuf<uint64_t> t_bmp;
    DevBuf<unsigned long long> prof;
    WinArgs g{};
    uint32_t U = 0;
    uint64_t n_hashed = 0;
    for (int attempt = 0;; attempt++) {
        device_for_each(CountWindows{rc, P, n_win.p}, rows, st);
        HostScalar<uint32_t> n_units_h;
        HostScalar<uint64_t> n_tab_h, n_scr_h;
        device_exclusive_scan<uint32_t>(NWinIn{n_win.p, rows}, PtrOutTotal<uint32_t>{unit_first.p, rows, n_units_h.dev()}, rows + 1, st);
        device_exclusive_scan<uint64_t>(PlanTableIn{n_win.p, A->rowptr, rows}, PtrOutTotal<uint64_t>{tab_first.p, rows, n_tab_h.dev()}, rows + 1, st);
        U = n_units_h.wait(st);
        const uint64_t NT = n_tab_h.wait(st);
        if (getenv("BMSP_WIN_DEBUG")) fprintf(stderr, "[win] attempt %d: %u windows, %llu table entries, cw_hash %u\n", attempt, U, (unsigned long long)NT, P.cw_hash);
        if (U == 0 || U >= (1u << 31) || NT >= (1ull << 32)) return false;
        units.alloc(U);
        device_for_each(EmitUnits{rc, P, n_win.p, unit_first.p, tab_first.p, units.p}, rows, st);
        // Where the windows are too thin the decomposition does not pay: every window of a block-row looks at every A tile of it, in both
        // passes, and with a handful of candidate pairs per (A tile, window) a step of 64 lanes is nearly empty (R-MAT scale 22, edge factor 1:
        // 4.9 pairs per table entry -- 329 ms for the passes against 86 ms for expand - sort - compress; R-MAT 2^16 x 8: 66 per entry).
        if (total < 16 * NT && !getenv("BMSP_WIN_THIN")) return false;
        stretch.alloc(NT);
        // scratch of a window's tile list: the size of its table -- or, where those add up to too much (wide operands), its candidate pairs
        // counted while the stretch table is built
        uint64_t S = 0;
        n_hashed = 0;
        {
            HostScalar<uint64_t> cap_h;
            device_exclusive_scan<uint64_t>(UnitScrIn{units.p, nullptr, U}, UnitScrOut{units.p, U, cap_h.dev()}, (uint64_t)U + 1, st);
            S = cap_h.wait(st);
            n_hashed = S >> 40;
            S &= (1ull << 40) - 1ull;
        }
        const bool count_cand = S * 16 > (4ull << 30);
        if (count_cand) {
            u_cand.alloc((size_t)U + 1);
            BMSP_HIP(hipMemsetAsync(u_cand.p, 0, 4 * ((size_t)U + 1), st));
        }
        device_for_each(BuildStretch{A->rowptr, n_win.p, unit_first.p, units.p, ncols, A->keys, B->keys, B->rowptr, B->col_index_row, B->col_index,
                                     (uint32_t)B->num_block_rows(), tab_first.p, stretch.p, count_cand ? u_cand.p : (uint32_t *)nullptr},
                        (uint64_t)A->block_num, st);
        if (count_cand) {
            device_exclusive_scan<uint64_t>(UnitScrIn{units.p, u_cand.p, U}, UnitScrOut{units.p, U, n_scr_h.dev()}, (uint64_t)U + 1, st);
            S = n_scr_h.wait(st) & ((1ull << 40) - 1ull);
        }
        if (getenv("BMSP_WIN_DEBUG")) fprintf(stderr, "[win] scratch %llu entries\n", (unsigned long long)S);
        if (S >= (1ull << 32) || S * 16 > (64ull << 30)) return false;
        t_col.alloc(S); t_cnt.alloc(S); t_bmp.alloc(S);
        u_tiles.alloc((size_t)U + 1); u_surv.alloc((size_t)U + 1); u_nnz.alloc((size_t)U + 1);
        BMSP_HIP(hipMemsetAsync(overflow.p, 0, 4, st));
        g = WinArgs{};
        g.stretch = stretch.p;
        g.a_keys = A->keys; g.a_bmps = A->bmps; g.a_rowptr = A->rowptr;
        g.b_keys = B->keys; g.b_bmps = B->bmps; g.b_recs = B->sym_recs; g.b_rowptr = B->rowptr; g.b_idx_row = B->col_index_row; g.b_idx = B->col_index;
        g.b_block_rows = (uint32_t)B->num_block_rows();
        g.units = units.p; g.n_units = U;
        g.t_col = t_col.p; g.t_cnt = t_cnt.p; g.t_bmp = t_bmp.p; g.u_tiles = u_tiles.p; g.u_surv = u_surv.p; g.u_nnz = u_nnz.p; g.overflow = overflow.p;
        if (getenv("BMSP_WIN_PROF")) {
            prof.alloc(4 * (size_t)U);
            BMSP_HIP(hipMemsetAsync(prof.p, 0, 32 * (size_t)U, st));
            g.prof = prof.p;
        }
        hipLaunchKernelGGL((rowwin_count_kernel<kWinSlots, false>), dim3(U), dim3(kWinThreads), 0, st, g);
        BMSP_CHECK_LAUNCH();
        if (!n_hashed) break;  // dense windows cannot overflow
        hipLaunchKernelGGL((rowwin_count_kernel<kHashSlots, true>), dim3(U), dim3(kWinThreads), 0, st, g);
        BMSP_CHECK_LAUNCH();
        if (read_back(overflow.p, st) == 0u) break;
        // a hashed window met more distinct C tiles than its table holds: every block-row with hashed windows is cut twice as fine
        if (getenv("BMSP_WIN_DEBUG")) fprintf(stderr, "[win] a hashed window overflowed\n");
        if (attempt == 3 || P.cw_hash <= 64u) return false;
        P.cw_hash /= 2;
        A->rm_partner_uid = B->uid; A->rm_partner_blocks = B->block_num; A->rm_partner_cw_hash = (int64_t)P.cw_hash;
    }
    DevBuf<uint32_t> tile_base((size_t)U + 1), task_base((size_t)U + 1);
    DevBuf<uint64_t> val_base((size_t)U + 1);
    HostScalar<uint32_t> c_size_h, n_tasks_h;
    HostScalar<uint64_t> nnz_h;
    // (C tiles <= surviving pairs <= candidate pairs < 2^32: both running sums fit the halves of one 64-bit scan)
    device_exclusive_scan<uint64_t>(TilesSurvIn{u_tiles.p, u_surv.p, U}, TilesSurvOut{tile_base.p, task_base.p, U, c_size_h.dev(), n_tasks_h.dev()}, (uint64_t)U + 1, st);
    device_exclusive_scan<uint64_t>(CntU64In{u_nnz.p, U}, PtrOutTotal<uint64_t>{val_base.p, U, nnz_h.dev()}, (uint64_t)U + 1, st);
    const uint32_t c_size = c_size_h.wait(st), n_tasks = n_tasks_h.wait(st);
    const uint64_t c_nnz = nnz_h.wait(st);
    C->block_num = c_size;
    C->rowptr = (uint32_t *)pool_alloc(sizeof(uint32_t) * (size_t)(rows + 1));
    C->rowptr_rows = (int64_t)rows;
    C->max_row_blocks = -1;  // (ensure_row_stats looks when somebody asks)
    C->keys = (uint64_t *)pool_alloc(8 * (size_t)(c_size ? c_size : 1));
    C->bmps = (uint64_t *)pool_alloc(8 * (size_t)(c_size ? c_size : 1));
    C->offsets = (uint64_t *)pool_alloc(8 * ((size_t)c_size + 1));
    C->nnz = (int64_t)c_nnz;
    device_for_each(RowPtrOfUnits{unit_first.p, tile_base.p, C->rowptr}, rows + 1, st);
    device_for_each(SetOne64{C->offsets + c_size, c_nnz}, 1, st);
    tasks.alloc(n_tasks);
    task_begin.alloc((size_t)c_size + 1);
    c_of_wave.alloc((size_t)n_tasks / 64 + 1);
    *n_tasks_out = n_tasks;
    device_for_each(SetOne32{task_begin.p + c_size, n_tasks}, 1, st);
    if (c_size) {
        g.tile_base = tile_base.p; g.task_base = task_base.p; g.val_base = val_base.p;
        g.c_keys = C->keys; g.c_bmps = C->bmps; g.c_offs = C->offsets; g.task_begin = task_begin.p; g.c_of_wave = c_of_wave.p; g.tasks = tasks.p;
        hipLaunchKernelGGL((rowwin_fill_kernel<kWinSlots, false>), dim3(U), dim3(kWinThreads), 0, st, g);
        BMSP_CHECK_LAUNCH();
        if (n_hashed) {
            hipLaunchKernelGGL((rowwin_fill_kernel<kHashSlots, true>), dim3(U), dim3(kWinThreads), 0, st, g);
            BMSP_CHECK_LAUNCH();
        }
    }
    if (g.prof) {  // experiment: the slowest windows, with what they hold
        std::vector<unsigned long long> hp(4 * (size_t)U);
        std::vector<WinUnit> hu(U);
        std::vector<uint32_t> hs(U), ht(U);
        BMSP_HIP(hipMemcpyAsync(hp.data(), prof.p, 32 * (size_t)U, hipMemcpyDeviceToHost, st));
        BMSP_HIP(hipMemcpyAsync(hu.data(), units.p, sizeof(WinUnit) * (size_t)U, hipMemcpyDeviceToHost, st));
        BMSP_HIP(hipMemcpyAsync(hs.data(), u_surv.p, 4 * (size_t)U, hipMemcpyDeviceToHost, st));
        BMSP_HIP(hipMemcpyAsync(ht.data(), u_tiles.p, 4 * (size_t)U, hipMemcpyDeviceToHost, st));
        std::vector<uint32_t> rp(rows + 1);
        BMSP_HIP(hipMemcpyAsync(rp.data(), A->rowptr, 4 * (rows + 1), hipMemcpyDeviceToHost, st));
        BMSP_HIP(hipStreamSynchronize(st));
        std::vector<uint32_t> order(U);
        for (uint32_t i = 0; i < U; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hp[4 * a] + hp[4 * a + 2] > hp[4 * b] + hp[4 * b + 2]; });
        unsigned long long tc = 0, tf = 0;
        for (uint32_t i = 0; i < U; i++) { tc += hp[4 * i]; tf += hp[4 * i + 2]; }
        fprintf(stderr, "[win prof] %u windows; clocks summed over windows: count %llu, fill %llu (s_memtime ticks)\n", U, tc, tf);
        for (uint32_t i = 0; i < 12 && i < U; i++) {
            const uint32_t q = order[i];
            fprintf(stderr, "[win prof] row %6u cols [%5u,%5u) a_len %5u survivors %8u tiles %5u | count %9llu (look-ups %9llu) fill %9llu (look-ups %9llu)\n", hu[q].row, hu[q].lo,
                    hu[q].hi, rp[hu[q].row + 1] - rp[hu[q].row], hs[q], ht[q], hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]);
        }
    }
    return true;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(rowwindow)
