// blockmac_strip.hip -- T_7 on the matrix cores, formulated around OPERAND REUSE: one wave owns a strip of two consecutive block-rows
// of C and walks the product row-wise (Gustavson order), k-group by k-group.
//
// Reference: the tensor-core block-MAC kernels multiplyV11..V14 (src/bmSparse_SPGEMM.cu:294-733) stage both tiles of every task
// (:238-276), and so did every kernel here until round 3 (blockmac32.hip): 256 bytes of operand lines per 1024-flop task and half of
// every MFMA wasted on the block-diagonal packing.  Here
//   * the strip's A tiles are the A operand of v_mfma_f32_16x16x32_f16 itself -- rows 0-7 = A(i0, k), rows 8-15 = A(i1, k), four k
//     (a "k-group": four consecutive entries of the merged column list of the two block-rows) along K -- loaded ONCE per k-group and
//     window and kept in registers while the B tiles stream by;
//   * a B tile B(k, j) is loaded ONCE per strip and serves both block-rows: columns 0-7 = B(k, j_2p), columns 8-15 = B(k, j_2p+1);
//     where both block-rows hold A(., k) all four 8x8 quadrants of the 16x16 result are real C tiles (16 block products per
//     instruction instead of 8);
//   * the accumulators of a WINDOW of 32 consecutive C-tile columns of the strip (16 column pairs x 4 VGPRs) stay in registers while
//     every k-group passes; C tiles are written exactly once, compacted by their bitmap.
// The kernel reads the operands' structure (A / B block-row pointers and keys), C's structure (keys, bitmaps, offsets and a block-row
// pointer) and the dense fp16 tile copies; it does NOT read the sorted task list -- the symbolic stages T_3 ... T_9 have fixed what C
// holds, the numeric stage recomputes which (k, j) meet from the two operands.  A candidate pair the bitmap filter dropped
// (multiplication_checker, :742-757) is skipped here by the same test on the cached row / column masks; where it is multiplied
// anyway (the other block-row of the strip needs the tile) its dense product is exactly zero.  Summation order inside a C tile: k
// ascending in groups of four, inside the MFMA in hardware order -- the matrix-core numerics of tc_version 1..4 (exact fp16 products,
// fp32 accumulation; tolerance stated in the tests), never V15's.
//
// Per wave (LDS): the merged k list of the strip {k, A tile of row 0 / row 1, cursor and end of B's block-row k, column mask}, the
// merged column list of the strip's C tiles {j, C tile of row 0 / row 1}, and a 4 x 32 schedule table of the current (window,
// k-group).  Per (window, k-group): 16 lanes per k walk B's block-row k from its cursor (coalesced keys + bitmaps, requested one
// k-group ahead), binary-search the window's columns in LDS and scatter the tile index into the schedule; the MFMA lanes then fetch
// their B line (one 16-byte line per lane) for up to four active column pairs at a time and issue the MFMAs.
//
// Limits (the launcher checks them and keeps the task-list kernels of blockmac32.hip otherwise): <= kKCap merged A tiles and <= kJCap
// merged C tiles per strip, finite operand values (a skipped pair's product must be an exact zero), 32-bit byte offsets.
#include "mac_common.hip.h"

namespace bmsp {
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2k_t __attribute__((ext_vector_type(2)));
typedef float f32x2k_t __attribute__((ext_vector_type(2)));

constexpr int kKCap = 192;  // merged A tiles of a strip (two block-rows)
constexpr int kJCap = 512;  // merged C tiles of a strip
constexpr uint32_t kNone = 0xffffffffu;
constexpr int kStageTile = 72;             // floats per staged C tile: 64 + 8, so the four tiles of a pair start in different LDS banks
constexpr uint32_t kNoTile = 0xffffff00u;  // schedule word of an absent tile: a byte offset no line offset brings back into a buffer

struct StripArgs {
    const uint64_t *a_keys, *a_bmps;
    const uint32_t *a_rowptr;
    const void *a_dense;      // fp16: tiles in position order, 128 B each; fp32: tiles in MFMA lane order (ensure_lane_tiles), 256 B each
    uint32_t a_dense_bytes;
    const uint64_t *b_keys, *b_bmps;
    const uint32_t *b_rowptr;
    const void *b_dense;
    uint32_t b_dense_bytes, b_block_rows, b_blocks;
    const uint64_t *c_keys, *c_bmps, *c_offs;
    const uint32_t *c_rowptr;
    float *c_vals;
    uint32_t block_rows;  // of A and C
    unsigned long long *prof;  // PROF builds: [0] set-up, [1] scans, [2] requests + B lines + MFMAs, [3] window stores, [4] whole wave (timer ticks, summed over waves), [5] waves
};

typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte store at 4-byte alignment

struct alignas(16) StripLds {
    uint32_t jj[kJCap];        // merged column list of the strip's C tiles
    uint16_t jc[kJCap][2];     // C tile of row 0 / row 1 relative to the row's first tile; 0xffff = none
    uint32_t kk[kKCap];        // merged k list
    uint32_t ka[kKCap][2];     // A tile of row 0 / row 1: its byte offset in A's dense copy (kNoTile = none)
    uint32_t kcur[kKCap];      // B's block-row k: next tile not yet consumed
    uint32_t kend[kKCap];
    uint32_t kcm[kKCap];       // OR of the column masks of the strip's A tiles in column k
    union {
        struct {  // merge temporaries
            uint32_t l0[kJCap / 2], l1[kJCap / 2];  // the two sorted lists
            uint16_t nd[kJCap / 2 + 2];             // non-duplicates among the first q entries of l1
        };
        struct {  // after the merges
            float stage[4 * kStageTile];  // the four C tiles of a column pair on their way out
            uint64_t cwb[64], cwo[64];    // bitmap and value offset of the window's C tiles [block-row of the strip][column of the window], requested when
                                          // the window is entered (offset ~0: no tile)
        };
    };
    uint32_t sched[2][4][2][16];  // two tables (item n & 1) x [k slot][column of the pair][pair]: byte offset of the B tile in its dense copy; kNoTile = none
    alignas(16) uint32_t pmask[2][4];  // per k slot: column pairs of the item's window with a tile of B's block-row k
};

__device__ __forceinline__ uint32_t lds_lower_bound(const uint32_t *a, uint32_t n, uint32_t v)
{  // first index with a[idx] >= v
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// merges the sorted lists l0[0..n0) and l1[0..n1) of S (duplicates across the lists collapse); emit(pos, value, idx0 or kNone, idx1 or
// kNone) is called once per list element (an element present in both lists is emitted twice with the same pos, once per side);
// returns the length of the merged list.  Wave-wide; l0 / l1 / nd live in LDS.
template <typename Emit>
__device__ __forceinline__ uint32_t merge_sorted(StripLds &S, uint32_t n0, uint32_t n1, int lane, Emit emit)
{
    // non-duplicate prefix counts of l1
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n1; base += 64) {
        const uint32_t q = base + (uint32_t)lane;
        bool nondup = false;
        if (q < n1) {
            const uint32_t y = S.l1[q];
            const uint32_t p = lds_lower_bound(S.l0, n0, y);
            nondup = !(p < n0 && S.l0[p] == y);
        }
        const uint64_t bal = __ballot(nondup);
        if (q < n1) S.nd[q] = (uint16_t)(carry + (uint32_t)__popcll(bal & lanemask_lt()));
        carry += (uint32_t)__popcll(bal);
    }
    if (lane == 0) S.nd[n1] = (uint16_t)carry;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < n0; base += 64) {
        const uint32_t p = base + (uint32_t)lane;
        if (p < n0) {
            const uint32_t x = S.l0[p];
            emit(p + (uint32_t)S.nd[lds_lower_bound(S.l1, n1, x)], x, p, kNone);
        }
    }
    for (uint32_t base = 0; base < n1; base += 64) {
        const uint32_t q = base + (uint32_t)lane;
        if (q < n1) {
            const uint32_t y = S.l1[q];
            const uint32_t p = lds_lower_bound(S.l0, n0, y);  // for a duplicate: its index in l0
            emit(p + (uint32_t)S.nd[q], y, kNone, q);
        }
    }
    return n0 + carry;
}

// what a k-group's scan needs from memory, requested one k-group ahead: keys and bitmaps of the next 32 tiles of B's block-row k
struct ScanPre {
    uint32_t col[2];   // block column of the tile (the low word of its key)
    uint64_t bmp[2];   // its bitmap: only requested where the bitmap filter can drop a pair (some column of the strip's A tiles is empty)
    uint32_t cur, end;
    bool filter;
};

// OR over each row of 16 lanes, valid in the row's lane 15 (DPP row shifts: no LDS traffic)
__device__ __forceinline__ uint32_t row_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    return v;
}

// PAIRS = column pairs of a window (2 PAIRS C-tile columns of the strip, 4 PAIRS accumulator registers), NB = B lines (column pairs) a wave
// requests together, OCC = waves per SIMD the register allocation is held to
// F32: fp32 operands on v_mfma_f32_16x16x4_f32 -- the k slots of an item are walked one after the other (two instructions per tile: kk 0-3,
// kk 4-7), which is V15's summation order exactly (ascending fmaf chain, established on the hardware by mfma_f32_selftest)
// PROF (BMSP_STRIP_PROF=1, timing builds): where a wave's time goes, by s_memtime stamps at the phase boundaries
template <int PAIRS, int NB, int OCC, bool F32 = false, bool PROF = false>
__global__ __launch_bounds__(kThreads, OCC) void block_mac_strip_kernel(StripArgs g)
{
    const uint64_t t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    uint64_t t_scan = 0, t_store = 0, t_wait = 0;
    __shared__ StripLds lds_all[4];
    const int w = wave_id(), lane = lane_id();
    StripLds &S = lds_all[w];
    // XCD-aware order (as in blockmac32.hip): the workgroups of one XCD take a contiguous eighth of the strips, so that neighbouring
    // strips -- which read the same block-rows of B -- meet in the same L2
    uint32_t wg;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        wg = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    const uint32_t strip = wg * 4 + (uint32_t)w;
    const uint32_t i0 = 2 * strip;
    if (i0 >= g.block_rows) return;
    const bool two = i0 + 1 < g.block_rows;
    const rsrc_t rda = make_rsrc(g.a_dense, g.a_dense_bytes), rdb = make_rsrc(g.b_dense, g.b_dense_bytes);
    const rsrc_t rbk = make_rsrc(g.b_keys, g.b_blocks << 3), rbb = make_rsrc(g.b_bmps, g.b_blocks << 3);
    constexpr uint32_t TS = F32 ? 8u : 7u;  // log2 of a tile's bytes in the operand copies

    // ---- the strip's rows of A and C ----
    const uint32_t a0b = g.a_rowptr[i0], a0e = g.a_rowptr[i0 + 1], a1e = two ? g.a_rowptr[i0 + 2] : a0e;
    const uint32_t c0b = g.c_rowptr[i0], c0e = g.c_rowptr[i0 + 1], c1e = two ? g.c_rowptr[i0 + 2] : c0e;
    const uint32_t n0 = a0e - a0b, n1 = a1e - a0e, m0 = c0e - c0b, m1 = c1e - c0e;
    if (m0 + m1 == 0) return;  // no C tile in the strip: nothing to compute

    // ---- merged k list ----
    for (uint32_t p = (uint32_t)lane; p < n0; p += 64) S.l0[p] = key_col(g.a_keys[a0b + p]);
    for (uint32_t q = (uint32_t)lane; q < n1; q += 64) S.l1[q] = key_col(g.a_keys[a0e + q]);
    for (uint32_t u = (uint32_t)lane; u < (uint32_t)kKCap; u += 64) { S.ka[u][0] = kNoTile; S.ka[u][1] = kNoTile; S.kcm[u] = 0u; }
    __builtin_amdgcn_wave_barrier();
    const uint32_t nK = merge_sorted(S, n0, n1, lane, [&](uint32_t pos, uint32_t k, uint32_t p, uint32_t q) {
        const uint32_t a = p != kNone ? a0b + p : a0e + q;
        S.kk[pos] = k;
        S.ka[pos][p != kNone ? 0 : 1] = a << TS;
        atomicOr(&S.kcm[pos], tile_or_bytes(g.a_bmps[a]));  // column k' of the tile holds a value <=> bit (7 - k')
    });
    __builtin_amdgcn_wave_barrier();
    for (uint32_t u = (uint32_t)lane; u < nK; u += 64) {
        const uint32_t k = S.kk[u];
        const bool in = k < g.b_block_rows;  // no matching block-row in B: no tile
        S.kcur[u] = in ? g.b_rowptr[k] : 0u;
        S.kend[u] = in ? g.b_rowptr[k + 1] : 0u;
    }
    // ---- merged column list of the strip's C tiles ----
    __builtin_amdgcn_wave_barrier();
    for (uint32_t p = (uint32_t)lane; p < m0; p += 64) S.l0[p] = key_col(g.c_keys[c0b + p]);
    for (uint32_t q = (uint32_t)lane; q < m1; q += 64) S.l1[q] = key_col(g.c_keys[c0e + q]);
    for (uint32_t s = (uint32_t)lane; s < (uint32_t)kJCap; s += 64) { S.jc[s][0] = 0xffffu; S.jc[s][1] = 0xffffu; }
    for (uint32_t s = (uint32_t)lane; s < 256u; s += 64) ((uint32_t *)S.sched)[s] = kNoTile;
    if (lane < 8) ((uint32_t *)S.pmask)[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t nJ = merge_sorted(S, m0, m1, lane, [&](uint32_t pos, uint32_t j, uint32_t p, uint32_t q) {
        S.jj[pos] = j;
        if (p != kNone) S.jc[pos][0] = (uint16_t)p;
        else S.jc[pos][1] = (uint16_t)q;
    });
    __builtin_amdgcn_wave_barrier();

    // ---- lane roles ----
    const int ks = lane >> 4;                          // K slot of the MFMA = entry of the k-group
    const int half_sel = (lane >> 3) & 1;              // A operand: block-row of the strip; B operand: column of the pair
    const int line = lane & 7;                         // tile row (A) / tile column (B)
    const uint32_t line16 = (uint32_t)(line * 16);
    const int q16 = lane & 15;                         // scan: tile q16 / q16 + 16 behind the cursor of k slot ks
    const int d_row = lane >> 5;                       // result: block-row of the strip this lane's D values belong to
    const uint32_t d_r0 = 4u * (uint32_t)((lane >> 4) & 1);  // first of its four tile rows
    constexpr uint32_t SLOTS = 2u * PAIRS;  // C-tile columns of a window
    const uint32_t nG = (nK + 3) / 4, nW = (nJ + SLOTS - 1) / SLOTS;
    if (nG == 0) return;  // (C tiles without an A tile cannot exist)

    // one (window, k-group) of the walk = an "item"; items are taken window by window, k-group by k-group
    struct Item {
        uint32_t wi, gi;
        bool on;
    };
    auto next_item = [&](Item it) {
        it.gi++;
        if (it.gi == nG) { it.gi = 0; it.wi++; }
        it.on = it.on && it.wi < nW;
        return it;
    };
    auto request = [&](const Item &it) {
        ScanPre pre;
        const uint32_t u = 4 * it.gi + (uint32_t)ks;
        pre.cur = 0; pre.end = 0; pre.filter = false;
        if (it.on && u < nK) { pre.cur = S.kcur[u]; pre.end = S.kend[u]; pre.filter = S.kcm[u] != 0xffu; }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t t = pre.cur + (uint32_t)(16 * h + q16);
            pre.col[h] = 0; pre.bmp[h] = ~0ull;
            // (little-endian: the key's low word is the block column; a tile past the block-row's end is requested out of range)
            pre.col[h] = __builtin_amdgcn_raw_buffer_load_b32(rbk, t < pre.end ? t << 3 : kOob, 0, 0);
            if (pre.filter) {
                const u32x2k_t bm = __builtin_amdgcn_raw_buffer_load_b64(rbb, t < pre.end ? t << 3 : kOob, 0, 0);
                pre.bmp[h] = ((uint64_t)bm[1] << 32) | bm[0];
            }
        }
        return pre;
    };
    // scan of an item into schedule table tb: the tiles of B's block-rows k (the item's k-group) that fall into the item's window.
    // 16 lanes per k take 32 tiles behind the cursor (keys and bitmaps were requested one item ahead), find the tile's column among
    // the window's C columns (directly when those are consecutive block columns, by binary search in LDS otherwise) and scatter the
    // tile index
    auto scan = [&](const Item &it, uint32_t tb, ScanPre &pre) {
        if (!it.on) return;
        const uint32_t s0 = SLOTS * it.wi, ns = min(SLOTS, nJ - s0);
        const uint32_t jlo = S.jj[s0], jhi = S.jj[s0 + ns - 1];
        const bool dense_win = jhi - jlo == ns - 1u;
        const uint32_t u = 4 * it.gi + (uint32_t)ks;
        const bool u_on = u < nK;
        const uint32_t cm = u_on ? S.kcm[u] : 0u;
        uint32_t mine = 0;
        for (;;) {
            uint32_t hits = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t t = pre.cur + (uint32_t)(16 * h + q16);
                const uint32_t j = pre.col[h];
                const bool inwin = t < pre.end && j <= jhi;
                if (inwin) {
                    uint32_t sl;
                    bool found;
                    if (dense_win) { sl = j - jlo; found = j >= jlo; }
                    else { sl = lds_lower_bound(S.jj + s0, ns, j); found = sl < ns && S.jj[s0 + sl] == j; }
                    // present in C's strip and not dropped by the bitmap filter (the tile's non-empty rows against the columns in use)
                    if (found && (!pre.filter || (cm & tile_or_bytes(pre.bmp[h])) != 0u)) {
                        S.sched[tb][ks][sl & 1u][sl >> 1] = t << TS;
                        mine |= 1u << (sl >> 1);
                    }
                }
                hits += (uint32_t)__popcll((__ballot(inwin) >> (16 * ks)) & 0xffffull);
            }
            pre.cur += hits;
            if (!__any(hits == 32u)) break;
            // every tile fetched for some k slot lay inside the window: there may be more of them.  (On a dense band the 32 tiles end exactly at
            // the window's last column and this second fetch finds nothing -- but it touches the NEXT window's first tiles; leaving it out
            // measured 796 instead of 750 us on the ceiling case.)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t t = pre.cur + (uint32_t)(16 * h + q16);
                pre.col[h] = 0; pre.bmp[h] = ~0ull;
                pre.col[h] = __builtin_amdgcn_raw_buffer_load_b32(rbk, t < pre.end ? t << 3 : kOob, 0, 0);
                if (pre.filter) {
                    const u32x2k_t bm = __builtin_amdgcn_raw_buffer_load_b64(rbb, t < pre.end ? t << 3 : kOob, 0, 0);
                    pre.bmp[h] = ((uint64_t)bm[1] << 32) | bm[0];
                }
            }
        }
        if (u_on && q16 == 0) S.kcur[u] = pre.cur;
        // the column pairs of the window that got a tile of k slot ks: an OR over the slot's 16 lanes (DPP row shifts; an LDS atomic per lane
        // would serialise on one word)
        const uint32_t all = row_or_u32(mine);
        if (q16 == 15) S.pmask[tb][ks] = all;
    };

    Item cur{0u, 0u, true};
    Item nxt = next_item(cur);
    ScanPre pre = request(cur);
    scan(cur, 0u, pre);
    __builtin_amdgcn_wave_barrier();
    pre = request(nxt);
    float4_t acc[PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; p++) acc[p] = float4_t{0.f, 0.f, 0.f, 0.f};
    uint32_t tb = 0;
    const uint64_t t_loop = PROF ? __builtin_readcyclecounter() : 0ull;
    while (cur.on) {
        const Item nn = next_item(nxt);
        // while the item's B lines travel: the scan of the next item into the other table, and the key request of the one after
        auto scan_next = [&]() {
            const uint64_t ts = PROF ? __builtin_readcyclecounter() : 0ull;
            scan(nxt, tb ^ 1u, pre);
            __builtin_amdgcn_wave_barrier();
            pre = request(nn);
            if (PROF) t_scan += __builtin_readcyclecounter() - ts;
        };
        // fp32 kernel: the window's C words (bitmap, value offset of its up to 64 tiles) are requested with the window's first item and parked in
        // LDS at that item's end; the store phase reads them there instead of making four dependent requests per window (A/B on one box, T_7:
        // FEM-like 1306 -> 1186 us, ceiling 1754 -> 1657 us).  The fp16 kernel has no register to spare for it and measured slower.
        uint64_t w_cb = 0, w_co = ~0ull;
        if (F32 && cur.gi == 0) {
            const uint32_t s0w = SLOTS * cur.wi, nsw = min(SLOTS, nJ - s0w);
            const uint32_t wrow = (uint32_t)lane >> 5, wslot = (uint32_t)lane & 31u;
            const uint32_t crel = wslot < nsw ? (uint32_t)S.jc[s0w + wslot][wrow] : 0xffffu;
            if (crel != 0xffffu) {
                const uint32_t c = (wrow ? c0e : c0b) + crel;
                w_cb = g.c_bmps[c]; w_co = g.c_offs[c];
            }
        }
        typedef uint32_t u32x4k_t __attribute__((ext_vector_type(4)));
        const u32x4k_t pm4 = *(const u32x4k_t *)S.pmask[tb];
        bool scanned = false;
        // The column pairs are walked with STATIC indices (accumulator registers cannot be indexed; a switch over the pair number makes
        // the compiler copy the accumulator file around every case) in batches of NB: requests of the batch's active pairs (wave-uniform
        // branches; an absent tile's request goes out of range and moves nothing, so the compiler's wait counts stay exact), then -- once
        // per item -- the next item's scan, then the batch's MFMAs.  The item's schedule was scattered one iteration ago.
        if constexpr (!F32) {
            // the A operand: line `line` of A(row half_sel, k slot ks); the four k slots share every instruction
            const uint32_t u = 4 * cur.gi + (uint32_t)ks;
            const uint32_t a = S.ka[min(u, (uint32_t)kKCap - 1u)][half_sel];  // entries past nK hold kNoTile
            const half8_t fa = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rda, a + line16, 0, 0));
            const uint32_t pm = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pm4[0] | pm4[1] | pm4[2] | pm4[3]));
            const uint32_t *my = S.sched[tb][ks][half_sel];
#pragma unroll
            for (int hb = 0; hb < PAIRS / NB; hb++) {
                const uint32_t mb = (pm >> (NB * hb)) & ((1u << NB) - 1u);
                if (mb) {
                    half8_t fb[NB];
#pragma unroll
                    for (int q = 0; q < NB; q++) {
                        if ((mb >> q) & 1u) {
                            fb[q] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rdb, my[NB * hb + q] + line16, 0, 0));
                        }
                    }
                    if (!scanned) { scan_next(); scanned = true; }
#pragma unroll
                    for (int q = 0; q < NB; q++) {
                        if ((mb >> q) & 1u) acc[NB * hb + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[q], acc[NB * hb + q], 0, 0, 0);
                    }
                }
            }
        } else {
            // lane (kq = ks, line) holds elements (line, kk = kq) and (line, kk = kq + 4) of its tile: 8 consecutive bytes of the lane-ordered copy
            const uint32_t lane8 = (uint32_t)(line * 32 + ks * 8);
            f32x2k_t fa[4];  // (float vectors: __builtin_bit_cast(float, v[1]) of an integer vector's element reads element 0 with this compiler)
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                const uint32_t a = S.ka[min(4 * cur.gi + (uint32_t)k4, (uint32_t)kKCap - 1u)][half_sel];
                fa[k4] = __builtin_bit_cast(f32x2k_t, __builtin_amdgcn_raw_buffer_load_b64(rda, a + lane8, 0, 0));
            }
            // the k slots one after the other, per slot batches of NB column pairs.  (Measured on the FEM-like product, T_7: this order 1317 us;
            // batches of four pairs with the four k slots of a batch requested together 1357 us; 8-pair windows with every line of an item
            // in flight 1424 us -- the request round trips are not what bounds the kernel.)
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                const uint32_t pm = (uint32_t)__builtin_amdgcn_readfirstlane((int)pm4[k4]);
                const uint32_t *my = S.sched[tb][k4][half_sel];
#pragma unroll
                for (int hb = 0; hb < PAIRS / NB; hb++) {
                    const uint32_t mb = (pm >> (NB * hb)) & ((1u << NB) - 1u);
                    if (mb) {
                        f32x2k_t fb[NB];
#pragma unroll
                        for (int q = 0; q < NB; q++) {
                            if ((mb >> q) & 1u) fb[q] = __builtin_bit_cast(f32x2k_t, __builtin_amdgcn_raw_buffer_load_b64(rdb, my[NB * hb + q] + lane8, 0, 0));
                        }
                        if (!scanned) { scan_next(); scanned = true; }
                        if (PROF) {  // (timing build: the wait for the batch's lines apart from its MFMAs)
                            const uint64_t tq = __builtin_readcyclecounter();
                            __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0)
                            t_wait += __builtin_readcyclecounter() - tq;
                        }
#pragma unroll
                        for (int q = 0; q < NB; q++) {
                            if ((mb >> q) & 1u) {
                                float4_t c = acc[NB * hb + q];
                                c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k4][0], fb[q][0], c, 0, 0, 0);  // kk 0 .. 3
                                c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k4][1], fb[q][1], c, 0, 0, 0);  // kk 4 .. 7
                                acc[NB * hb + q] = c;
                            }
                        }
                    }
                }
            }
        }
        if (!scanned) scan_next();
        __builtin_amdgcn_wave_barrier();
        // clear the item's schedule (512 bytes: one 8-byte store per lane); the item after next will scatter into it
        ((uint64_t *)S.sched[tb])[lane] = ((uint64_t)kNoTile << 32) | kNoTile;
        if (lane < 4) S.pmask[tb][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (F32 && cur.gi == 0) {
            S.cwb[lane] = w_cb; S.cwo[lane] = w_co;
            __builtin_amdgcn_wave_barrier();
        }
        if (cur.gi + 1 == nG) {
            const uint64_t tw = PROF ? __builtin_readcyclecounter() : 0ull;
            // ---- the window is complete: lane holds D[4 * (lane >> 4) + i][lane & 15] = rows d_r0 + i of C(row d_row, column 2p + half_sel) ----
            const uint32_t s0 = SLOTS * cur.wi, ns = min(SLOTS, nJ - s0);
            // A lane's four D values are one column of half a tile: through LDS they become 16 consecutive bytes of a tile row, so a
            // full C tile leaves as one 256-byte run (16 lanes x 16 bytes); tiles with holes take the per-value path
            const int o_row = lane >> 5, o_col = (lane >> 4) & 1, o_q = lane & 15;  // outgoing layout: tile (o_row, column o_col of the pair), floats 4 o_q .. 4 o_q + 3
            float *const st_in = S.stage + (d_row * 2 + half_sel) * kStageTile + (int)d_r0 * 8 + line;
            const float *const st_out = S.stage + (o_row * 2 + o_col) * kStageTile + 4 * o_q;
            if constexpr (F32) {
                // pair by pair; the C words of the pair's tiles come from LDS, in the outgoing lane layout (full-tile runs) and in the accumulator
                // layout (per-value stores)
    #pragma unroll
                for (int p = 0; p < PAIRS; p++) {
                    if (2u * (uint32_t)p < ns) {
                        const int wo = o_row * 32 + 2 * p + o_col;
                        const uint64_t ocb = S.cwb[wo], oco = S.cwo[wo];
                        if (__all(ocb == ~0ull || oco == ~0ull)) {
    #pragma unroll
                            for (int i = 0; i < 4; i++) st_in[8 * i] = acc[p][i];
                            __builtin_amdgcn_wave_barrier();
                            const float4_t v = *(const float4_t *)st_out;
                            if (oco != ~0ull) *(float4_u *)(g.c_vals + oco + 4u * (uint32_t)o_q) = v;
                            __builtin_amdgcn_wave_barrier();
                        } else {
                            const int wa = d_row * 32 + 2 * p + half_sel;
                            const uint64_t co = S.cwo[wa], cb = co != ~0ull ? S.cwb[wa] : 0ull;
    #pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const uint32_t pos = (d_r0 + (uint32_t)i) * 8u + (uint32_t)line;
                                if ((cb >> (63u - pos)) & 1ull) g.c_vals[co + (uint64_t)__popcll(cb >> 1 >> (63u - pos))] = acc[p][i];
                            }
                        }
                    }
                    acc[p] = float4_t{0.f, 0.f, 0.f, 0.f};
                }
            } else {
                // (fp16: the window's C words requested four pairs at a time here -- with the words parked at the window's entry the dense-tile
                // ceiling measured 866 instead of 750 us, and a refill-free scan 796: the A/B runs are in DESIGN.md)
                const uint32_t crow0 = d_row ? c0e : c0b, orow0 = o_row ? c0e : c0b;
                constexpr int PG = 4;  // pairs whose C words are requested together
    #pragma unroll
                for (int p0 = 0; p0 < PAIRS; p0 += PG) {
                    if (2u * (uint32_t)p0 < ns) {
                        uint64_t ocb[PG], oco[PG];
    #pragma unroll
                        for (int q = 0; q < PG; q++) {
                            const uint32_t sl = 2u * (uint32_t)(p0 + q) + (uint32_t)o_col;
                            const uint32_t crel = sl < ns ? (uint32_t)S.jc[s0 + sl][o_row] : 0xffffu;
                            ocb[q] = ~0ull; oco[q] = ~0ull;  // no tile: nothing to store, and no reason to leave the fast path
                            if (crel != 0xffffu) { ocb[q] = g.c_bmps[orow0 + crel]; oco[q] = g.c_offs[orow0 + crel]; }
                        }
    #pragma unroll
                        for (int q = 0; q < PG; q++) {
                            const int p = p0 + q;
                            if (2u * (uint32_t)p >= ns) continue;
                            if (__all(ocb[q] == ~0ull)) {
    #pragma unroll
                                for (int i = 0; i < 4; i++) st_in[8 * i] = acc[p][i];
                                __builtin_amdgcn_wave_barrier();
                                const float4_t v = *(const float4_t *)st_out;
                                if (oco[q] != ~0ull) *(float4_u *)(g.c_vals + oco[q] + 4u * (uint32_t)o_q) = v;
                                __builtin_amdgcn_wave_barrier();
                            } else {
                                const uint32_t sl = 2u * (uint32_t)p + (uint32_t)half_sel;
                                const uint32_t crel = sl < ns ? (uint32_t)S.jc[s0 + sl][d_row] : 0xffffu;
                                if (crel != 0xffffu) {
                                    const uint32_t c = crow0 + crel;
                                    const uint64_t cb = g.c_bmps[c], co = g.c_offs[c];
    #pragma unroll
                                    for (int i = 0; i < 4; i++) {
                                        const uint32_t pos = (d_r0 + (uint32_t)i) * 8u + (uint32_t)line;
                                        if ((cb >> (63u - pos)) & 1ull) g.c_vals[co + (uint64_t)__popcll(cb >> 1 >> (63u - pos))] = acc[p][i];
                                    }
                                }
                            }
                        }
                    }
    #pragma unroll
                    for (int q = 0; q < PG; q++) acc[p0 + q] = float4_t{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (PROF) t_store += __builtin_readcyclecounter() - tw;
        }
        cur = nxt; nxt = nn;
        tb ^= 1u;
    }
    if (PROF && lane == 0) {
        const uint64_t t_end = __builtin_readcyclecounter();
        atomicAdd(g.prof + 0, (unsigned long long)(t_loop - t_begin));
        atomicAdd(g.prof + 1, (unsigned long long)t_scan);
        atomicAdd(g.prof + 2, (unsigned long long)(t_end - t_loop - t_scan - t_store));
        atomicAdd(g.prof + 3, (unsigned long long)t_store);
        atomicAdd(g.prof + 4, (unsigned long long)(t_end - t_begin));
        atomicAdd(g.prof + 5, 1ull);
        atomicAdd(g.prof + 6, (unsigned long long)t_wait);
    }
}

// values that are not finite (exponent all ones): a skipped candidate pair must contribute an exact zero
struct NonFiniteF16 {
    const uint16_t *v;
    uint32_t *flag;
    __device__ void operator()(uint64_t i) const
    {
        if ((v[i] & 0x7c00u) == 0x7c00u) *flag = 1u;
    }
};
// fp32: the largest biased exponent, and 255 - the smallest biased exponent among the non-zero values, in one reduction each
struct ExpMaxF32 {
    const uint32_t *v;
    __device__ uint64_t operator()(uint64_t i) const { return (uint64_t)((v[i] >> 23) & 0xffu); }
};
struct ExpMinF32 {
    const uint32_t *v;
    __device__ uint64_t operator()(uint64_t i) const
    {
        const uint32_t w = v[i] & 0x7fffffffu;
        return w ? (uint64_t)(255u - (w >> 23)) : 0ull;  // explicit zeros do not count: their products are exact zeros
    }
};

}  // namespace

void ensure_finite_flag(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->values_finite >= 0) return;
    if (m->dtype == BMSP_F64 || m->nnz == 0) { m->values_finite = 1; return; }
    const uint64_t base = m->view_values_end ? read_back(m->offsets, st) : 0;
    const uint64_t n = (uint64_t)m->values_extent() - base;
    if (m->dtype == BMSP_F16) {
        DevBuf<uint32_t> flag(1);
        BMSP_HIP(hipMemsetAsync(flag.p, 0, 4, st));
        device_for_each(NonFiniteF16{(const uint16_t *)m->values + base, flag.p}, n, st);
        m->values_finite = read_back(flag.p, st) ? 0 : 1;
        return;
    }
    // fp32: the exponent range as well -- the fp32 MFMA reproduces the fmaf chain bit for bit only while no product underflows
    // (measured: products around 2^-149 round differently in the matrix pipe) and none overflows
    DevBuf<unsigned long long> mx(2);
    BMSP_HIP(hipMemsetAsync(mx.p, 0, 16, st));
    device_max_sum(ExpMaxF32{(const uint32_t *)m->values + base}, n, mx.p, (unsigned long long *)nullptr, st);
    device_max_sum(ExpMinF32{(const uint32_t *)m->values + base}, n, mx.p + 1, (unsigned long long *)nullptr, st);
    unsigned long long h[2];
    read_back_bytes(h, mx.p, 16, st);
    m->f32_exp_max = (int)h[0];
    m->f32_exp_min = h[1] ? 255 - (int)h[1] : 255;  // no non-zero value: nothing can underflow
    m->values_finite = h[0] == 255ull ? 0 : 1;
}

// what the strip kernel needs of the operands alone: fp16 tiles the K = 32 MFMA path addresses, a strip's merged A tiles within the k
// list, finite values (cached per-matrix figures; the read-backs behind them happen once per matrix)
bool mac_strip_operands_ok(bmsp_matrix_s *A, bmsp_matrix_s *B, hipStream_t st)
{
    if (A->dtype == BMSP_F16) {
        if (!mac_mfma32_supported(A, B)) return false;
    } else if (A->dtype == BMSP_F32) {
        // 256-byte tiles behind 32-bit byte offsets; the instruction's summation order is checked on the device once per process
        if (B->dtype != BMSP_F32 || A->block_num >= (1ll << 24) || B->block_num >= (1ll << 24) || !mac_f32_mfma_usable(st)) return false;
        // every product a normal number (|a| |b| >= 2^(ea + eb - 254) >= 2^-126) and every sum far from overflow: outside that range the
        // matrix pipe's rounding is not the fmaf chain's, and V15's vector-ALU kernel (task-list mode or the pipeline) takes the product
        ensure_finite_flag(A, st);
        ensure_finite_flag(B, st);
        if (A->f32_exp_min + B->f32_exp_min < mac_f32_exp_floor(st) || A->f32_exp_max + B->f32_exp_max > 254 + 100) return false;
    } else {
        return false;
    }
    if ((uint64_t)A->num_block_rows() >= (1ull << 31)) return false;
    ensure_row_stats(A, st);
    if (2 * A->max_row_blocks > kKCap) return false;
    ensure_finite_flag(A, st);
    ensure_finite_flag(B, st);
    return A->values_finite == 1 && B->values_finite == 1;
}

// ... and of C: a strip's merged C tiles within the column list (two block-rows of at most mac_strip_row_cap() tiles always fit)
uint32_t mac_strip_row_cap() { return (uint32_t)kJCap / 2u; }
bool mac_strip_fits_c(bmsp_matrix_s *C, hipStream_t st)
{
    if (C->block_num >= (1ll << 31) || C->block_num == 0) return false;
    ensure_row_stats(C, st);
    return 2 * C->max_row_blocks <= kJCap;
}

// whether the strip kernel takes a product whose task list exists (the task-list kernels are the alternative)
bool mac_strip_eligible(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, uint64_t candidates, uint64_t n_tasks, hipStream_t st)
{
    const char *force = getenv("BMSP_MAC_STRIP");  // 0 / 1: experiment and test switch (the capacity limits still hold)
    if (force && force[0] == '0') return false;
    if (!force) {
        // The walk visits every candidate pair and pays a fixed price per (window, k-group) item: it wins where items are fat -- C tiles
        // that collect many tasks from candidate pairs that nearly all survive (dense-tile ceiling: 32.8 tasks per C tile, 840 us against
        // 1430 us for the direct kernel).  On the FEM-like product (14 tasks per C tile, 61 % of the pairs survive, 3 MFMAs per item) the
        // direct kernel stays ahead (530 vs 640 us), so the bar sits between the two.
        if (n_tasks < 20 * (uint64_t)C->block_num || 10 * n_tasks < 9 * candidates) return false;
    }
    return mac_strip_fits_c(C, st) && mac_strip_operands_ok(A, B, st);
}

// the numeric stages that need no task list: the row-sparse kernel (V15 numerics on nearly empty tiles) or the strip kernels
bool mac_structure_numeric_ok(bmsp_matrix_s *A, bmsp_matrix_s *B, int tc_version, hipStream_t st)
{
    if (mac_rowsparse_applies(A, B, tc_version, st)) return true;
    if (A->dtype == BMSP_F16 && tc_version != 4) return false;  // the fp16 strip kernel has the matrix cores' numerics
    return mac_strip_operands_ok(A, B, st);
}

int launch_mac_strip(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, int tc_version, hipStream_t st)
{
    const bool f32 = A->dtype == BMSP_F32;
    // V15 numerics on nearly empty tiles: only the scalar products that exist (blockmac_rowsparse.hip)
    if (!getenv("BMSP_STRIP_PROF") && mac_rowsparse_applies(A, B, tc_version, st)) {
        launch_mac_rowsparse(A, B, C, st);
        return BMSP_MAC_ROWSPARSE;
    }
    if (f32) { ensure_lane_tiles(A, st); ensure_lane_tiles(B, st); }
    else { ensure_dense_tiles(A, st); ensure_dense_tiles(B, st); }
    ensure_rowptr(A, st);
    ensure_rowptr(B, st);
    ensure_rowptr(C, st);
    StripArgs g{};
    const uint32_t tile_bytes = f32 ? 256u : 128u;
    g.a_keys = A->keys; g.a_bmps = A->bmps; g.a_rowptr = A->rowptr;
    g.a_dense = f32 ? A->lane_tiles : A->dense_tiles; g.a_dense_bytes = (uint32_t)A->block_num * tile_bytes;
    g.b_keys = B->keys; g.b_bmps = B->bmps; g.b_rowptr = B->rowptr;
    g.b_dense = f32 ? B->lane_tiles : B->dense_tiles; g.b_dense_bytes = (uint32_t)B->block_num * tile_bytes;
    g.b_block_rows = (uint32_t)B->num_block_rows(); g.b_blocks = (uint32_t)B->block_num;
    g.c_keys = C->keys; g.c_bmps = C->bmps; g.c_offs = C->offsets; g.c_rowptr = C->rowptr; g.c_vals = (float *)C->values;
    g.block_rows = (uint32_t)A->num_block_rows();
    const uint32_t strips = (g.block_rows + 1) / 2;
    // Measured on MI355X (T_7, us; dense-tile ceiling / FEM-like forced): 16 pairs per window, 8 lines requested together, 3 waves per SIMD
    // 840 / 637; 16 lines together at 2 waves per SIMD 975 / 736, at 3 waves (28 spilled registers) 1580 / 878; 8-pair windows 881 / 662;
    // one workgroup per strip (shared tables, windows dealt to the four waves, cursors re-seeded per window) 1058 / 692.  Timing-only builds
    // of the 803 us kernel on the ceiling case: the next item's scan run twice +208 us, the B lines requested out of range -99 us; C tiles
    // leaving as 256-byte runs through LDS instead of per-value stores: 803 -> 776 us.
    const dim3 grid((strips + 3) / 4);
    if (getenv("BMSP_STRIP_PROF")) {  // timing build: per-phase timer ticks, printed per launch
        DevBuf<unsigned long long> prof(8);
        BMSP_HIP(hipMemsetAsync(prof.p, 0, 64, st));
        g.prof = prof.p;
        if (f32) hipLaunchKernelGGL((block_mac_strip_kernel<16, 8, 3, true, true>), grid, dim3(kThreads), 0, st, g);
        else hipLaunchKernelGGL((block_mac_strip_kernel<16, 8, 3, false, true>), grid, dim3(kThreads), 0, st, g);
        BMSP_CHECK_LAUNCH();
        unsigned long long h[8];
        BMSP_HIP(hipMemcpyAsync(h, prof.p, 64, hipMemcpyDeviceToHost, st));
        BMSP_HIP(hipStreamSynchronize(st));
        const double w = (double)(h[5] ? h[5] : 1);
        fprintf(stderr, "strip prof (ticks per wave, %llu waves): set-up %.0f  scans %.0f  requests+lines+mfma %.0f (of which waiting for lines, fp32 build: %.0f)  stores %.0f  whole %.0f\n", h[5], h[0] / w,
                h[1] / w, h[2] / w, h[6] / w, h[3] / w, h[4] / w);
        return BMSP_MAC_STRIP;
    }
    if (f32) hipLaunchKernelGGL((block_mac_strip_kernel<16, 8, 3, true>), grid, dim3(kThreads), 0, st, g);
    else if (getenv("BMSP_STRIP_WIDE")) hipLaunchKernelGGL((block_mac_strip_kernel<16, 16, 2>), grid, dim3(kThreads), 0, st, g);  // experiment switch
    else hipLaunchKernelGGL((block_mac_strip_kernel<16, 8, 3>), grid, dim3(kThreads), 0, st, g);
    BMSP_CHECK_LAUNCH();
    return BMSP_MAC_STRIP;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(blockmac_strip)
