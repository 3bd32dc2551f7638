// blockmac32.hip -- T_7 on the gfx950 matrix cores: v_mfma_f32_16x16x32_f16, two C tiles block-diagonal x FOUR tasks along K.
//
// Reference: the tensor-core block-MAC kernels multiplyV11..V14 (src/bmSparse_SPGEMM.cu:294-733); V14 packs two C blocks
// block-diagonally and two tasks along K into one 16x16x16 WMMA (:362-386).  The K = 32 MFMA of gfx950 takes the same
// block-diagonal pair with four tasks along K: eight 8x8x8 block products per instruction.
//
//   operand lanes (verified on hardware by mfma32_selftest below): lane l supplies A[row l&15][k = 8*(l>>4) .. +7] and
//   B[k = 8*(l>>4) .. +7][col l&15]; the result lane l holds D[row 4*(l>>4) + i][col l&15], i = 0..3.
//   => lane (s = l>>4, tile = (l>>3)&1, r = l&7) feeds LINE r (row of the A tile, column of the column-major B tile: eight
//      consecutive tile positions = 16 bytes) of ONE task: task 4*step + s of C tile `tile` of the pair.
//
// Structure (measured motivation in DESIGN.md, block-MAC log): the r1 kernel decoded bitmaps in the MFMA lanes -- with ~1.2 tasks per
// C block most K slots are empty and their decode is wasted (56 VALU wave-instructions per task) -- and strode C blocks across
// waves, so a B tile line shared by neighbouring C blocks was fetched by several CUs / XCDs (3.6x the compulsory HBM traffic).
// Here
//   * a wave owns a CONTIGUOUS range of C tiles (equal task quota per wave) and walks it in windows of whole C tiles holding
//     <= W tasks; a hub C tile with more tasks is walked in W-task slices with the accumulator carried in registers;
//   * operands are fetched TASK-parallel, 8 lanes per task, every lane one 16-byte tile line: the A line from a per-matrix
//     dense-expanded fp16 copy of the tiles (128 B per block, built once like block_meta: one 16-byte load, no decode -- an A
//     tile is reused by every task of its block-row, so the copy is served from L1/L2), the B line either from B's dense
//     copy (matrices with mostly full tiles: no bitmap, no gather of the block record) or decoded from the compact values
//     (two nibble requests of <= 4 consecutive halves, `load_nibble_wide`); the 16-byte lines are parked in LDS, one
//     256-byte slot per task, so the decode runs with every lane group busy whatever the tasks-per-C-tile ratio is;
//   * the MFMA lanes then read their task's lines with two ds_read_b128 (dead K slots read a zero slot);
//   * finished C tiles are transposed through the slot of their own first task, compacted IN PLACE by the C bitmap
//     (position per lane, rank = mbcnt of the wave-uniform bitmap) into one contiguous run of the window's C values and
//     leave with coalesced stores.
//   The next window's task words and C-tile words are requested while the current window computes.
#include "mac_common.hip.h"

namespace bmsp {
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// ---- dense-expanded tiles (cached per matrix) -------------------------------------------------------------------------
template <typename T>
struct ExpandDense {
    const uint64_t *bmps, *offsets;
    const T *values;
    T *out;
    __device__ void operator()(uint64_t i) const
    {
        const uint64_t b = i >> 6;
        const int p = (int)(i & 63u);
        const uint64_t bm = bmps[b];
        out[i] = tile_has(bm, p) ? values[offsets[b] + (uint64_t)tile_rank(bm, p)] : (T)0;
    }
};

// ---- the kernel -------------------------------------------------------------------------------------------------------
constexpr int kW = 32;  // tasks per window

struct Mac32Args {
    const uint64_t *tasks;
    uint32_t n_tasks;
    const uint32_t *task_begin, *c_of_wave;
    const _Float16 *a_dense;
    uint32_t a_dense_bytes;
    const uint32_t *b_meta;
    uint32_t b_meta_bytes;
    const _Float16 *b_vals;
    uint32_t b_bytes;
    const _Float16 *b_dense;
    uint32_t b_dense_bytes;
    const uint64_t *c_bmps, *c_offs;
    float *c_vals;
    uint32_t c_size, quota;
};

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// 4 workgroups per CU (LDS: 4 x 38 KB): the register allocation is held to 128 so that the fourth one fits
template <int W, bool B_DENSE, int V>
__global__ __launch_bounds__(kThreads, 4) void block_mac_mfma32_kernel(Mac32Args g)
{
    static_assert(W == 32, "the head-flag ballots below are 32 bits wide");
    // per wave: W records (16 B: task records while staging, C-tile records while storing) | W + 1 slots of 256 B (A lines 0-7,
    // B lines 0-7; slot W stays zero) | the MFMA schedule: W steps x 8 slot bytes, W step words, W head flags
    constexpr int kUnits = 17 * W + 16 + (16 * W + 4 * W + 4 * W) / 16;
    __shared__ u32x4_t lds_all[4][kUnits];
    __shared__ uint64_t s_sel[16];
    const int w = wave_id(), lane = lane_id();
    if (threadIdx.x < 16) s_sel[threadIdx.x] = nibble_selector(threadIdx.x);
    u32x4_t *rec = lds_all[w];
    u32x4_t *tiles = rec + W;
    float *tf = (float *)tiles;
    u32x4_t *sched = tiles + 16 * (W + 1);                   // [W] per step: 8 x uint16 = byte offset of the slot feeding (tile of the pair, K slot); slot W = dead
    uint16_t *sched16 = (uint16_t *)sched;
    uint32_t *dinfo = (uint32_t *)(sched + W);               // [W] bit 0: first step of a pair, bit 1: last; bytes 1, 2: D slots
    uint32_t *hflag = dinfo + W;                             // [W] 1 = a C tile begins at this task of the window
    if (lane < 16) tiles[16 * W + lane] = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();

    const int r = lane & 7, grp = lane >> 3;      // staging / storing: lane group `grp` = one task / one C tile, line r
    const int ks = lane >> 4, half_sel = grp & 1;  // MFMA: K slot, which C tile of the pair
    const NibbleLane nl0 = make_nibble_lane(r, 0), nl1 = make_nibble_lane(r, 1);
    const rsrc_t rda = make_rsrc(g.a_dense, g.a_dense_bytes);
    const rsrc_t rdb = make_rsrc(B_DENSE ? (const void *)g.b_dense : (const void *)g.b_vals, B_DENSE ? g.b_dense_bytes : g.b_bytes);
    const rsrc_t rmb = make_rsrc(g.b_meta, g.b_meta_bytes);

    // XCD-aware order: workgroups b and b + 8 share an XCD (round-robin dispatch), so the workgroups of one XCD take a contiguous
    // eighth of the C tiles and neighbouring block-rows -- which read the same B block-rows -- meet in the same L2
    uint32_t wg;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        wg = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    // the wave's C-tile range: tiles whose first task falls into [wv * quota, (wv + 1) * quota)
    const uint32_t wv = wg * 4 + w;
    uint32_t rs = g.c_size;
    if (lane < 2) {
        const uint64_t t = (uint64_t)(wv + lane) * g.quota;
        if (t < g.n_tasks) {
            const uint32_t c = g.c_of_wave[t >> 6];
            rs = g.task_begin[c] == (uint32_t)t ? c : c + 1;
        }
    }
    uint32_t c = rl(rs, 0);
    const uint32_t ce = rl(rs, 1);
    if (c >= ce) return;

    // lane l: words of C tile c + l
    uint32_t tbv = g.task_begin[min(c + (uint32_t)lane, g.c_size)];
    uint64_t cbmp = g.c_bmps[min(c + (uint32_t)lane, g.c_size - 1)], coff = g.c_offs[min(c + (uint32_t)lane, g.c_size)];
    uint64_t tk = 0;
    {
        const uint32_t t0 = rl(tbv, 0);
        if (lane < W && t0 + lane < g.n_tasks) tk = g.tasks[t0 + lane];
    }
    uint32_t hub_lo = 0;
    bool in_hub = false;
    float4_t acc = {0.f, 0.f, 0.f, 0.f};  // lives across windows: a hub tile's slices accumulate into it

    while (c < ce) {
        // ---- window: whole C tiles with <= W tasks, or a W-task slice of one hub tile ----
        const uint32_t t0 = rl(tbv, 0);
        const uint32_t lim = min((uint32_t)W, ce - c);
        const bool fits = lane >= 1 && (uint32_t)lane <= lim && tbv - t0 <= (uint32_t)W;
        const uint32_t n = (uint32_t)__popcll(__ballot(fits));
        uint32_t w_lo, w_n, ntiles, c_next;
        bool first = true, last = true;
        if (n > 0) {
            w_lo = t0; w_n = rl(tbv, n) - t0; ntiles = n; c_next = c + n;
        } else {
            const uint32_t t1 = rl(tbv, 1);
            first = !in_hub;
            if (first) hub_lo = t0;
            w_lo = hub_lo; w_n = min((uint32_t)W, t1 - hub_lo); ntiles = 1;
            last = w_lo + w_n == t1;
            in_hub = !last;
            c_next = last ? c + 1 : c;
        }
        // ---- requests for the next window (consumed at the bottom of the loop) ----
        uint32_t tbv_n = tbv;
        uint64_t cbmp_n = cbmp, coff_n = coff, tk_n = 0;
        if (c_next != c && c_next < ce) {
            tbv_n = g.task_begin[min(c_next + (uint32_t)lane, g.c_size)];
            cbmp_n = g.c_bmps[min(c_next + (uint32_t)lane, g.c_size - 1)];
            coff_n = g.c_offs[min(c_next + (uint32_t)lane, g.c_size)];
        }
        {
            const uint32_t nt = w_lo + w_n;
            if (lane < W && nt + lane < g.n_tasks) tk_n = g.tasks[nt + lane];
        }
        // ---- task records: lane-per-task gather of B's block record (compact B only); the request is issued here, the record
        //      is written once the schedule knows the task's K slot ----
        u32x4_t rc_me = {0u, 0u, 0u, 0u};
        if ((uint32_t)lane < w_n) {
            const uint32_t a = (uint32_t)(tk >> 32), b = (uint32_t)tk;
            if (B_DENSE) {
                rc_me[0] = b; rc_me[3] = a;
            } else {
                const u32x4_t m = __builtin_amdgcn_raw_buffer_load_b128(rmb, b << 4, 0, 0);
                rc_me[0] = m[0]; rc_me[1] = m[1]; rc_me[2] = m[2] * 2u; rc_me[3] = a;
            }
        }
        if (lane < W) {
            sched[lane] = u32x4_t{0x00010001u * (256u * W), 0x00010001u * (256u * W), 0x00010001u * (256u * W), 0x00010001u * (256u * W)};  // every K slot dead
            dinfo[lane] = 0u;
            hflag[lane] = 0u;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the MFMA schedule of the window, lane-parallel.  Tile lanes: steps of their pair = max of the two task counts / 4,
        //      exclusive sum over pairs = first step of the pair.  Task lanes: tile and rank inside it from the head-flag ballot;
        //      the task's slot number goes into byte (tile & 1) * 4 + rank % 4 of step first(pair) + rank / 4. ----
        uint32_t n_steps;
        {
            const uint32_t tb_next = (uint32_t)__shfl_down((int)tbv, 1, kWave);
            const bool is_tile = (uint32_t)lane < ntiles;
            const uint32_t cnt = !is_tile ? 0u : (n > 0 ? tb_next - tbv : w_n);
            const uint32_t slot_me = n > 0 ? tbv - w_lo : 0u;  // slot of the tile's first task
            const uint32_t cnt_o = (uint32_t)__shfl_xor((int)cnt, 1, kWave), slot_o = (uint32_t)__shfl_xor((int)slot_me, 1, kWave);
            const uint32_t steps = (max(cnt, cnt_o) + 3) / 4;
            const uint32_t mine = (lane & 1) ? 0u : steps;
            const uint32_t incl = wave_inclusive_sum(mine);
            n_steps = rl(incl, 63);
            const uint32_t base = incl - mine;  // even lane: first step of my pair; odd lane: one past my pair's last step
            if (is_tile) hflag[slot_me] = 1u;
            if (is_tile && !(lane & 1)) {
                // the pair's last step carries the D slots (bit 1); a single-step pair also sets bit 0 (no accumulator needed).
                // A hub slice that is not the last keeps its sum in registers: no D word
                const uint32_t dsl = (slot_me << 8) | ((cnt_o ? slot_o : 0xffu) << 16);
                if (n > 0 || last) dinfo[base + steps - 1] = 2u | dsl | ((steps == 1 && n > 0) ? 1u : 0u);
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t heads = (uint32_t)__ballot((uint32_t)lane < w_n && hflag[lane & (W - 1)] != 0u);
            const uint32_t le = (heads & (0xffffffffu >> (31 - (lane & 31)))) | 1u;  // task 0 always begins a tile
            const uint32_t j = (uint32_t)__builtin_popcount(le) - 1u;
            const uint32_t idx = (uint32_t)(lane & 31) - (31u - (uint32_t)__builtin_clz(le));
            const uint32_t pbase = (uint32_t)__shfl((int)base, (int)(j & ~1u), kWave);  // every lane takes part in the shuffle
            if ((uint32_t)lane < w_n) sched16[(pbase + (idx >> 2)) * 8u + (j & 1u) * 4u + (idx & 3u)] = (uint16_t)(lane * 256);
            // bank swizzle: tasks in odd K slots keep their A lines in the upper half of the slot (B lines in the lower), so the
            // 16 lanes a ds_read_b128 serves together -- lines 0-3 / 4-7 of K slots k and k+1 -- fall on 16 different 16-byte banks
            if (!(V & 2)) rc_me[3] |= (idx & 1u) << 31;
            if (lane < W) rec[lane] = rc_me;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- staging: 8 lanes per task, one 16-byte line each; all requests of the window first, then the decode ----
        if (!(V & 16)) {
            constexpr int U = W / 8;
            u32x4_t av[U];
            u32x4_t bd[U];        // dense B line
            u32x3_t bq0[U], bq1[U];  // compact B: the two nibble windows
            uint32_t nib0[U], nib1[U], ad0[U], ad1[U], swz[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t t = (uint32_t)(8 * u + grp);
                const u32x4_t rc = rec[t];
                swz[u] = (rc[3] >> 31) * 8u;
                av[u] = __builtin_amdgcn_raw_buffer_load_b128(rda, (rc[3] << 7) + (uint32_t)(r * 16), 0, 0);  // dead tasks: record 0, unused (bit 31 shifts out)
                if (B_DENSE) {
                    bd[u] = __builtin_amdgcn_raw_buffer_load_b128(rdb, (rc[0] << 7) + (uint32_t)(r * 16), 0, 0);
                } else {
                    const uint32_t lo = rc[0], hi = rc[1];
                    nib0[u] = ((nl0.use_hi ? hi : lo) >> nl0.shift) & 0xfu;  // dead tasks carry an all-zero record: no nibble, no request
                    nib1[u] = ((nl1.use_hi ? hi : lo) >> nl1.shift) & 0xfu;
                    ad0[u] = rc[2] + 2u * ((uint32_t)__builtin_popcount(hi & nl0.hi_mask) + (uint32_t)__builtin_popcount(lo & nl0.lo_mask));
                    ad1[u] = ad0[u] + 2u * (uint32_t)__builtin_popcount(nib0[u]);
                    bq0[u] = __builtin_amdgcn_raw_buffer_load_b96(rdb, nib0[u] ? (ad0[u] & ~3u) : kOob, 0, 0);
                    bq1[u] = __builtin_amdgcn_raw_buffer_load_b96(rdb, nib1[u] ? (ad1[u] & ~3u) : kOob, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t t = (uint32_t)(8 * u + grp);
                u32x4_t bl;
                if (B_DENSE) {
                    bl = bd[u];
                } else {
                    const uint64_t s0 = s_sel[nib0[u]], s1 = s_sel[nib1[u]];
                    const uint32_t sh0 = (ad0[u] & 2u) * 8u, sh1 = (ad1[u] & 2u) * 8u;
                    const uint32_t a01 = __builtin_amdgcn_alignbit(bq0[u][1], bq0[u][0], sh0), a23 = __builtin_amdgcn_alignbit(bq0[u][2], bq0[u][1], sh0);
                    const uint32_t b01 = __builtin_amdgcn_alignbit(bq1[u][1], bq1[u][0], sh1), b23 = __builtin_amdgcn_alignbit(bq1[u][2], bq1[u][1], sh1);
                    bl[0] = __builtin_amdgcn_perm(a23, a01, (uint32_t)s0);
                    bl[1] = __builtin_amdgcn_perm(a23, a01, (uint32_t)(s0 >> 32));
                    bl[2] = __builtin_amdgcn_perm(b23, b01, (uint32_t)s1);
                    bl[3] = __builtin_amdgcn_perm(b23, b01, (uint32_t)(s1 >> 32));
                }
                if (t < w_n) {
                    tiles[16 * t + (r ^ swz[u])] = av[u];
                    tiles[16 * t + (r ^ swz[u] ^ 8u)] = bl;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- C-tile records for the store phase (the task records are dead now): {bitmap, first value inside the window, slot} ----
        const uint32_t off0 = rl((uint32_t)coff, 0);
        if ((uint32_t)lane < ntiles)
            rec[lane] = u32x4_t{(uint32_t)cbmp, (uint32_t)(cbmp >> 32), (uint32_t)coff - off0, n > 0 ? tbv - w_lo : 0u};
        // ---- MFMA steps ----
        {
            const char *tbytes = (const char *)tiles;
            const uint16_t *sp = sched16 + (half_sel * 4 + ks);
            const uint32_t lofs = (uint32_t)(r * 16) ^ ((V & 2) ? 0u : (uint32_t)((ks & 1) * 128));  // A line of this lane's K slot (B line: ^ 128)
            const int dt = lane >> 5;
            const bool d_lane = dt == ((lane >> 3) & 1);
            float *dbase = tf + 32 * ((lane >> 4) & 1) + (lane & 7);
            for (uint32_t q = 0; q < ((V & 8) ? 0u : n_steps); q++, sp += 8) {
                const uint32_t di = (uint32_t)__builtin_amdgcn_readfirstlane((int)dinfo[q]);
                const char *src = tbytes + (uint32_t)*sp;
                const half8_t fa = __builtin_bit_cast(half8_t, *(const u32x4_t *)(src + lofs));
                const half8_t fb = __builtin_bit_cast(half8_t, *(const u32x4_t *)(src + (lofs ^ 128u)));
                // D[4*(lane>>4) + i][lane & 15]: tile 0 = rows/cols 0-7, tile 1 = rows/cols 8-15.  Row-major into the slot of the
                // tile's own first task (its operand lines are consumed): float index row * 8 + col
                const uint32_t dslot = (di >> (dt ? 16 : 8)) & 0xffu;
                if (!(V & 1) && (di & 3u) == 3u) {  // the whole pair in one step: no accumulator
                    const float4_t d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, float4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    if (d_lane && dslot != 0xffu) {
                        float *d = dbase + 64 * dslot;
                        d[0] = d4[0]; d[8] = d4[1]; d[16] = d4[2]; d[24] = d4[3];
                    }
                } else {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
                    if (di & 2u) {
                        if (d_lane && dslot != 0xffu) {
                            float *d = dbase + 64 * dslot;
                            d[0] = acc[0]; d[8] = acc[1]; d[16] = acc[2]; d[24] = acc[3];
                        }
                        acc = float4_t{0.f, 0.f, 0.f, 0.f};  // the next pair (or hub tile) starts from zero
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- compaction by the C bitmaps: 8 lanes per C tile, lane = tile row; rows are read whole before any compacted value
        //      is written (the compacted run [0, total) grows over the D tiles it has consumed), then one coalesced store ----
        if (last && !(V & 4)) {
            for (uint32_t q = 0; q < ntiles; q += 8) {
                const uint32_t j = q + (uint32_t)grp;
                const bool on = j < ntiles;
                const u32x4_t rc = rec[on ? j : 0u];
                const uint32_t lo = rc[0], hi = rc[1];
                const uint32_t byte = on ? ((nl0.use_hi ? hi : lo) >> (nl0.shift - 4u)) & 0xffu : 0u;
                uint32_t dst = rc[2] + (uint32_t)__builtin_popcount(hi & nl0.hi_mask) + (uint32_t)__builtin_popcount(lo & nl0.lo_mask);
                const float4_t v0 = *(const float4_t *)(tf + 64 * rc[3] + 8 * r), v1 = *(const float4_t *)(tf + 64 * rc[3] + 8 * r + 4);
                __builtin_amdgcn_wave_barrier();
                const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int cc = 0; cc < 8; cc++) {
                    if (byte & (0x80u >> cc)) { tf[dst] = vv[cc]; dst++; }
                }
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t total = rl((uint32_t)coff, ntiles) - off0;
            float *dst = g.c_vals + (((uint64_t)rl((uint32_t)(coff >> 32), 0) << 32) | off0);
            for (uint32_t i = (uint32_t)lane; i < total; i += 64) dst[i] = tf[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (!last) hub_lo += w_n;
        c = c_next;
        tbv = tbv_n; cbmp = cbmp_n; coff = coff_n; tk = tk_n;
    }
}

// ---- the direct kernel: operands straight from the dense copies into the MFMA lanes, no LDS -------------------------------------------
// For products with many tasks per C tile (FEM-like: 14) the staging above is overhead: the MFMA lane (K slot ks = l >> 4, tile of the
// pair (l >> 3) & 1, line r = l & 7) wants exactly ONE 16-byte line of A's dense copy and ONE of B's -- the lines 8 lanes of a
// task-parallel fetch would park in LDS.  Here every lane loads its own two lines (8 lanes = one 128-byte tile) of task 4 * step + ks of
// its C tile and feeds them to the MFMA; a pair of consecutive C tiles takes max(n0, n1) / 4 steps.  The pipeline is three deep over the
// flattened (pair, step) sequence: task word of step g + 2, lines of step g + 1, MFMA of step g.  Finished pairs are stored from the
// accumulator lanes: lane (tile dt = l >> 5, rows 4 * ((l >> 4) & 1) + i, column l & 7) holds four elements of one column; each goes to
// rank(C bitmap, position) of its tile.
//   Measured (T_7, us): FEM-like 890 -> 580, full-tile banded 73 -> 55; R-MAT 2^16 (4.5 tasks per C tile, skewed) 2590 -> 3110 and
//   cage-like (1.2) 428 -> 652: pairs of tiles with unequal, small task counts leave most operand slots empty, so the launcher takes this
//   kernel at >= 4.6 tasks per C tile only.  Running the two diagonal blocks as independent tile streams (no max(n0, n1)) was slower on
//   every case (FEM-like 666, banded 69): the streams' tiles are half a quota apart and stop sharing A / B lines in the L1.
// Round 3: the pipeline moves BLOCKS of U steps (a lane's task words of U steps in one request group, then its 2 U operand lines, then
// U MFMAs), because a wave's loads return in order and the wait for the youngest request drains the older ones: with one step per
// stage every step cost a full memory round trip (profiles/r02d: 73 % of the wave cycles parked in s_waitcnt).  The C bitmap and value
// offset of the lane's tile are requested when the pair is ENTERED (two blocks ahead), not when it completes -- the store used to stall
// on them with every newer request in flight.
typedef uint32_t u32x2v_t __attribute__((ext_vector_type(2)));
struct DirectPos {
    uint32_t c;       // first tile of the pair
    uint32_t s, steps;  // first step of the block; steps of the pair
    uint32_t tb, n;   // per lane: first task and task count of the lane's tile of the pair
    uint64_t cb, co;  // per lane: C bitmap and value offset of the tile this lane stores (accumulator lanes only)
};
template <int U>
struct DirectTasks {
    uint64_t t[U];
};
template <int U>
struct DirectLines {
    half8_t a[U], b[U];
};

template <int U>
__global__ __launch_bounds__(kThreads) void block_mac_direct_kernel(Mac32Args g)
{
    const int w = wave_id(), lane = lane_id();
    const int r = lane & 7, sel = (lane >> 3) & 1, ks = lane >> 4;
    const rsrc_t rda = make_rsrc(g.a_dense, g.a_dense_bytes), rdb = make_rsrc(g.b_dense, g.b_dense_bytes);
    const rsrc_t rtk = make_rsrc(g.tasks, g.n_tasks * 8u);
    uint32_t wg;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        wg = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    const uint32_t wv = wg * 4 + w;
    uint32_t rs = g.c_size;
    if (lane < 2) {
        const uint64_t t = (uint64_t)(wv + lane) * g.quota;
        if (t < g.n_tasks) {
            const uint32_t c = g.c_of_wave[t >> 6];
            rs = g.task_begin[c] == (uint32_t)t ? c : c + 1;
        }
    }
    const uint32_t c0 = rl(rs, 0), ce = rl(rs, 1);
    if (c0 >= ce) return;
    const int dt = lane >> 5;
    const bool d_lane = dt == sel;

    // a pair's words: task_begin[c .. c + 2] (wave-uniform scalar loads), the lane's tile = c + sel; C words of the tile it will store
    auto enter = [&](uint32_t c) {
        DirectPos p;
        p.c = c;
        p.s = 0;
        const uint32_t t0 = g.task_begin[c], t1 = g.task_begin[c + 1], t2 = c + 1 < ce ? g.task_begin[c + 2] : t1;
        const uint32_t n0 = t1 - t0, n1 = t2 - t1;
        p.steps = (max(n0, n1) + 3u) / 4u;
        p.tb = sel ? t1 : t0;
        p.n = sel ? n1 : n0;
        p.cb = 0; p.co = 0;
        if (d_lane && c + (uint32_t)dt < ce) { p.cb = g.c_bmps[c + (uint32_t)dt]; p.co = g.c_offs[c + (uint32_t)dt]; }
        return p;
    };
    auto advance = [&](const DirectPos &p) {
        DirectPos q = p;
        q.s = p.s + (uint32_t)U;
        if (q.s >= p.steps) {
            if (p.c + 2 < ce) q = enter(p.c + 2);
            else { q.c = ce; q.s = 0; q.steps = 0; q.tb = 0; q.n = 0; q.cb = 0; q.co = 0; }  // past the end: its loads are masked off
        }
        return q;
    };
    auto load_tasks = [&](const DirectPos &p) {
        DirectTasks<U> k;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t ti = 4u * (p.s + (uint32_t)u) + (uint32_t)ks;
            const u32x2v_t t = __builtin_amdgcn_raw_buffer_load_b64(rtk, (p.c < ce && ti < p.n) ? (p.tb + ti) * 8u : kOob, 0, 0);
            k.t[u] = ((uint64_t)t[1] << 32) | t[0];
        }
        return k;
    };
    auto load_lines = [&](const DirectPos &p, const DirectTasks<U> &k) {
        DirectLines<U> o;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool on = p.c < ce && 4u * (p.s + (uint32_t)u) + (uint32_t)ks < p.n;
            o.a[u] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rda, on ? ((uint32_t)(k.t[u] >> 32) << 7) + (uint32_t)(r * 16) : kOob, 0, 0));
            o.b[u] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rdb, on ? ((uint32_t)k.t[u] << 7) + (uint32_t)(r * 16) : kOob, 0, 0));
        }
        return o;
    };

    DirectPos pa = enter(c0);
    DirectTasks<U> tka = load_tasks(pa);
    DirectPos pb = advance(pa);
    DirectTasks<U> tkb = load_tasks(pb);
    DirectLines<U> oa = load_lines(pa, tka);
    float4_t acc = {0.f, 0.f, 0.f, 0.f};

    while (pa.c < ce) {
        // lines of the next block, task words of the one after; then this block's instructions
        const DirectLines<U> ob = load_lines(pb, tkb);
        const DirectPos pc = advance(pb);
        const DirectTasks<U> tkc = load_tasks(pc);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (pa.s + (uint32_t)u < pa.steps) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(oa.a[u], oa.b[u], acc, 0, 0, 0);
                if (pa.s + (uint32_t)u + 1 == pa.steps) {
                    // the pair is complete: compacted store by the C bitmap of the lane's tile
                    if (d_lane && pa.c + (uint32_t)dt < ce) {
                        const uint64_t cb = pa.cb, co = pa.co;
                        const uint32_t row0 = 4u * (uint32_t)((lane >> 4) & 1);
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t p = (row0 + (uint32_t)i) * 8u + (uint32_t)r;
                            if ((cb >> (63u - p)) & 1ull) g.c_vals[co + (uint64_t)__popcll(cb >> 1 >> (63u - p))] = acc[i];
                        }
                    }
                    acc = float4_t{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        pa = pb; pb = pc;
        tkb = tkc;
        oa = ob;
    }
}

// lane-layout self test of v_mfma_f32_16x16x32_f16: A[i][k] = i + 16 k (exact in fp16 up to 2048), B[k][j] = asymmetric small
// integers; the host checks D = A * B element by element (bmsp_selftest_mfma_layout)
__global__ void mfma32_selftest_kernel(float *d_out)
{
    const int lane = (int)threadIdx.x;
    half8_t a, b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = 8 * (lane >> 4) + j;
        a[j] = (_Float16)(float)((lane & 15) + ((k * 7) % 5) - 2);  // A[row = lane & 15][k]
        b[j] = (_Float16)(float)(((lane & 15) * 3 + k) % 7 - 3);     // B[k][col = lane & 15]
    }
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) d_out[(4 * (lane >> 4) + i) * 16 + (lane & 15)] = acc[i];  // D[row][col]
}

}  // namespace

void ensure_dense_tiles(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->dense_tiles) return;
    // fp16: 128 B per block (the MFMA kernels' operand lines and the V15 staging); fp32: 256 B per block (V15 staging only)
    if (m->dtype != BMSP_F16 && m->dtype != BMSP_F32) fail(BMSP_ERR_INVALID, "dense tile copies exist for fp16 and fp32 matrices");
    const size_t es = dtype_size(m->dtype);
    m->dense_tiles = pool_alloc(64 * es * (size_t)(m->block_num ? m->block_num : 1) + 64);
    if (!m->block_num) return;
    if (m->dtype == BMSP_F16)
        device_for_each(ExpandDense<_Float16>{m->bmps, m->offsets, (const _Float16 *)m->values, (_Float16 *)m->dense_tiles}, (uint64_t)m->block_num * 64, st);
    else
        device_for_each(ExpandDense<float>{m->bmps, m->offsets, (const float *)m->values, (float *)m->dense_tiles}, (uint64_t)m->block_num * 64, st);
}

// true when the kernel can run this product (32-bit byte offsets into the dense copies and the records)
bool mac_mfma32_supported(const bmsp_matrix_s *A, const bmsp_matrix_s *B)
{
    return A->dtype == BMSP_F16 && A->block_num < (1ll << 25) && B->block_num < (1ll << 25) && (uint64_t)B->values_extent() * 2 + 16 < (1ull << 32);
}

// which operand form B takes.  Measured (block-MAC log): the dense copy wins on every generator case -- the kernel is bound
// by instruction issue and LDS traffic, not by bytes, and the dense line needs no block record and no nibble decode (cage-like
// 421 vs 468 us, R-MAT 2^16 2.58 vs 2.91 ms, full tiles 72 vs 84 us) -- at 128 B per block of extra memory and ~2x the L2 miss
// traffic.  So: dense unless the copy would be large (> 4 GiB) and the tiles are sparse, or BMSP_MAC_B_DENSE says otherwise;
// always dense when the value array is borrowed without the read slack the 12-byte nibble loads need.
bool mac_mfma32_b_dense(const bmsp_matrix_s *B)
{
    const char *force = getenv("BMSP_MAC_B_DENSE");  // experiment / test switch (read per call)
    if (force) return force[0] == '1';
    if (!pool_owns(B->values)) return true;
    const bool full_tiles = B->block_num && (double)B->nnz / (double)B->block_num >= 32.0;
    return full_tiles || (uint64_t)B->block_num * 128ull <= (4ull << 30);
}

int launch_mac_mfma32(const uint64_t *tasks, uint64_t n_tasks, const uint32_t *task_begin, const uint32_t *c_of_wave, bmsp_matrix_s *A,
                      bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st)
{
    const uint32_t cs = (uint32_t)C->block_num;
    if (!cs) return BMSP_MAC_STAGED;
    ensure_dense_tiles(A, st);
    const bool b_dense = mac_mfma32_b_dense(B);
    if (b_dense) ensure_dense_tiles(B, st);
    else ensure_block_meta(B, st);
    Mac32Args g{};
    g.tasks = tasks; g.n_tasks = (uint32_t)n_tasks; g.task_begin = task_begin; g.c_of_wave = c_of_wave;
    g.a_dense = (const _Float16 *)A->dense_tiles; g.a_dense_bytes = (uint32_t)(A->block_num * 128);
    g.b_meta = B->block_meta; g.b_meta_bytes = (uint32_t)(B->block_num * 16);
    g.b_vals = (const _Float16 *)B->values; g.b_bytes = (uint32_t)(B->values_extent() * 2) + 16u;
    g.b_dense = (const _Float16 *)B->dense_tiles; g.b_dense_bytes = (uint32_t)(B->block_num * 128);
    g.c_bmps = C->bmps; g.c_offs = C->offsets; g.c_vals = (float *)C->values; g.c_size = cs;
    // equal task quotas per wave, a multiple of 64 (c_of_wave is indexed per 64 tasks); ~8 quotas per resident wave slot so
    // that hub tiles and uneven tiles-per-task do not leave a tail
    // >= 4.6 tasks per C tile: the direct kernel (no LDS); sparser task lists keep the staged one (BMSP_MAC_DIRECT = 0 / 1 forces)
    const char *de = getenv("BMSP_MAC_DIRECT");
    const bool direct = b_dense && (de ? de[0] == '1' : 10 * n_tasks >= 46 * (uint64_t)cs);
    const char *qenv = getenv("BMSP_MAC_QUOTA");
    // (measured, T_7 in us at 256 / 512 / 1024 / 2048 / 4096 tasks per wave: R-MAT 2^16 x 8 2586 / 2496 / 2511 / 2548 / 2699, cage-like 427 (128) /
    // 421 (512) / 454 (2048): short quotas keep an XCD's resident waves on few block-rows of B at a time -- capped at 512 while that leaves
    // at most 2^20 waves)
    uint64_t quota = qenv ? (uint64_t)atoll(qenv) : std::min<uint64_t>((n_tasks + 32767) / 32768, std::max<uint64_t>(512, n_tasks >> 20));
    quota = std::max<uint64_t>(256, (quota + 63) / 64 * 64);
    // the direct kernel reads every operand line from the L2 or beyond: short quotas keep the resident waves of an XCD on few block-rows
    // at a time, whose B tiles then meet in its L2 (FEM-like T_7: 584 us at 1024 tasks per wave, 574 at 768, 530 at 256, 526 at 128;
    // dense ceiling 1590 -> 1429 us)
    if (direct && !qenv) quota = 256;
    const uint64_t waves = (n_tasks + quota - 1) / quota;
    const uint32_t grid = (uint32_t)((waves + 3) / 4);
    g.quota = (uint32_t)quota;
    if (const char *ab = getenv("BMSP_MAC_ABLATE")) {  // timing-only builds: a zero-size descriptor drops every load through it
        const int m = atoi(ab);
        if (m & 1) g.a_dense_bytes = 0;
        if (m & 2) { g.b_bytes = 0; g.b_dense_bytes = 0; }
        if (m & 4) g.b_meta_bytes = 0;
    }
    const char *venv = getenv("BMSP_MAC_VARIANT");  // timing experiments: 4 = no store phase, 8 = no MFMA loop, 16 = no staging (wrong results)
    const int v = venv ? atoi(venv) : 0;
    const char *ue = getenv("BMSP_MAC_DIRECT_U");  // experiment switch: steps per pipeline block (1, 2, 4)
    const int du = ue ? atoi(ue) : 1;  // measured on MI355X (FEM-like / dense ceiling T_7, us): U = 1: 574 / 1631, 2: 640 / 1590, 4: 696 / 1596 -- the kernel is bound by the fabric
                                         // rate of its 128-byte line gathers (~11 TB/s), not by round trips, so deeper blocks only cost occupancy
    if (direct && du == 4) hipLaunchKernelGGL(block_mac_direct_kernel<4>, dim3(grid), dim3(kThreads), 0, st, g);
    else if (direct && du == 2) hipLaunchKernelGGL(block_mac_direct_kernel<2>, dim3(grid), dim3(kThreads), 0, st, g);
    else if (direct) hipLaunchKernelGGL(block_mac_direct_kernel<1>, dim3(grid), dim3(kThreads), 0, st, g);
    else if (b_dense) hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, true, 0>), dim3(grid), dim3(kThreads), 0, st, g);
    else if (v == 4) hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, false, 4>), dim3(grid), dim3(kThreads), 0, st, g);
    else if (v == 8) hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, false, 8>), dim3(grid), dim3(kThreads), 0, st, g);
    else if (v == 16) hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, false, 16>), dim3(grid), dim3(kThreads), 0, st, g);
    else if (v == 28) hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, false, 28>), dim3(grid), dim3(kThreads), 0, st, g);
    else hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, false, 0>), dim3(grid), dim3(kThreads), 0, st, g);
    BMSP_CHECK_LAUNCH();
    return direct ? BMSP_MAC_DIRECT : BMSP_MAC_STAGED;
}

// runs the lane-layout self test; returns the number of mismatching elements of the 16x16 result
int mfma32_selftest(hipStream_t st)
{
    DevBuf<float> d(256);
    hipLaunchKernelGGL(mfma32_selftest_kernel, dim3(1), dim3(64), 0, st, d.p);
    BMSP_CHECK_LAUNCH();
    float h[256];
    BMSP_HIP(hipMemcpyAsync(h, d.p, sizeof h, hipMemcpyDeviceToHost, st));
    BMSP_HIP(hipStreamSynchronize(st));
    int bad = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            float ref = 0.f;
            for (int k = 0; k < 32; k++) ref += (float)(i + ((k * 7) % 5) - 2) * (float)((j * 3 + k) % 7 - 3);
            if (h[i * 16 + j] != ref) bad++;
        }
    return bad;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(blockmac32)
