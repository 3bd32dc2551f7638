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
struct ExpandDense {
    const uint64_t *bmps, *offsets;
    const _Float16 *values;
    _Float16 *out;
    __device__ void operator()(uint64_t i) const
    {
        const uint64_t b = i >> 6;
        const int p = (int)(i & 63u);
        const uint64_t bm = bmps[b];
        out[i] = tile_has(bm, p) ? values[offsets[b] + (uint64_t)tile_rank(bm, p)] : (_Float16)0;
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

template <int W, bool B_DENSE>
__global__ __launch_bounds__(kThreads) void block_mac_mfma32_kernel(Mac32Args g)
{
    // per wave: W task records (16 B) | W + 1 slots of 256 B (A lines 0-7, B lines 0-7); slot W stays zero
    constexpr int kUnits = 17 * W + 16;
    __shared__ u32x4_t lds_all[4][kUnits];
    __shared__ uint64_t s_sel[16];
    const int w = wave_id(), lane = lane_id();
    if (threadIdx.x < 16) s_sel[threadIdx.x] = nibble_selector(threadIdx.x);
    u32x4_t *rec = lds_all[w];
    u32x4_t *tiles = rec + W;
    float *tf = (float *)tiles;
    if (lane < 16) tiles[16 * W + lane] = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();

    const int r = lane & 7, grp = lane >> 3;      // staging: lane group `grp` = one task, line r
    const int ks = lane >> 4, half_sel = grp & 1;  // MFMA: K slot, which C tile of the pair
    const NibbleLane nl0 = make_nibble_lane(r, 0), nl1 = make_nibble_lane(r, 1);
    const rsrc_t rda = make_rsrc(g.a_dense, g.a_dense_bytes);
    const rsrc_t rdb = make_rsrc(B_DENSE ? (const void *)g.b_dense : (const void *)g.b_vals, B_DENSE ? g.b_dense_bytes : g.b_bytes);
    const rsrc_t rmb = make_rsrc(g.b_meta, g.b_meta_bytes);

    // the wave's C-tile range: tiles whose first task falls into [wv * quota, (wv + 1) * quota)
    const uint32_t wv = blockIdx.x * 4 + w;
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
    float4_t hub_acc = {0.f, 0.f, 0.f, 0.f};

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
        // ---- task records: lane-per-task gather of B's block record (compact B only) ----
        {
            u32x4_t rc = {0u, 0u, 0u, 0u};
            if ((uint32_t)lane < w_n) {
                const uint32_t a = (uint32_t)(tk >> 32), b = (uint32_t)tk;
                if (B_DENSE) {
                    rc[0] = b; rc[3] = a;
                } else {
                    const u32x4_t m = __builtin_amdgcn_raw_buffer_load_b128(rmb, b << 4, 0, 0);
                    rc[0] = m[0]; rc[1] = m[1]; rc[2] = m[2] * 2u; rc[3] = a;
                }
            }
            if (lane < W) rec[lane] = rc;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- staging: 8 lanes per task, one 16-byte line each; all requests of the window first, then the decode ----
        {
            constexpr int U = W / 8;
            u32x4_t av[U];
            u32x4_t bd[U];        // dense B line
            u32x3_t bq0[U], bq1[U];  // compact B: the two nibble windows
            uint32_t nib0[U], nib1[U], ad0[U], ad1[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t t = (uint32_t)(8 * u + grp);
                const bool live = t < w_n;
                const u32x4_t rc = rec[t];
                av[u] = __builtin_amdgcn_raw_buffer_load_b128(rda, live ? (rc[3] << 7) + (uint32_t)(r * 16) : kOob, 0, 0);
                if (B_DENSE) {
                    bd[u] = __builtin_amdgcn_raw_buffer_load_b128(rdb, live ? (rc[0] << 7) + (uint32_t)(r * 16) : kOob, 0, 0);
                } else {
                    const uint32_t lo = rc[0], hi = rc[1];
                    nib0[u] = live ? ((nl0.use_hi ? hi : lo) >> nl0.shift) & 0xfu : 0u;
                    nib1[u] = live ? ((nl1.use_hi ? hi : lo) >> nl1.shift) & 0xfu : 0u;
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
                    tiles[16 * t + r] = av[u];
                    tiles[16 * t + 8 + r] = bl;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- MFMA: pairs of C tiles, four tasks of each per instruction ----
        const uint32_t pairs = (ntiles + 1) / 2;
        for (uint32_t p = 0; p < pairs; p++) {
            const uint32_t j0 = 2 * p, j1 = 2 * p + 1;
            uint32_t b0, e0, e1;
            if (n > 0) {
                b0 = rl(tbv, j0); e0 = rl(tbv, j0 + 1);
                e1 = j1 < ntiles ? rl(tbv, j1 + 1) : e0;
            } else {
                b0 = w_lo; e0 = w_lo + w_n; e1 = e0;
            }
            const uint32_t n0 = e0 - b0, n1 = e1 - e0;
            const uint32_t steps = (max(n0, n1) + 3) / 4;
            const uint32_t my_n = half_sel ? n1 : n0, my_b = (half_sel ? e0 : b0) - w_lo;
            float4_t acc = (n == 0 && !first) ? hub_acc : float4_t{0.f, 0.f, 0.f, 0.f};
            for (uint32_t s = 0; s < steps; s++) {
                const uint32_t idx = 4 * s + (uint32_t)ks;
                const uint32_t slot = idx < my_n ? my_b + idx : (uint32_t)W;
                const half8_t fa = __builtin_bit_cast(half8_t, tiles[16 * slot + r]);
                const half8_t fb = __builtin_bit_cast(half8_t, tiles[16 * slot + 8 + r]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
            }
            if (n == 0 && !last) {
                hub_acc = acc;
            } else {
                // D[4*(lane>>4) + i][lane & 15]: tile 0 = rows/cols 0-7, tile 1 = rows/cols 8-15.  Column-major into the slot
                // of the tile's own first task: float index (col & 7) * 8 + row
                const int dt = lane >> 5;  // which tile this lane's rows belong to
                const bool useful = dt == ((lane >> 3) & 1) && (dt == 0 || j1 < ntiles);
                if (useful) {
                    const uint32_t slot = (dt ? e0 : b0) - w_lo;
                    tiles[16 * slot + 2 * (lane & 7) + ((lane >> 4) & 1)] = __builtin_bit_cast(u32x4_t, acc);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- compaction by the C bitmaps, in place, and the coalesced store of the window's values ----
        if (last) {
            const uint32_t off0 = rl((uint32_t)coff, 0);
            for (uint32_t j = 0; j < ntiles; j++) {
                const uint32_t blo = rl((uint32_t)cbmp, j), bhi = rl((uint32_t)(cbmp >> 32), j);
                const uint32_t slot = n > 0 ? rl(tbv, j) - w_lo : 0u;
                const uint32_t o = rl((uint32_t)coff, j) - off0;
                // position `lane` is bit 63-lane of the bitmap = bit `lane` of its reversal
                const uint32_t rlo = __builtin_bitreverse32(bhi), rhi = __builtin_bitreverse32(blo);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(rhi, __builtin_amdgcn_mbcnt_lo(rlo, 0u));
                const float v = tf[64 * slot + (uint32_t)((lane & 7) * 8 + (lane >> 3))];
                __builtin_amdgcn_wave_barrier();
                if (((lane < 32 ? rlo : rhi) >> (lane & 31)) & 1u) tf[o + rank] = v;
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
    if (m->dtype != BMSP_F16) fail(BMSP_ERR_INVALID, "dense fp16 tiles need an fp16 matrix");
    m->dense_tiles = pool_alloc(128 * (size_t)(m->block_num ? m->block_num : 1) + 64);
    if (m->block_num)
        device_for_each(ExpandDense{m->bmps, m->offsets, (const _Float16 *)m->values, (_Float16 *)m->dense_tiles}, (uint64_t)m->block_num * 64, st);
}

// true when the kernel can run this product (32-bit byte offsets into the dense copies and the records)
bool mac_mfma32_supported(const bmsp_matrix_s *A, const bmsp_matrix_s *B)
{
    return A->dtype == BMSP_F16 && A->block_num < (1ll << 25) && B->block_num < (1ll << 25) && (uint64_t)B->values_extent() * 2 + 16 < (1ull << 32);
}

// which operand form B takes: dense copy when its tiles are mostly full (compact = 16-byte record + 2 B per value) or when its
// value array is borrowed without the read slack the 12-byte nibble loads need
bool mac_mfma32_b_dense(const bmsp_matrix_s *B)
{
    const char *force = getenv("BMSP_MAC_B_DENSE");  // experiment / test switch (read per call)
    if (force) return force[0] == '1';
    return !pool_owns(B->values) || (B->block_num && (double)B->nnz / (double)B->block_num >= 32.0);
}

void launch_mac_mfma32(const uint64_t *tasks, uint64_t n_tasks, const uint32_t *task_begin, const uint32_t *c_of_wave, bmsp_matrix_s *A,
                       bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st)
{
    const uint32_t cs = (uint32_t)C->block_num;
    if (!cs) return;
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
    const char *qenv = getenv("BMSP_MAC_QUOTA");
    uint64_t quota = qenv ? (uint64_t)atoll(qenv) : (n_tasks + 32767) / 32768;
    quota = std::max<uint64_t>(256, (quota + 63) / 64 * 64);
    const uint64_t waves = (n_tasks + quota - 1) / quota;
    const uint32_t grid = (uint32_t)((waves + 3) / 4);
    g.quota = (uint32_t)quota;
    if (b_dense) hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, true>), dim3(grid), dim3(kThreads), 0, st, g);
    else hipLaunchKernelGGL((block_mac_mfma32_kernel<kW, false>), dim3(grid), dim3(kThreads), 0, st, g);
    BMSP_CHECK_LAUNCH();
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
