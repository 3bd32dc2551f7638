// spmv.hip -- u = A * v on the bmSparse format, gfx950 / wave64.
//
// Reference: bmSparse_SpMV<VI,VO>, src/bmSparse_SPMV.cu:191-230 with spmv_kernel (:153-189, one 64-thread block per
// block-row, one lane per tile ELEMENT, serial loop over the row's tiles) and spmv_kernel_new (:84-150, the "batched"
// path: several tiles per step, wide lane reduction).  Both rebuild a block-row pointer on every call (:199-206).
//
// MI355X design (variant 0, the default): the block-vector sweep over a cached plan.  Three kernels share the plan; the launcher picks by
// the matrix: spmv_vstream_kernel (lane per stored value; matrices of sparse tiles -- see its own header below), spmv_rowgroup_kernel
// (16 lanes per block-row, 16-byte value loads; >= 16 values per tile and no hub block-row), spmv_sweep_kernel (lane per tile, described
// here; since round 2 only its FULL-tile variant runs by default, BMSP_SPMV_OLD=1 brings the whole kernel back).
//   * A per-matrix SWEEP PLAN (built once, cached like the block-row pointer) cuts the block array into wave-sized
//     work items.  Short block-rows are grouped into items aligned to block-row boundaries (<= 16 block-rows inside
//     one aligned 16-row window, < 512 tiles; ~96 tiles for the value-stream kernel), so an item owns a contiguous slice of u outright.
//     A block-row with more than 256 tiles (hub rows of web / R-MAT graphs) is cut into 256-tile items of its own.
//   * One wave per item.  Lane l loads key / bitmap / offset of tiles l and l+64 of a 128-tile batch: three fully
//     coalesced streams, no dependent pointer chase.  Work per lane is one tile whatever the row lengths are, so
//     skewed graphs stay balanced.
//       sparse tiles (<= 8 stored values): the lane peels the first two elements off the top of the bitmap and issues
//         their value / x gathers back to back (buffer loads against wave-uniform descriptors; an absent element points
//         out of range and reads 0, so there is no branch and no 64-bit address arithmetic), then adds the products into
//         the item's u tile in LDS (16 block-rows x 8 rows per wave); tiles with more values wait in an LDS queue that
//         is drained 64 tiles at a time, so every lane of a drain has work.
//       dense tiles: queued in LDS and swept by the whole wave, lane p = tile position p (contiguous value loads),
//         8-lane DPP row sums, one LDS add per tile row.
//   * Short items store their u slice with coalesced stores (empty block-rows come out as zeros for free).
//     Long-row items park their 8 partial sums in a carry slot and bump a per-row arrival counter; the last wave to
//     arrive folds all slots in a fixed order and writes the row (write-through stores + agent-scope counter,
//     self-resetting), so the whole product is ONE launch.
//   Measured alternatives (DESIGN.md "SpMV design log"): ds_add_f32 costs ~3 cycles per ACTIVE lane on gfx950 while
//   integer LDS atomics run at full rate (experiments/lds_atomic_rate.hip); an atomic-free pull formulation and an
//   integer-atomic counting-sort exchange were both built and were slower (2-3x the instructions per tile, lower
//   occupancy).  This kernel needs ~2.8 wave instructions per tile, inside the budget of an HBM-bound sweep.
// variant 1 ("batched"): a whole wave sweeps one block-row, 8 tiles x 8 tile rows per step, and folds the eight partial
//     rows with xor-shuffles (wavefront reduction): contiguous value loads per tile, for matrices with dense tiles.
// variant 2: 8-lane group per block-row (kept for comparison).
#include "spmv_plan.h"
#include "prims.hip.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <vector>

namespace bmsp {
namespace {

constexpr uint32_t kVsGroup = 96;  // plan granularity for the value-stream kernel: most items are one 128-tile batch
// the value-stream kernel takes matrices whose tiles are mostly sparse (full tiles go to the FULL sweep / the row-group kernel)
bool vstream_eligible(const bmsp_matrix_s *A)
{
    return !(A->spmv_full_tiles * 4 >= A->block_num) && !getenv("BMSP_SPMV_OLD") && A->num_block_cols() < (1ll << 28);
}

// ---------------------------------------------------------------------------------------------------------
// plan construction (once per matrix)
// ---------------------------------------------------------------------------------------------------------
struct RowClass {
    const uint32_t *rowptr;
    const uint64_t *offsets;
    uint32_t nbr;
    uint32_t group;  // short block-rows are grouped up to the next multiple of `group` tiles
    __device__ bool is_long(uint32_t r) const { return rowptr[r + 1] - rowptr[r] > kItemTiles; }
    __device__ bool starts(uint32_t r) const
    {
        if (r == 0 || (r % kItemRows) == 0) return true;
        if (rowptr[r] / group != rowptr[r - 1] / group) return true;
        return is_long(r) || is_long(r - 1);
    }
    // packed counts: low 36 bits = items contributed by row r, high bits = 1 if long
    __device__ uint64_t operator()(uint64_t r64) const
    {
        if (r64 >= nbr) return 0;
        uint32_t r = (uint32_t)r64;
        if (is_long(r)) {
            uint32_t len = rowptr[r + 1] - rowptr[r];
            return (uint64_t)((len + kItemTiles - 1) / kItemTiles) | (1ull << 36);
        }
        return starts(r) ? 1ull : 0ull;
    }
};
struct PlanTotals {
    uint64_t n;
    uint64_t *out;
    __device__ void operator()(uint64_t i, uint64_t ex) const
    {
        if (i == n) *out = ex;
    }
};
struct FullTileIn {
    const uint64_t *bmps;
    uint64_t n;
    __device__ uint64_t operator()(uint64_t b) const { return (b < n && bmps[b] == ~0ull) ? 1ull : 0ull; }
};
struct PlanFill {
    RowClass rc;
    uint32_t nb;
    SweepItem *items;
    __device__ void operator()(uint64_t r64, uint64_t ex) const
    {
        const uint32_t base = (uint32_t)(ex & ((1ull << 36) - 1)), lbase = (uint32_t)(ex >> 36);
        const uint32_t r = (uint32_t)r64;
        if (r == rc.nbr) {
            if (rc.nbr && !rc.is_long(rc.nbr - 1)) { items[base - 1].row_end = rc.nbr; items[base - 1].blk_end = nb; items[base - 1].first_item = (uint32_t)rc.offsets[nb]; }
            return;
        }
        const bool lg = rc.is_long(r), st = lg || rc.starts(r);
        if (!st) return;
        const uint32_t s = rc.rowptr[r], e = rc.rowptr[r + 1];
        if (r > 0 && !rc.is_long(r - 1)) { items[base - 1].row_end = r; items[base - 1].blk_end = s; items[base - 1].first_item = (uint32_t)rc.offsets[s]; }
        if (lg) {
            const uint32_t cnt = (e - s + kItemTiles - 1) / kItemTiles;
            for (uint32_t c = 0; c < cnt; c++) {
                SweepItem it;
                it.row_begin = r; it.row_end = r + 1;
                it.blk_begin = s + c * kItemTiles;
                it.blk_end = min(e, s + (c + 1) * kItemTiles);
                it.first_item = base; it.num_items = cnt; it.long_idx = lbase; it.val_begin = (uint32_t)rc.offsets[it.blk_begin];
                items[base + c] = it;
            }
        } else {
            items[base].row_begin = r; items[base].blk_begin = s;
            items[base].num_items = 0; items[base].long_idx = 0xffffffffu; items[base].val_begin = (uint32_t)rc.offsets[s];
        }
    }
};

struct SweepPlan {
    SweepItem *items;
    uint32_t num_items, num_long;
    void *carry;         // num_items x 8 accumulators (only long items use their slot)
    uint32_t *counters;  // num_long arrival counters, zero between calls
};

}  // namespace

void build_plan(bmsp_matrix_s *A, hipStream_t st)
{
    if (A->spmv_chunks) return;
    ensure_rowptr(A, st);
    const uint32_t nbr = (uint32_t)A->num_block_rows(), nb = (uint32_t)A->block_num;
    HostScalar<uint64_t> full_h;
    device_exclusive_scan<uint64_t>(FullTileIn{A->bmps, nb}, PlanTotals{nb, full_h.dev()}, (uint64_t)nb + 1, st);
    A->spmv_full_tiles = (int64_t)full_h.wait(st);
    // short block-rows are grouped up to the next multiple of `group` tiles: one 128-tile batch for the value-stream kernel
    const char *ge = getenv("BMSP_SPMV_GROUP");
    RowClass rc{A->rowptr, A->offsets, nbr, ge ? (uint32_t)std::max(16, atoi(ge)) : (vstream_eligible(A) ? kVsGroup : kItemTiles)};
    DevBuf<uint64_t> tot(1);
    device_exclusive_scan<uint64_t>(rc, PlanTotals{nbr, tot.p}, (uint64_t)nbr + 1, st);
    const uint64_t packed = read_back(tot.p, st);
    const uint32_t n_items = (uint32_t)(packed & ((1ull << 36) - 1)), n_long = (uint32_t)(packed >> 36);
    // one allocation: header | items | counters | carry
    const size_t acc_sz = A->dtype == BMSP_F64 ? 8 : 4;
    const size_t off_items = 64, off_cnt = off_items + sizeof(SweepItem) * (size_t)n_items;
    const size_t off_carry = (off_cnt + 4 * (size_t)n_long + 63) & ~size_t(63);
    const size_t total = off_carry + acc_sz * 8 * (size_t)n_items + 64;
    char *mem = (char *)pool_alloc(total);
    BMSP_HIP(hipMemsetAsync(mem, 0, total, st));
    uint32_t hdr[4] = {n_items, n_long, (uint32_t)off_cnt, (uint32_t)off_carry};
    BMSP_HIP(hipMemcpyAsync(mem, hdr, sizeof hdr, hipMemcpyHostToDevice, st));
    if (nbr) device_exclusive_scan<uint64_t>(rc, PlanFill{rc, nb, (SweepItem *)(mem + off_items)}, (uint64_t)nbr + 1, st);
    A->spmv_chunks = (uint32_t *)mem;
    A->spmv_num_chunks = n_items;
    A->spmv_plan_long = n_long;
    A->spmv_plan_off_cnt = off_cnt;
    A->spmv_plan_off_carry = off_carry;
}

namespace {

// ---------------------------------------------------------------------------------------------------------
// the sweep kernel
// ---------------------------------------------------------------------------------------------------------
// Buffer (SRSRC) loads: 32-bit byte offsets against a wave-uniform descriptor, and an out-of-range offset returns 0.
// That removes the 64-bit address arithmetic and -- by steering absent elements to offset ~0 -- every exec-mask branch
// of the predicated gathers; x's descriptor ends at num_cols, so the ragged last block column reads as 0 by itself.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t kOob = 0xffffffffu;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
template <typename T>
struct Buf;
template <>
struct Buf<float> {
    static __device__ __forceinline__ float ld(rsrc_t r, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
};
template <>
struct Buf<_Float16> {
    static __device__ __forceinline__ float ld(rsrc_t r, uint32_t off) { return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0)); }
};
template <>
struct Buf<double> {
    static __device__ __forceinline__ double ld(rsrc_t r, uint32_t off) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0)); }
};

// four consecutive elements with one buffer load (16 bytes for float): the full-tile path below
template <typename T>
struct Buf4;
template <>
struct Buf4<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void ld(rsrc_t r, uint32_t off, float (&o)[4])
    {
        const v4 v = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
};
template <>
struct Buf4<_Float16> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void ld(rsrc_t r, uint32_t off, float (&o)[4])
    {
        const h4 v = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    }
};
template <>
struct Buf4<double> {
    typedef double d2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void ld(rsrc_t r, uint32_t off, double (&o)[4])
    {
        const d2 a = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        const d2 b = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u, 0, 0));
        o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
    }
};

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}

constexpr int kSparseMax = 8;   // tiles with more stored values than this go to the wave-wide dense pass (6 would fit 7 waves per SIMD -- 71 VGPRs instead of 83 -- but sends more tiles to the dense pass: 44 vs 35.7 us on the webbase-like matrix)
constexpr int kInlineSlots = 2; // stored values of a tile handled in the streaming loop; the rest of a tile waits in a queue
                                // (1, 2 and 3 measure the same within 2 %)
constexpr int kDenseTrip = 8;   // dense tiles per trip of the wave-wide pass
constexpr uint32_t kLeftCap = 128;  // leftover queue: flushed 64 tiles at a time, so every lane has work
constexpr uint32_t kDenseCap = 64;  // dense queue, swept after each half batch
constexpr int kFullTrip = 4;        // FULL tiles (all 64 values stored): 4 tiles per wave step x kFullTrip steps in flight

template <typename A>
__device__ __forceinline__ void lds_add(A *p, A v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // ds_add_f32 / ds_add_f64
}

// sum over the 8 lanes of a tile row (aligned groups of 8 lanes) with DPP adds -- no LDS traffic
__device__ __forceinline__ float row8_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}
__device__ __forceinline__ double row8_sum(double v)
{
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

// up to N stored elements of one tile, handled by ONE lane: positions peeled off the top of the bitmap, all value / x
// loads issued back to back (absent ones point out of range and read 0), then the LDS adds (exec-masked).  Returns the
// bitmap of the elements that are left.
template <typename T, typename A, int N>
__device__ __forceinline__ uint64_t peel_tile(uint64_t bm, uint32_t voff, uint32_t xbase, A *__restrict__ trow, rsrc_t rv, rsrc_t rx)
{
    A a[N], xv[N];
    uint32_t pr[N];
    bool has[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        has[j] = bm != 0;
        const uint32_t p = (uint32_t)__clzll((long long)bm) & 63u;
        bm &= ~(0x8000000000000000ull >> p);
        pr[j] = p >> 3;
        a[j] = Buf<T>::ld(rv, has[j] ? voff + (uint32_t)(j * sizeof(T)) : kOob);
        xv[j] = Buf<T>::ld(rx, has[j] ? xbase + (p & 7u) * (uint32_t)sizeof(T) : kOob);
    }
#pragma unroll
    for (int j = 0; j < N; j++)
        if (has[j]) lds_add(trow + pr[j], a[j] * xv[j]);
    return bm;
}

// the same in two steps, so that the requests of BOTH tiles of a lane can be in flight before the first LDS add
template <typename T, typename A, int N>
struct Peeled {
    A a[N], xv[N];
    uint32_t pr[N];
    bool has[N];
    uint64_t rest;
};
template <typename T, typename A, int N>
__device__ __forceinline__ Peeled<T, A, N> peel_issue(uint64_t bm, uint32_t voff, uint32_t xbase, rsrc_t rv, rsrc_t rx)
{
    Peeled<T, A, N> q;
#pragma unroll
    for (int j = 0; j < N; j++) {
        q.has[j] = bm != 0;
        const uint32_t p = (uint32_t)__clzll((long long)bm) & 63u;
        bm &= ~(0x8000000000000000ull >> p);
        q.pr[j] = p >> 3;
        q.a[j] = Buf<T>::ld(rv, q.has[j] ? voff + (uint32_t)(j * sizeof(T)) : kOob);
        q.xv[j] = Buf<T>::ld(rx, q.has[j] ? xbase + (p & 7u) * (uint32_t)sizeof(T) : kOob);
    }
    q.rest = bm;
    return q;
}
template <typename T, typename A, int N>
__device__ __forceinline__ void peel_commit(const Peeled<T, A, N> &q, A *__restrict__ trow)
{
#pragma unroll
    for (int j = 0; j < N; j++)
        if (q.has[j]) lds_add(trow + q.pr[j], q.a[j] * q.xv[j]);
}

// FULL: matrices whose plan counted many full tiles (FEM-like) get the variant with the 16-byte-per-lane full-tile pass; the
// graph-like default keeps the leaner code (the extra pass costs the webbase-like case 7 % even when it never runs)
template <typename T, bool FULL, bool NT, bool PERSIST = false, int OCC = 1>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(OCC, 8))) void spmv_sweep_kernel(const SweepItem *__restrict__ items, uint32_t num_items,
                                                              const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bmps,
                                                              const uint64_t *__restrict__ offsets, const T *__restrict__ values,
                                                              const T *__restrict__ x, typename Acc<T>::type *__restrict__ y,
                                                              typename Acc<T>::type *__restrict__ carry, uint32_t *__restrict__ counters,
                                                              uint32_t num_rows, uint32_t num_cols, uint32_t values_bytes, uint32_t row_lo)
{
    using A = typename Acc<T>::type;
    __shared__ A tile_all[4][kItemRows * 8];
    // queues of tiles waiting for a full-wave pass: (bitmap, value byte offset, x byte offset, u-tile row base)
    __shared__ uint64_t l_bmp_all[4][kLeftCap], d_bmp_all[4][kDenseCap];
    __shared__ uint32_t l_off_all[4][kLeftCap], l_xb_all[4][kLeftCap], l_tb_all[4][kLeftCap];
    __shared__ uint32_t d_off_all[4][kDenseCap], d_xb_all[4][kDenseCap], d_tb_all[4][kDenseCap];
    __shared__ uint32_t f_off_all[4][kDenseCap], f_xb_all[4][kDenseCap], f_tb_all[4][kDenseCap];
    const int w = wave_id(), lane = lane_id();
    A *tile = tile_all[w];
    uint64_t *l_bmp = l_bmp_all[w], *d_bmp = d_bmp_all[w];
    uint32_t *l_off = l_off_all[w], *l_xb = l_xb_all[w], *l_tb = l_tb_all[w];
    uint32_t *d_off = d_off_all[w], *d_xb = d_xb_all[w], *d_tb = d_tb_all[w];
    uint32_t *f_off = f_off_all[w], *f_xb = f_xb_all[w], *f_tb = f_tb_all[w];
    const rsrc_t rv = make_rsrc(values, values_bytes), rx = make_rsrc(x, num_cols * (uint32_t)sizeof(T));
    const uint64_t lt = lanemask_lt();
    // PERSIST: the grid is sized to the chip and every wave walks items with the grid stride (no relaunch gaps)
    for (uint32_t item_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w); item_id < num_items; item_id += PERSIST ? gridDim.x * 4 : num_items) {
    const SweepItem it = items[item_id];
    tile[lane] = A(0);
    tile[64 + lane] = A(0);
    uint32_t n_left = 0;  // wave-uniform fill of the leftover queue
    __builtin_amdgcn_wave_barrier();

    for (uint32_t base = it.blk_begin; base < it.blk_end; base += kBatch) {
        // lane-per-tile: three coalesced streams, two tiles per lane
        const uint32_t b0 = base + lane, b1 = base + 64 + lane;
        uint64_t bm0 = 0, bm1 = 0, k0 = (uint64_t)it.row_begin << 32, k1 = k0, o0 = 0, o1 = 0;
        if (NT) {  // streamed once: non-temporal, so that the tile words do not push x out of the L2
            if (b0 < it.blk_end) { bm0 = __builtin_nontemporal_load(bmps + b0); k0 = __builtin_nontemporal_load(keys + b0); o0 = __builtin_nontemporal_load(offsets + b0); }
            if (b1 < it.blk_end) { bm1 = __builtin_nontemporal_load(bmps + b1); k1 = __builtin_nontemporal_load(keys + b1); o1 = __builtin_nontemporal_load(offsets + b1); }
        } else {
            if (b0 < it.blk_end) { bm0 = bmps[b0]; k0 = keys[b0]; o0 = offsets[b0]; }
            if (b1 < it.blk_end) { bm1 = bmps[b1]; k1 = keys[b1]; o1 = offsets[b1]; }
        }
        const uint32_t x0 = key_col(k0) * 8u * (uint32_t)sizeof(T), x1 = key_col(k1) * 8u * (uint32_t)sizeof(T);
        const uint32_t tb0 = (key_row(k0) - it.row_begin) * 8u, tb1 = (key_row(k1) - it.row_begin) * 8u;
        const uint32_t vo0 = (uint32_t)o0 * (uint32_t)sizeof(T), vo1 = (uint32_t)o1 * (uint32_t)sizeof(T);
        // the inline element requests of both tiles go out together (one memory round trip for the pair, not two)
        const bool sp0 = !(FULL && bm0 == ~0ull) && __popcll(bm0) <= kSparseMax, sp1 = !(FULL && bm1 == ~0ull) && __popcll(bm1) <= kSparseMax;
        const Peeled<T, A, kInlineSlots> pe0 = peel_issue<T, A, kInlineSlots>(sp0 ? bm0 : 0ull, vo0, x0, rv, rx);
        const Peeled<T, A, kInlineSlots> pe1 = peel_issue<T, A, kInlineSlots>(sp1 ? bm1 : 0ull, vo1, x1, rv, rx);
        // the two tiles of a lane are queued one after the other so that the queues never hold more than 64 new entries
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint64_t bm = h ? bm1 : bm0;
            const uint32_t vo = h ? vo1 : vo0, xb = h ? x1 : x0, tb = h ? tb1 : tb0;
            // full tiles have their own queue: no bitmap, values at fixed places
            const bool full = FULL && bm == ~0ull && (sizeof(T) >= 4 || (vo & 3u) == 0u);  // multi-dword loads want dword-aligned values
            const uint64_t fm = __ballot(full);
            const int fn = __popcll(fm);
            if (full) {
                const int s = __popcll(fm & lt);
                f_off[s] = vo; f_xb[s] = xb; f_tb[s] = tb;
            }
            // other dense tiles are queued for the wave-wide pass
            const bool dense = !full && __popcll(bm) > kSparseMax;
            const uint64_t dm = __ballot(dense);
            const int qn = __popcll(dm);
            if (dense) {
                const int s = __popcll(dm & lt);
                d_bmp[s] = bm; d_off[s] = vo; d_xb[s] = xb; d_tb[s] = tb;
            }
            // every sparse tile: its first kInlineSlots stored values right here (covers most tiles of a graph matrix entirely)
            peel_commit<T, A, kInlineSlots>(h ? pe1 : pe0, tile + tb);
            const uint64_t rest = h ? pe1.rest : pe0.rest;
            // tiles with more values wait in the leftover queue until 64 of them make a full wave
            const uint64_t lm = __ballot(rest != 0);
            if (rest) {
                const uint32_t s = n_left + (uint32_t)__popcll(lm & lt);
                l_bmp[s] = rest; l_off[s] = vo + kInlineSlots * (uint32_t)sizeof(T); l_xb[s] = xb; l_tb[s] = tb;
            }
            n_left += (uint32_t)__popcll(lm);
            __builtin_amdgcn_wave_barrier();
            if (n_left >= 64u) {
                n_left -= 64u;
                const uint32_t s = n_left + (uint32_t)lane;
                peel_tile<T, A, kSparseMax - kInlineSlots>(l_bmp[s], l_off[s], l_xb[s], tile + l_tb[s], rv, rx);
            }
            // full tiles: 16 lanes per tile, each lane four consecutive values of one tile row (one 16-byte load) against the
            // matching four x entries; the two half-rows meet in one DPP add; four tiles per step, kFullTrip steps in flight
            for (int q = 0; FULL && q < fn; q += 4 * kFullTrip) {
                const int ts = lane >> 4, rr = (lane >> 1) & 7, hf = lane & 1;
                A part[kFullTrip];
#pragma unroll
                for (int t = 0; t < kFullTrip; t++) {
                    const int e = q + 4 * t + ts;
                    const bool on = e < fn;
                    const int ee = on ? e : 0;
                    A a4[4], x4[4];
                    Buf4<T>::ld(rv, on ? f_off[ee] + (uint32_t)((rr * 8 + hf * 4) * sizeof(T)) : kOob, a4);
                    Buf4<T>::ld(rx, on ? f_xb[ee] + (uint32_t)(hf * 4 * sizeof(T)) : kOob, x4);
                    part[t] = a4[0] * x4[0] + a4[1] * x4[1] + a4[2] * x4[2] + a4[3] * x4[3];
                }
#pragma unroll
                for (int t = 0; t < kFullTrip; t++) {
                    const int e = q + 4 * t + ts;
                    const A other = __shfl_xor(part[t], 1, kWave);
                    if (e < fn && hf == 0) lds_add(tile + f_tb[e] + rr, part[t] + other);
                }
            }
            // dense tiles: the whole wave per tile, lane p owns tile position p (coalesced value loads), kDenseTrip tiles per trip
            for (int q = 0; q < qn; q += kDenseTrip) {
                A pa[kDenseTrip];
#pragma unroll
                for (int t = 0; t < kDenseTrip; t++) {
                    const int e = min(q + t, qn - 1);
                    const uint64_t bb = q + t < qn ? d_bmp[e] : 0ull;
                    const bool has = tile_has(bb, lane);
                    const A av = Buf<T>::ld(rv, has ? d_off[e] + (uint32_t)tile_rank(bb, lane) * (uint32_t)sizeof(T) : kOob);
                    const A xx = Buf<T>::ld(rx, has ? d_xb[e] + ((uint32_t)lane & 7u) * (uint32_t)sizeof(T) : kOob);
                    pa[t] = av * xx;
                }
#pragma unroll
                for (int t = 0; t < kDenseTrip; t++) {
                    const int e = min(q + t, qn - 1);
                    const A sum = row8_sum(pa[t]);
                    if (q + t < qn && (lane & 7) == 0 && tile_byte(d_bmp[e], lane >> 3)) lds_add(tile + d_tb[e] + (lane >> 3), sum);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the leftover tiles that never filled a wave
    if (n_left) {
        const bool on = (uint32_t)lane < n_left;
        peel_tile<T, A, kSparseMax - kInlineSlots>(on ? l_bmp[lane] : 0ull, on ? l_off[lane] : 0u, on ? l_xb[lane] : 0u, tile + (on ? l_tb[lane] : 0u), rv, rx);
    }
    __builtin_amdgcn_wave_barrier();

    if (it.num_items == 0) {
        // short item: the wave owns u[row_begin*8, row_end*8)
        const uint32_t n_out = (it.row_end - it.row_begin) * 8u, out0 = it.row_begin * 8u;
        for (uint32_t e = lane; e < n_out; e += 64)
            if (out0 + e >= row_lo && out0 + e < num_rows) y[out0 + e] = tile[e];
        __builtin_amdgcn_wave_barrier();
        continue;
    }
    // long row: park the partial sums, the last arriver folds them.  Write-through (sc1) stores + drained vmcnt +
    // agent-scope counter add; the wave whose add comes last reads every slot with sc1 loads (MI355X_MICROARCH.md
    // "Valid forms": every store and load of the handed-off bytes sc1, no cache-wide fence -- a release fence per item
    // writes back the whole L2 thousands of times per launch: measured 10x slower).
    if (lane < 8) __hip_atomic_store(&carry[(size_t)item_id * 8 + lane], tile[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(&counters[it.long_idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != it.num_items - 1) continue;
    // lanes 0..7 own tile row `r`; lane group g = lane/8 walks items g, g+8, ...; fixed order
    const int r = lane & 7, g = lane >> 3;
    A sum = 0;
    for (uint32_t c = g; c < it.num_items; c += 8)
        sum += __hip_atomic_load(&carry[(size_t)(it.first_item + c) * 8 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) sum += __shfl_xor(sum, d, kWave);
    const uint32_t row = it.row_begin * 8u + (uint32_t)r;
    if (g == 0 && row >= row_lo && row < num_rows) y[row] = sum;
    if (lane == 0) __hip_atomic_store(&counters[it.long_idx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------
// value-stream sweep (round 2; the default for matrices whose tiles are mostly sparse)
// ---------------------------------------------------------------------------------------------------------
// Same plan, same items, same epilogue as the sweep above, but the elements are walked LANE PER STORED VALUE instead of lane per tile:
//   1. lane-per-tile: the tile words of up to 128 tiles (two per lane, coalesced); the batch is the longest run of those tiles that holds
//      at most 512 stored values;
//   2. every stored value has a 16-bit ENTRY {tile slot in the batch, position in the tile} at its index inside the batch.  kDecode
//      builds the entries in LDS from the bitmaps (a loop of max-popcount trips over the tile lanes: 36 % of the kernel's VALU work on
//      the webbase-like matrix, 52 % on the cage-like one).  kCached reads them from the POSITION CACHE -- the same entries, written
//      once per matrix by kBuild into a 2-byte-per-value array that sits beside the values (part of the cached SpMV plan, like the
//      block-row pointer and the items; bmsp_matrix_prepare builds it, BMSP_SPMV_NO_POSCACHE=1 or a cache above BMSP_SPMV_POSCACHE_MAX
//      bytes keeps the decode in the kernel).  With the cache the bitmaps are not read at all: keys + next offsets + entries + values.
//   3. lane-per-value: value v is ONE coalesced load of the value array (no value gather), its x entry one gather with every lane busy.
//   4. reduction into the item's u tile in LDS.  kAtomic: ds_add_f32 -- 3 cycles per active lane on gfx950
//      (experiments/lds_atomic_rate.hip), affordable at < 2 values per tile.  kSorted: the products are counting-sorted by row with
//      INTEGER LDS atomics (full rate), run sums are taken with DPP steps, the last lane of a run adds to the tile with a plain
//      read-modify-write.
#ifndef BMSP_VS_TILES
#define BMSP_VS_TILES 128
#endif
#ifndef BMSP_VS_PRE
#define BMSP_VS_PRE 4
#endif
constexpr uint32_t kVsTiles = BMSP_VS_TILES;  // tiles per batch at most
constexpr int kVsTpl = (int)kVsTiles / 64;    // ... = tiles per lane
constexpr uint32_t kVsVals = 512;   // values per batch at most
constexpr int kVsChunks = kVsVals / 64;
enum { kDecode = 0, kCached = 1, kBuild = 2 };
enum { kAtomic = 0, kSorted = 1 };

// lane i receives `src` of the lane the DPP control names; lanes without a source (or outside ROW_MASK) keep `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_val(float old, float src)
{
    return __builtin_bit_cast(float, dpp_u32<CTRL, ROW_MASK>(__builtin_bit_cast(uint32_t, old), __builtin_bit_cast(uint32_t, src)));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_val(double old, double src)
{
    const uint64_t o = __builtin_bit_cast(uint64_t, old), v = __builtin_bit_cast(uint64_t, src);
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)o, (uint32_t)v), hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(o >> 32), (uint32_t)(v >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// inclusive sums over runs of equal `key` in lane order (keys are sorted, so equal keys d lanes apart mean one run)
template <typename A>
__device__ __forceinline__ A wave_run_sum(A v, uint32_t key)
{
#define BMSP_RUN_STEP(CTRL, RM)                                   \
    {                                                             \
        const A pv = dpp_val<CTRL, RM>(A(0), v);                  \
        const uint32_t pk = dpp_u32<CTRL, RM>(0xffffffffu, key);  \
        v += pk == key ? pv : A(0);                               \
    }
    BMSP_RUN_STEP(0x111, 0xf)  // row_shr:1
    BMSP_RUN_STEP(0x112, 0xf)  // row_shr:2
    BMSP_RUN_STEP(0x114, 0xf)  // row_shr:4
    BMSP_RUN_STEP(0x118, 0xf)  // row_shr:8
    BMSP_RUN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1 and 3
    BMSP_RUN_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2 and 3
#undef BMSP_RUN_STEP
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_sum_dpp(uint32_t v)
{
    v += dpp_u32<0x111>(0, v);
    v += dpp_u32<0x112>(0, v);
    v += dpp_u32<0x114>(0, v);
    v += dpp_u32<0x118>(0, v);
    v += dpp_u32<0x142, 0xa>(0, v);
    v += dpp_u32<0x143, 0xc>(0, v);
    return v;
}

template <typename A, int MODE, int RED>
struct VsLds {
    // u tile (16 block-rows x 8) | tile slots | [row counters, then row starts] | [entries, then the products in row order] | [their rows]
    static constexpr uint32_t tile = 0, tinfo = tile + kItemRows * 8 * sizeof(A), cnt = tinfo + kVsTiles * 4;
    static constexpr uint32_t region = cnt + (RED == kSorted ? kItemRows * 8 * 4 : 0);
    static constexpr uint32_t region_bytes = (RED == kSorted ? kVsVals * sizeof(A) : (MODE == kCached ? 0 : kVsVals * 2));
    static constexpr uint32_t srow = region + region_bytes;
    static constexpr uint32_t bytes = srow + (RED == kSorted ? kVsVals : 0);
};

// what a wave requests for an item before it works on it (kCached): the keys of its first 128 tiles, their end offsets unless the item is
// a single batch, and entries + values of its first 256 values
constexpr int kPre = BMSP_VS_PRE;
template <typename A>
struct VsPre {
    uint32_t ti[kVsTpl];  // the tile's slot word {block column, block-row inside the item's window}, from the tile cache
    uint32_t oe[kVsTpl];  // end of the tile's values relative to the item's first value (multi-batch items only)
    uint32_t e[kPre];
    A a[kPre];
};
// values of the item (short items carry their value range; a long-row item holds 256 tiles, its count is not in the plan)
__host__ __device__ __forceinline__ uint32_t vs_item_values(const SweepItem &it) { return it.num_items == 0 ? it.first_item - it.val_begin : 64u * kPre; }
// the common case: the whole item is one batch, and the item says how many values that is -- no offset is read
__host__ __device__ __forceinline__ bool vs_single(const SweepItem &it)
{
    return it.num_items == 0 && it.blk_end - it.blk_begin <= kVsTiles && vs_item_values(it) <= kVsVals;
}
template <typename T>
__device__ __forceinline__ void vs_request(const SweepItem &it, int lane, const uint32_t *__restrict__ tcache, const uint16_t *__restrict__ eoff,
                                           rsrc_t rv, rsrc_t rp, uint32_t pos_base, VsPre<typename Acc<T>::type> &pre)
{
    const uint32_t bend = min(it.blk_begin + kVsTiles, it.blk_end);
    const bool single = vs_single(it);
#pragma unroll
    for (int t = 0; t < kVsTpl; t++) {
        const uint32_t b = it.blk_begin + 64u * (uint32_t)t + (uint32_t)lane;
        pre.ti[t] = 0;
        pre.oe[t] = 0;
        if (b < bend) pre.ti[t] = tcache[b];
        if (!single && b < bend) pre.oe[t] = (uint32_t)eoff[b];
    }
    const uint32_t n_item = vs_item_values(it);
#pragma unroll
    for (int u = 0; u < kPre; u++) {
        const uint32_t idx = 64u * (uint32_t)u + (uint32_t)lane;
        const bool on = idx < n_item;
        pre.e[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rp, on ? (it.val_begin - pos_base + idx) * 2u : kOob, 0, 0);
        pre.a[u] = Buf<T>::ld(rv, on ? (it.val_begin + idx) * (uint32_t)sizeof(T) : kOob);
    }
}

// One wave per workgroup (a finished wave frees its slot and its LDS at once), one item per wave.  (A resident grid whose waves walk items
// id, id + grid, ... with the next item's requests in flight behind the current item's x gathers was built and measured: 33.9 us instead
// of 28.3 on the webbase-like matrix -- 79 / 114 VGPRs for the two request sets, and the hardware's wave dispatcher balances one-item
// waves better than a static walk does.)
template <typename T, int MODE, int RED>
__global__ __launch_bounds__(64, sizeof(typename Acc<T>::type) == 4 ? 8 : 1) void spmv_vstream_kernel(const SweepItem *__restrict__ items, uint32_t num_items,
                                                          const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bmps,
                                                          const uint64_t *__restrict__ offsets, const T *__restrict__ values,
                                                          const T *__restrict__ x, typename Acc<T>::type *__restrict__ y,
                                                          typename Acc<T>::type *__restrict__ carry, uint32_t *__restrict__ counters,
                                                          uint32_t num_rows, uint32_t num_cols, uint32_t values_bytes,
                                                          uint16_t *__restrict__ pos, uint32_t pos_base, uint32_t pos_count, uint32_t row_lo,
                                                          uint32_t *__restrict__ tcache, uint16_t *__restrict__ eoff)
{
    using A = typename Acc<T>::type;
    using L = VsLds<A, MODE, RED>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[L::bytes];
    A *const tile = (A *)(lds + L::tile);
    uint32_t *const tinfo = (uint32_t *)(lds + L::tinfo);
    uint32_t *const cnt = (uint32_t *)(lds + L::cnt);
    uint16_t *const ent = (uint16_t *)(lds + L::region);
    A *const sorted = (A *)(lds + L::region);
    uint8_t *const srow = lds + L::srow;
    const int lane = lane_id();
    const rsrc_t rv = make_rsrc(values, values_bytes), rx = make_rsrc(x, num_cols * (uint32_t)sizeof(T));
    const rsrc_t rp = make_rsrc(pos, pos_count * 2u);
    const uint32_t item_id = blockIdx.x;
    const SweepItem it = items[item_id];
    VsPre<A> pre;
    if (MODE == kCached) vs_request<T>(it, lane, tcache, eoff, rv, rp, pos_base, pre);
    {
        tile[lane] = A(0);
        tile[64 + lane] = A(0);
        const uint32_t n_item = vs_item_values(it);
        const bool single = MODE == kCached && vs_single(it);
        const bool long_item = RED == kAtomic && it.num_items != 0;  // (in the kSorted kernels the eight sums cost registers the sort needs: 64 -> spills)
        A acc8[8] = {A(0), A(0), A(0), A(0), A(0), A(0), A(0), A(0)};

        for (uint32_t base = it.blk_begin; base < it.blk_end;) {
            const bool first = MODE == kCached && base == it.blk_begin;
            const uint32_t bend = min(base + kVsTiles, it.blk_end);
            uint64_t bm[kVsTpl], kk[kVsTpl];
            uint32_t ti[kVsTpl];              // kCached: the tile's slot word straight from the tile cache
            uint32_t ob[kVsTpl], eb[kVsTpl];  // start / end of the tile's values, relative to the batch's first value
            uint32_t v_first = first ? it.val_begin : 0u;
            uint32_t v_rel = 0;               // kCached, later batches: the batch's first value relative to the item's
            if (MODE == kCached && !first) { v_rel = (uint32_t)eoff[base - 1]; v_first = it.val_begin + v_rel; }
#pragma unroll
            for (int t = 0; t < kVsTpl; t++) {
                const uint32_t b = base + 64u * (uint32_t)t + (uint32_t)lane;
                bm[t] = 0; kk[t] = (uint64_t)it.row_begin << 32; ob[t] = 0; eb[t] = kOob; ti[t] = 0;
                if (first) {
                    ti[t] = pre.ti[t];
                    if (b < bend) eb[t] = single ? 0u : pre.oe[t];
                } else if (MODE == kCached) {
                    if (b < bend) { ti[t] = tcache[b]; eb[t] = (uint32_t)eoff[b] - v_rel; }
                } else {
                    if (b < bend) { bm[t] = bmps[b]; kk[t] = keys[b]; ob[t] = (uint32_t)offsets[b]; }
                }
            }
            if (MODE != kCached) {
                v_first = __builtin_amdgcn_readfirstlane(ob[0]);
#pragma unroll
                for (int t = 0; t < kVsTpl; t++) {
                    ob[t] -= v_first;
                    eb[t] = ob[t] + (uint32_t)__popcll(bm[t]);  // lanes past the batch: 0 - v_first, far above the cut
                }
            }
            // the batch: tiles whose values end inside the first kVsVals values (a prefix: offsets ascend; >= 8 tiles, a tile holds <= 64)
            bool ok[kVsTpl];
            uint32_t nb = 0;
#pragma unroll
            for (int t = 0; t < kVsTpl; t++) {
                ok[t] = base + 64u * (uint32_t)t + (uint32_t)lane < bend && eb[t] <= kVsVals;
                nb += (uint32_t)__popcll(__ballot(ok[t]));
            }
            uint32_t nvals = n_item;
            if (!single) {
                uint32_t el = eb[0];
#pragma unroll
                for (int t = 1; t < kVsTpl; t++) el = (nb - 1u) >> 6 == (uint32_t)t ? eb[t] : el;
                nvals = (uint32_t)__builtin_amdgcn_readlane((int)el, (int)((nb - 1u) & 63u));
            }
            // tile slots: column and row-in-window of every tile of the batch
#pragma unroll
            for (int t = 0; t < kVsTpl; t++) tinfo[64 * t + lane] = MODE == kCached ? ti[t] : (key_col(kk[t]) | ((key_row(kk[t]) - it.row_begin) << 28));
            if (RED == kSorted) {
                cnt[lane] = 0;
                cnt[64 + lane] = 0;
            }
            if (MODE != kCached) {
                // entries from the bitmaps: {slot, position} of every stored value, at the value's index inside the batch
                uint64_t any = 0;
#pragma unroll
                for (int t = 0; t < kVsTpl; t++) {
                    bm[t] = ok[t] ? bm[t] : 0;
                    any |= bm[t];
                }
                while (__any(any != 0)) {
                    any = 0;
#pragma unroll
                    for (int t = 0; t < kVsTpl; t++) {
                        if (bm[t]) {
                            const uint32_t p = (uint32_t)__clzll((long long)bm[t]);
                            bm[t] &= ~(0x8000000000000000ull >> p);
                            ent[ob[t]++] = (uint16_t)(((64u * (uint32_t)t + (uint32_t)lane) << 6) | p);
                        }
                        any |= bm[t];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (MODE == kBuild) {
                // the tile cache: slot word and end of the tile's values relative to the item's first value (every tile of the batch)
#pragma unroll
                for (int t = 0; t < kVsTpl; t++) {
                    const uint32_t b = base + 64u * (uint32_t)t + (uint32_t)lane;
                    if (ok[t]) {
                        tcache[b] = key_col(kk[t]) | ((key_row(kk[t]) - it.row_begin) << 28);
                        eoff[b] = (uint16_t)(v_first + eb[t] - it.val_begin);
                    }
                }
                for (uint32_t idx = (uint32_t)lane; idx < nvals; idx += 64u) pos[v_first - pos_base + idx] = ent[idx];
                __builtin_amdgcn_wave_barrier();
                base += nb;
                continue;
            }
            const uint32_t pos_first = (v_first - pos_base) * 2u;

            if (RED == kAtomic || long_item) {
                for (uint32_t c0 = 0; c0 < nvals; c0 += 256u) {
                    A av[4], xv[4];
                    uint32_t rowl[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t idx = c0 + 64u * (uint32_t)u + (uint32_t)lane;
                        const bool on = idx < nvals;
                        uint32_t e;
                        if (first && c0 == 0) {
                            e = pre.e[u];
                            av[u] = pre.a[u];
                        } else {
                            e = MODE == kCached ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rp, on ? pos_first + idx * 2u : kOob, 0, 0) : (on ? (uint32_t)ent[idx] : 0u);
                            av[u] = Buf<T>::ld(rv, on ? (v_first + idx) * (uint32_t)sizeof(T) : kOob);
                        }
                        const uint32_t ti = tinfo[e >> 6], p = e & 63u;
                        rowl[u] = (ti >> 28) * 8u + (p >> 3);
                        xv[u] = Buf<T>::ld(rx, on ? ((ti & 0x0fffffffu) * 8u + (p & 7u)) * (uint32_t)sizeof(T) : kOob);
                    }
                    if (long_item) {
                        // a long block-row: every product belongs to one of EIGHT rows -- eight register sums per lane, folded once
                        // per item, instead of 64 LDS float adds on eight addresses per instruction
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const A pr = c0 + 64u * (uint32_t)u + (uint32_t)lane < nvals ? av[u] * xv[u] : A(0);
#pragma unroll
                            for (int k = 0; k < 8; k++) acc8[k] += rowl[u] == (uint32_t)k ? pr : A(0);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (c0 + 64u * (uint32_t)u + (uint32_t)lane < nvals) lds_add(tile + rowl[u], av[u] * xv[u]);
                    }
                }
            } else {
                // the value's rank among the values of its row: integer LDS counter
                A prod[kVsChunks];
                A xg[kVsChunks];
                uint32_t rr[kVsChunks];  // row | rank << 8
#pragma unroll
                for (int u = 0; u < kVsChunks; u++) {
                    if (64u * (uint32_t)u < nvals) {
                        const uint32_t idx = 64u * (uint32_t)u + (uint32_t)lane;
                        const bool on = idx < nvals;
                        uint32_t e;
                        A av;
                        if (first && u < kPre) {
                            e = pre.e[u];
                            av = pre.a[u];
                        } else {
                            e = MODE == kCached ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rp, on ? pos_first + idx * 2u : kOob, 0, 0) : (on ? (uint32_t)ent[idx] : 0u);
                            av = Buf<T>::ld(rv, on ? (v_first + idx) * (uint32_t)sizeof(T) : kOob);
                        }
                        const uint32_t ti = tinfo[e >> 6], p = e & 63u;
                        const uint32_t rowl = (ti >> 28) * 8u + (p >> 3);
                        xg[u] = Buf<T>::ld(rx, on ? ((ti & 0x0fffffffu) * 8u + (p & 7u)) * (uint32_t)sizeof(T) : kOob);
                        uint32_t rank = 0;
                        if (on) rank = __hip_atomic_fetch_add(&cnt[rowl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        rr[u] = rowl | (rank << 8);
                        prod[u] = av;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // row counters -> row starts (exclusive sums; lane l owns rows 2l and 2l+1)
                {
                    const uint32_t c0 = cnt[2 * lane], c1 = cnt[2 * lane + 1];
                    const uint32_t ex = wave_inclusive_sum_dpp(c0 + c1) - (c0 + c1);
                    cnt[2 * lane] = ex;
                    cnt[2 * lane + 1] = ex + c0;
                }
                __builtin_amdgcn_wave_barrier();
                // products into row order (the entries of kDecode live in the same bytes: all of them were read above)
#pragma unroll
                for (int u = 0; u < kVsChunks; u++) {
                    if (64u * (uint32_t)u < nvals) {
                        if (64u * (uint32_t)u + (uint32_t)lane < nvals) {
                            const uint32_t rowl = rr[u] & 0xffu, at = cnt[rowl] + (rr[u] >> 8);
                            sorted[at] = prod[u] * xg[u];
                            srow[at] = (uint8_t)rowl;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // run sums; the last lane of every run adds its sum to the u tile (one lane per row and chunk: no atomic)
#pragma unroll
                for (int u = 0; u < kVsChunks; u++) {
                    if (64u * (uint32_t)u < nvals) {
                        const uint32_t idx = 64u * (uint32_t)u + (uint32_t)lane;
                        const bool on = idx < nvals;
                        const A sv = on ? sorted[idx] : A(0);
                        const uint32_t row = on ? (uint32_t)srow[idx] : 0xfeu;
                        const uint32_t nxt = (lane < 63 && idx + 1u < nvals) ? (uint32_t)srow[idx + 1u] : 0xffu;
                        const A run = wave_run_sum(sv, row);
                        if (on && nxt != row) tile[row] += run;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            base += nb;
        }

        if (MODE != kBuild) {
            if (it.num_items == 0) {
                // short item: the wave owns u[row_begin*8, row_end*8)
                const uint32_t n_out = (it.row_end - it.row_begin) * 8u, out0 = it.row_begin * 8u;
                for (uint32_t e = lane; e < n_out; e += 64)
                    if (out0 + e >= row_lo && out0 + e < num_rows) y[out0 + e] = tile[e];
            } else {
                // long row: park the partial sums, the last arriver folds them (same protocol as spmv_sweep_kernel)
                A mine = lane < 8 ? tile[lane & 7] : A(0);
                if (long_item) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const A sk = wave_sum(acc8[k]);
                        mine = lane == k ? sk : mine;
                    }
                }
                if (lane < 8) __hip_atomic_store(&carry[(size_t)item_id * 8 + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t ticket = 0;
                if (lane == 0) ticket = __hip_atomic_fetch_add(&counters[it.long_idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ticket = __builtin_amdgcn_readfirstlane(ticket);
                if (ticket == it.num_items - 1) {
                    const int r = lane & 7, g = lane >> 3;
                    A sum = 0;
                    for (uint32_t c = g; c < it.num_items; c += 8)
                        sum += __hip_atomic_load(&carry[(size_t)(it.first_item + c) * 8 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int d = 8; d < 64; d <<= 1) sum += __shfl_xor(sum, d, kWave);
                    const uint32_t row = it.row_begin * 8u + (uint32_t)r;
                    if (g == 0 && row >= row_lo && row < num_rows) y[row] = sum;
                    if (lane == 0) __hip_atomic_store(&counters[it.long_idx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// the position cache: kBuild pass over the plan's items, once per matrix
void build_pos_cache(bmsp_matrix_s *A, hipStream_t st)
{
    if (A->spmv_pos || A->spmv_pos_tried) return;
    A->spmv_pos_tried = 1;
    const char *cap_e = getenv("BMSP_SPMV_POSCACHE_MAX");
    const size_t cap = cap_e ? (size_t)strtoull(cap_e, nullptr, 10) : (size_t)4 << 30;
    if (getenv("BMSP_SPMV_NO_POSCACHE") || A->block_num == 0) return;
    const uint64_t base = A->view_values_end ? read_back(A->offsets, st) : 0;  // a row-panel view keeps the parent's absolute offsets
    const uint64_t count = (uint64_t)A->values_extent() - base;
    if (count == 0 || count * 2 > cap || count >= (1ull << 31)) return;
    // one allocation: entries (2 B per stored value) | tile slot words (4 B per tile) | tile value ends relative to the item (2 B per tile)
    const size_t nb = (size_t)A->block_num;
    const size_t off_ti = (count * 2 + 63) & ~size_t(63), off_eo = off_ti + ((nb * 4 + 63) & ~size_t(63));
    char *blk = (char *)pool_alloc(off_eo + nb * 2 + 64);
    uint16_t *pos = (uint16_t *)blk;
    A->spmv_tinfo = (uint32_t *)(blk + off_ti);
    A->spmv_eoff = (uint16_t *)(blk + off_eo);
    const char *mem = (const char *)A->spmv_chunks;
    const uint32_t n_items = (uint32_t)A->spmv_num_chunks;
    hipLaunchKernelGGL((spmv_vstream_kernel<float, kBuild, kAtomic>), dim3(n_items), dim3(64), 0, st, (const SweepItem *)(mem + 64), n_items, A->keys,
                       A->bmps, A->offsets, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, (float *)nullptr, (uint32_t *)nullptr,
                       (uint32_t)A->num_rows, (uint32_t)A->num_cols, 0u, pos, (uint32_t)base, (uint32_t)count, 0u, A->spmv_tinfo, A->spmv_eoff);
    BMSP_CHECK_LAUNCH();
    A->spmv_pos = pos;
    A->spmv_pos_base = (int64_t)base;
    A->spmv_pos_count = (int64_t)count;
}

// ---------------------------------------------------------------------------------------------------------
// block-row kernels (variants 1 and 2)
// ---------------------------------------------------------------------------------------------------------
// one tile row: byte = bits of row r (MSB = column 0), vals points at the first stored value of that row
template <typename T, typename A>
__device__ __forceinline__ A tile_row_dot(uint32_t byte, const T *__restrict__ vals, const T *__restrict__ x, uint32_t xbase,
                                          uint32_t num_cols, A acc)
{
    while (byte) {
        int c = __clz((int)byte) - 24;  // leading set bit of an 8-bit value -> column
        byte &= ~(0x80u >> c);
        uint32_t col = xbase + (uint32_t)c;
        A a = (A)(*vals++);
        A xv = col < num_cols ? (A)x[col] : A(0);
        acc = __builtin_fma(a, xv, acc);
    }
    return acc;
}

template <typename T, int LANES_PER_ROW>
__global__ __launch_bounds__(kThreads) void spmv_blockrow_kernel(const uint32_t *__restrict__ rowptr, const uint64_t *__restrict__ keys,
                                                                 const uint64_t *__restrict__ bmps, const uint64_t *__restrict__ offsets,
                                                                 const T *__restrict__ values, const T *__restrict__ x,
                                                                 typename Acc<T>::type *__restrict__ y, uint32_t num_rows,
                                                                 uint32_t num_cols, uint32_t num_block_rows, uint32_t row_lo)
{
    using A = typename Acc<T>::type;
    constexpr int GROUPS = LANES_PER_ROW / 8;  // tiles in flight per block-row per step
    const uint32_t gid = blockIdx.x * (kThreads / LANES_PER_ROW) + threadIdx.x / LANES_PER_ROW;
    if (gid >= num_block_rows) return;  // uniform per LANES_PER_ROW-lane group
    const int sub = threadIdx.x % LANES_PER_ROW;
    const int g = sub >> 3, r = sub & 7;
    const uint32_t b0 = rowptr[gid], b1 = rowptr[gid + 1];
    A acc = 0;
    for (uint32_t b = b0 + g; b < b1; b += GROUPS) {
        uint64_t bmp = bmps[b];
        uint32_t byte = tile_byte(bmp, r);
        if (byte) {
            uint32_t before = r ? (uint32_t)__popcll(bmp >> (64 - 8 * r)) : 0u;  // values stored in rows above
            acc = tile_row_dot<T, A>(byte, values + offsets[b] + before, x, key_col(keys[b]) * 8u, num_cols, acc);
        }
    }
    if (GROUPS > 1) {
#pragma unroll
        for (int d = 8; d < LANES_PER_ROW; d <<= 1) acc += __shfl_xor(acc, d, kWave);
    }
    uint32_t row = gid * 8u + (uint32_t)r;
    if (g == 0 && row >= row_lo && row < num_rows) y[row] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// row-group kernel (variant 3): matrices whose tiles are mostly dense (FEM / banded)
// ---------------------------------------------------------------------------------------------------------
// The sweep above is built for hyper-sparse tiles (a lane per TILE).  When a tile holds tens of values the bytes are in the value
// array, and the job is to stream it with 16-byte loads: here 16 lanes own one block-row (4 block-rows per wave) and walk its
// tiles; lane j of the group takes tile positions 4j .. 4j+3 (half a tile row): ONE 16-byte load of the <= 4 consecutive stored
// values behind rank(4j) -- the whole wave moves 1 KB of values per instruction when tiles are full -- against one 16-byte load
// of the four x entries (block column cached in L1/L2: neighbouring block-rows of a banded matrix read the same x lines).
// Partly filled nibbles are expanded with three selects; sums stay in registers for the whole block-row (no LDS, no atomics:
// y is bit-reproducible), the two half-rows meet in one DPP add, 8 lanes store the 8 rows.
template <typename T, int U>
__global__ __launch_bounds__(kThreads) void spmv_rowgroup_kernel(const uint32_t *__restrict__ rowptr, const uint64_t *__restrict__ keys,
                                                                 const uint64_t *__restrict__ bmps, const uint64_t *__restrict__ offsets,
                                                                 const T *__restrict__ values, const T *__restrict__ x,
                                                                 typename Acc<T>::type *__restrict__ y, uint32_t num_rows, uint32_t num_cols,
                                                                 uint32_t nbr, uint32_t values_bytes, uint32_t passes, uint32_t row_lo)
{
    using A = typename Acc<T>::type;
    const int lane = lane_id(), j = lane & 15, g = lane >> 4;
    const uint32_t wv = blockIdx.x * 4 + wave_id();
    const rsrc_t rv = make_rsrc(values, values_bytes), rx = make_rsrc(x, num_cols * (uint32_t)sizeof(T));
    const uint32_t xlane = (uint32_t)((j & 1) * 4);        // first of this lane's four tile columns
    const uint32_t bshift = (uint32_t)(60 - 4 * j) & 31u;   // nibble of positions 4j .. 4j+3 inside its bitmap word
    const bool in_hi = j < 8;
    const uint32_t hi_mask = j >= 8 ? 0xffffffffu : (j == 0 ? 0u : 0xffffffffu << (32 - 4 * j));  // bitmap bits of positions < 4j
    const uint32_t lo_mask = j > 8 ? 0xffffffffu << (64 - 4 * j) : 0u;
    const bool ragged_x = (num_cols & 3u) != 0u;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t br = (wv * passes + p) * 4u + (uint32_t)g;
        if ((wv * passes + p) * 4u >= nbr) break;  // wave-uniform
        const bool valid = br < nbr;
        uint32_t t = valid ? rowptr[br] : 0u;
        const uint32_t t1 = valid ? rowptr[br + 1] : 0u;
        A acc = 0;
        // tile words of the first trip; inside the loop the NEXT trip's words are requested behind the current trip's value / x
        // loads, so a trip waits for one round trip (its values), not two
        uint32_t blo[U], bhi[U], col[U], off[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool on = t + u < t1;
            const uint64_t bm = on ? bmps[t + u] : 0ull;
            blo[u] = (uint32_t)bm; bhi[u] = (uint32_t)(bm >> 32);
            col[u] = on ? (uint32_t)keys[t + u] : 0u;
            off[u] = on ? (uint32_t)offsets[t + u] : 0u;
        }
        while (__any(t < t1)) {
            A av[U][4], xv[U][4];
            uint32_t nib[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                nib[u] = ((in_hi ? bhi[u] : blo[u]) >> bshift) & 0xfu;
                const uint32_t rank = (uint32_t)__builtin_popcount(bhi[u] & hi_mask) + (uint32_t)__builtin_popcount(blo[u] & lo_mask);
                const uint32_t vaddr = (off[u] + rank) * (uint32_t)sizeof(T);
                if (sizeof(T) == 2) {
                    // halves start at any even byte: fetch the enclosing dwords and funnel-shift the odd leading half away
                    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const u32x3 d = __builtin_amdgcn_raw_buffer_load_b96(rv, nib[u] ? (vaddr & ~3u) : kOob, 0, 0);
                    const uint32_t sh = (vaddr & 2u) * 8u;
                    const h2 p0 = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(d[1], d[0], sh)), p1 = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(d[2], d[1], sh));
                    av[u][0] = (A)p0[0]; av[u][1] = (A)p0[1]; av[u][2] = (A)p1[0]; av[u][3] = (A)p1[1];
                } else {
                    Buf4<T>::ld(rv, nib[u] ? vaddr : kOob, av[u]);
                }
                const uint32_t xc = col[u] * 8u + xlane;
                if (ragged_x && xc + 4u > num_cols) {  // a 16-byte load that leaves x reads as zero altogether: last block column only
#pragma unroll
                    for (int q = 0; q < 4; q++) xv[u][q] = (A)Buf<T>::ld(rx, nib[u] && xc + q < num_cols ? (xc + q) * (uint32_t)sizeof(T) : kOob);
                } else {
                    Buf4<T>::ld(rx, nib[u] ? xc * (uint32_t)sizeof(T) : kOob, xv[u]);
                }
            }
            uint32_t nlo[U], nhi[U], ncol[U], noff[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool on = t + U + u < t1;
                const uint64_t bm = on ? bmps[t + U + u] : 0ull;
                nlo[u] = (uint32_t)bm; nhi[u] = (uint32_t)(bm >> 32);
                ncol[u] = on ? (uint32_t)keys[t + U + u] : 0u;
                noff[u] = on ? (uint32_t)offsets[t + U + u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t n = nib[u];
                A e0, e1, e2, e3;
                if (__all(n == 0xfu || n == 0u)) {  // full (or absent) nibbles everywhere: values already sit at their positions
                    e0 = av[u][0]; e1 = av[u][1]; e2 = av[u][2]; e3 = av[u][3];
                    if (n == 0u) { e0 = e1 = e2 = e3 = A(0); }
                } else {
                    // stored values of the nibble are consecutive: position q holds value number popc(bits before q)
                    const bool b0 = n & 8u, b1 = n & 4u, b2 = n & 2u, b3 = n & 1u;
                    const uint32_t i2 = (uint32_t)b0 + (uint32_t)b1, i3 = i2 + (uint32_t)b2;
                    e0 = b0 ? av[u][0] : A(0);
                    e1 = b1 ? (b0 ? av[u][1] : av[u][0]) : A(0);
                    e2 = b2 ? (i2 == 0 ? av[u][0] : (i2 == 1 ? av[u][1] : av[u][2])) : A(0);
                    e3 = b3 ? (i3 == 0 ? av[u][0] : (i3 == 1 ? av[u][1] : (i3 == 2 ? av[u][2] : av[u][3]))) : A(0);
                }
                acc += e0 * xv[u][0] + e1 * xv[u][1] + e2 * xv[u][2] + e3 * xv[u][3];
            }
            t += U;
#pragma unroll
            for (int u = 0; u < U; u++) { blo[u] = nlo[u]; bhi[u] = nhi[u]; col[u] = ncol[u]; off[u] = noff[u]; }
        }
        // lanes 2r and 2r+1 hold the two halves of tile row r
        if (sizeof(A) == 4) acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)acc), 0xB1, 0xf, 0xf, false));  // quad_perm 1,0,3,2
        else acc += __shfl_xor(acc, 1, kWave);
        const uint32_t row = br * 8u + (uint32_t)(j >> 1);
        if (valid && !(j & 1) && row >= row_lo && row < num_rows) y[row] = acc;
    }
}

template <typename T>
void launch(bmsp_matrix_s *A, const void *v, void *u, int variant, hipStream_t st, uint32_t row_lo, uint32_t row_hi)
{
    using Ac = typename Acc<T>::type;
    uint32_t nbr = (uint32_t)A->num_block_rows();
    if (nbr == 0) return;
    if (variant == BMSP_SPMV_BATCHED) {
        hipLaunchKernelGGL((spmv_blockrow_kernel<T, 64>), dim3((nbr + 3) / 4), dim3(kThreads), 0, st, A->rowptr, A->keys, A->bmps,
                           A->offsets, (const T *)A->values, (const T *)v, (Ac *)u, row_hi,
                           (uint32_t)A->num_cols, nbr, row_lo);
    } else if (variant == 2) {
        hipLaunchKernelGGL((spmv_blockrow_kernel<T, 8>), dim3((nbr + 31) / 32), dim3(kThreads), 0, st, A->rowptr, A->keys, A->bmps,
                           A->offsets, (const T *)A->values, (const T *)v, (Ac *)u, row_hi,
                           (uint32_t)A->num_cols, nbr, row_lo);
    } else if ((size_t)A->values_extent() * sizeof(T) >= (1ull << 32) || (size_t)A->num_cols * sizeof(T) >= (1ull << 32)) {
        // buffer descriptors address 4 GiB; beyond that fall back to the pointer-based block-row kernel
        hipLaunchKernelGGL((spmv_blockrow_kernel<T, 64>), dim3((nbr + 3) / 4), dim3(kThreads), 0, st, A->rowptr, A->keys, A->bmps,
                           A->offsets, (const T *)A->values, (const T *)v, (Ac *)u, row_hi,
                           (uint32_t)A->num_cols, nbr, row_lo);
    } else {
        build_plan(A, st);
        // dense tiles and no hub block-row: the row-group kernel streams the value array with 16-byte loads (variant 3); the 16-byte
        // value loads may run 12 bytes past the last stored value: arrays from this library's allocator carry that slack
        const bool dense_tiles = A->block_num > 0 && A->nnz >= 16 * A->block_num;
        if ((variant == 3 || (variant == BMSP_SPMV_DEFAULT && dense_tiles && A->spmv_plan_long == 0 && !getenv("BMSP_SPMV_NO_ROWGROUP"))) && pool_owns(A->values)) {
            constexpr int U = 3;
            const uint32_t groups = (nbr + 3) / 4;
            const char *pe = getenv("BMSP_SPMV_PASSES");
            const uint32_t passes = pe ? (uint32_t)std::max(1, atoi(pe)) : (groups > 32768 ? 2u : 1u);
            const uint32_t waves = (groups + passes - 1) / passes;
            hipLaunchKernelGGL((spmv_rowgroup_kernel<T, U>), dim3((waves + 3) / 4), dim3(kThreads), 0, st, A->rowptr, A->keys, A->bmps, A->offsets,
                               (const T *)A->values, (const T *)v, (Ac *)u, row_hi, (uint32_t)A->num_cols, nbr,
                               (uint32_t)((size_t)A->values_extent() * sizeof(T)) + 16u, passes, row_lo);
            BMSP_CHECK_LAUNCH();
            return;
        }
        const uint32_t n_items = (uint32_t)A->spmv_num_chunks;
        char *mem = (char *)A->spmv_chunks;
        if (vstream_eligible(A)) {
            build_pos_cache(A, st);
            const bool cached = A->spmv_pos != nullptr;
            const char *re = getenv("BMSP_SPMV_RED");
            const int red = re ? atoi(re) : (A->nnz < 2 * A->block_num ? kAtomic : kSorted);
#define BMSP_VS_LAUNCH(MODE, RED)                                                                                                                \
    hipLaunchKernelGGL((spmv_vstream_kernel<T, MODE, RED>), dim3(n_items), dim3(64), 0, st, (const SweepItem *)(mem + 64), n_items, A->keys, A->bmps, \
                       A->offsets, (const T *)A->values, (const T *)v, (Ac *)u, (Ac *)(mem + A->spmv_plan_off_carry),                             \
                       (uint32_t *)(mem + A->spmv_plan_off_cnt), row_hi, (uint32_t)A->num_cols,                                                    \
                       (uint32_t)((size_t)A->values_extent() * sizeof(T)), A->spmv_pos, (uint32_t)A->spmv_pos_base, (uint32_t)A->spmv_pos_count, row_lo,                     \
                       A->spmv_tinfo, A->spmv_eoff)
            if (cached && red == kAtomic) BMSP_VS_LAUNCH(kCached, kAtomic);
            else if (cached) BMSP_VS_LAUNCH(kCached, kSorted);
            else if (red == kAtomic) BMSP_VS_LAUNCH(kDecode, kAtomic);
            else BMSP_VS_LAUNCH(kDecode, kSorted);
            BMSP_CHECK_LAUNCH();
            return;
        }
        const bool nt = getenv("BMSP_SPMV_NT") != nullptr;
        const char *pers = getenv("BMSP_SPMV_PERSIST");
        const char *occ = getenv("BMSP_SPMV_OCC");
        auto kern = A->spmv_full_tiles * 4 >= A->block_num ? spmv_sweep_kernel<T, true, false> : (nt ? spmv_sweep_kernel<T, false, true> : (pers ? spmv_sweep_kernel<T, false, false, true> : spmv_sweep_kernel<T, false, false>));
        if (occ && atoi(occ) == 6) kern = spmv_sweep_kernel<T, false, false, false, 6>;
        if (occ && atoi(occ) == 8) kern = spmv_sweep_kernel<T, false, false, false, 8>;
        const uint32_t grid = pers ? std::min<uint32_t>((n_items + 3) / 4, (uint32_t)atoi(pers)) : (n_items + 3) / 4;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, st, (const SweepItem *)(mem + 64), n_items,
                           A->keys, A->bmps, A->offsets, (const T *)A->values, (const T *)v, (Ac *)u,
                           (Ac *)(mem + A->spmv_plan_off_carry), (uint32_t *)(mem + A->spmv_plan_off_cnt), row_hi,
                           (uint32_t)A->num_cols, (uint32_t)((size_t)A->values_extent() * sizeof(T)), row_lo);
    }
    BMSP_CHECK_LAUNCH();
}

}  // namespace

// Which kernel bmsp_spmv launches for (A, variant) -- the launcher's own decisions, in its order -- and the bytes that kernel's layout
// must move per launch ("compulsory": every array it reads or writes, once; counted from the plan, not estimated):
//   value-stream, position + tile cache: items (32 B) + slot word (4 B per tile) + for items that are NOT a single batch the 2-byte value
//     end of every tile + entries (2 B) and values per stored value + x + y + carry slots;
//   value-stream, in-kernel decode: items + keys, bitmaps, offsets (24 B per tile) + values + x + y;
//   row-group / block-row kernels: block-row pointer + 24 B per tile + values + x + y;   sweep: items + 24 B per tile + values + x + y.
// format_bytes is SURVEY 8(d)'s figure for the bmSparse layout whatever the kernel reads: 24 B per tile + values + row pointer + x + y.
void spmv_launch_info(bmsp_matrix_s *A, int variant, hipStream_t st, char *kernel, size_t kernel_cap, int64_t *compulsory, int64_t *format_bytes)
{
    if (A->transposed) fail(BMSP_ERR_INVALID, "SpMV needs a matrix built with transposed=0");
    ensure_rowptr(A, st);
    const int64_t es = (int64_t)dtype_size(A->dtype), as = A->dtype == BMSP_F64 ? 8 : 4;
    const int64_t nbr = A->num_block_rows(), nb = A->block_num, nv = A->nnz;
    const int64_t xy = es * A->num_cols + as * A->num_rows;
    const int64_t fmt = 24 * nb + es * nv + 4 * (nbr + 1) + xy;
    if (format_bytes) *format_bytes = fmt;
    const char *name = "";
    int64_t bytes = fmt;
    const bool wide = (size_t)A->values_extent() * (size_t)es >= (1ull << 32) || (size_t)A->num_cols * (size_t)es >= (1ull << 32);
    if (nbr == 0) { name = "none (empty matrix)"; bytes = 0; }
    else if (variant == BMSP_SPMV_BATCHED || wide) name = "spmv_blockrow_kernel<64 lanes per block-row>";
    else if (variant == 2) name = "spmv_blockrow_kernel<8 lanes per block-row>";
    else {
        build_plan(A, st);
        const bool dense_tiles = nb > 0 && nv >= 16 * nb;
        if ((variant == 3 || (variant == BMSP_SPMV_DEFAULT && dense_tiles && A->spmv_plan_long == 0 && !getenv("BMSP_SPMV_NO_ROWGROUP"))) && pool_owns(A->values)) {
            name = "spmv_rowgroup_kernel";
        } else {
            const int64_t n_items = A->spmv_num_chunks;
            if (vstream_eligible(A)) {
                build_pos_cache(A, st);
                const bool cached = A->spmv_pos != nullptr;
                const char *re = getenv("BMSP_SPMV_RED");
                const int red = re ? atoi(re) : (nv < 2 * nb ? kAtomic : kSorted);
                name = cached ? (red == kAtomic ? "spmv_vstream_kernel<kCached, kAtomic>" : "spmv_vstream_kernel<kCached, kSorted>")
                              : (red == kAtomic ? "spmv_vstream_kernel<kDecode, kAtomic>" : "spmv_vstream_kernel<kDecode, kSorted>");
                std::vector<SweepItem> items((size_t)n_items);
                BMSP_HIP(hipStreamSynchronize(st));
                if (n_items) copy_d2h_staged(items.data(), plan_items(A), sizeof(SweepItem) * (size_t)n_items);
                int64_t multi_tiles = 0, long_items = 0;
                for (const SweepItem &it : items) {
                    if (!vs_single(it)) multi_tiles += (int64_t)(it.blk_end - it.blk_begin);
                    if (it.num_items) long_items++;
                }
                bytes = 32 * n_items + xy + 64 * long_items + es * nv;
                bytes += cached ? 4 * nb + 2 * multi_tiles + 2 * nv : 24 * nb;
            } else {
                name = A->spmv_full_tiles * 4 >= nb ? "spmv_sweep_kernel<FULL>" : "spmv_sweep_kernel";
                bytes = 32 * n_items + 24 * nb + es * nv + xy;
            }
        }
    }
    if (kernel && kernel_cap) snprintf(kernel, kernel_cap, "%s", name);
    if (compulsory) *compulsory = bytes;
}

void prepare_spmv(bmsp_matrix_s *A, hipStream_t st)
{
    if (A->transposed || A->num_block_rows() == 0) return;
    const size_t es = dtype_size(A->dtype);
    if ((size_t)A->values_extent() * es >= (1ull << 32) || (size_t)A->num_cols * es >= (1ull << 32)) return;  // block-row kernel: no plan
    build_plan(A, st);
    const bool rowgroup = A->block_num > 0 && A->nnz >= 16 * A->block_num && A->spmv_plan_long == 0 && !getenv("BMSP_SPMV_NO_ROWGROUP") && pool_owns(A->values);
    if (vstream_eligible(A) && !rowgroup) build_pos_cache(A, st);
}

// row_lo / row_hi: only rows [row_lo, row_hi) of u are written (the sharded sweep: a rank writes its own slice and nothing else);
// the default is every row
void spmv(bmsp_matrix_s *A, const void *v, void *u, int variant, hipStream_t st, int64_t row_lo, int64_t row_hi)
{
    if (A->transposed) fail(BMSP_ERR_INVALID, "SpMV needs a matrix built with transposed=0");
    if (!v || !u) fail(BMSP_ERR_INVALID, "null vector");
    if (variant < 0 || variant > 3) fail(BMSP_ERR_INVALID, "unknown SpMV variant %d", variant);
    ensure_rowptr(A, st);
    const uint32_t lo = (uint32_t)std::max<int64_t>(0, row_lo);
    const uint32_t hi = (uint32_t)(row_hi < 0 ? (int64_t)A->num_rows : std::min<int64_t>(row_hi, A->num_rows));
    switch (A->dtype) {
    case BMSP_F32: launch<float>(A, v, u, variant, st, lo, hi); break;
    case BMSP_F16: launch<_Float16>(A, v, u, variant, st, lo, hi); break;
    case BMSP_F64: launch<double>(A, v, u, variant, st, lo, hi); break;
    }
}

}  // namespace bmsp

BMSP_DEFINE_WARM(spmv)
