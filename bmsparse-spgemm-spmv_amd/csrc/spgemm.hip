// spgemm.hip -- C = A * B on the bmSparse format: block-pair expansion, bitmap filter, sort of the pairs by
// their (row, col) block key, segmented compress into C's layout, 8x8x8 block multiply-accumulate.
//
// Reference: bmSparse_mult<VI,VO>, src/bmSparse_SPGEMM.cu:827-1223 (stages T_1..T_9, T_7) and the block-MAC
// kernels multiplyV15 (:204-291) / multiplyV11..V14 (:294-733).  Stage names below are the reference's.
//
// Differences by design (MI355X-first, same results):
//   * tasks are 8 bytes (A block index, B block index as two uint32), not 16, and the unfiltered task list is
//     never written to HBM: expansion and the bitmap filter (T_3 + T_4) are one load-balanced pass that is run
//     twice (count, then write) instead of expand + remove_if;
//   * the bitmap filter is two byte-OR reductions and an AND (tile_product_empty), not a 64-iteration loop;
//   * the sort key is the packed (block_row, block_col) of C restricted to the bits in use; the sort is stable,
//     so tasks of one C block stay in k-ascending order (the reference's order is unspecified above BORDER);
//   * B's dense block-row pointer is cached on the matrix (T_1 is free after the first product).
#include "matrix.h"
#include "prims.hip.h"
#include "mac_common.hip.h"
#include <memory>
#include <string>

namespace bmsp {
namespace {

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, kWave));
    return v;
}

// ---- T_2 / T_3: fan-out of every A block = number of B blocks in block-row col(a) ------------------------
struct FanOut {
    const uint64_t *a_keys;
    const uint32_t *b_rowptr;
    uint64_t n_a;
    uint32_t b_block_rows;
    __device__ uint64_t operator()(uint64_t a) const
    {
        if (a >= n_a) return 0;
        uint32_t col = key_col(a_keys[a]);
        if (col >= b_block_rows) return 0;  // no matching block-row in B (reference indexes out of bounds here)
        return (uint64_t)(b_rowptr[col + 1] - b_rowptr[col]);
    }
};

// ---- T_3 + T_4: load-balanced expansion with the bitmap filter -------------------------------------------
// A workgroup owns kTile consecutive candidate pairs.  The A blocks that generate them form a contiguous range
// found by two binary searches; their first-candidate positions are staged in LDS and every candidate finds
// its A block by a binary search there.
constexpr int kSpanMax = kTile;  // staged first_pos entries

struct ExpandArgs {
    const uint64_t *first_pos;  // n_a + 1, exclusive scan of the fan-out
    const uint64_t *a_keys, *a_bmps, *b_keys, *b_bmps;
    const uint32_t *b_rowptr;
    uint64_t n_a;
    uint64_t total;  // candidates
    int jbits;       // bits of C's block column in the sort key
};

__device__ __forceinline__ uint32_t upper_bound_u64(const uint64_t *p, uint32_t lo, uint32_t hi, uint64_t v)
{  // first index in [lo,hi) with p[idx] > v
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (p[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// ITEMS candidates per thread: 16 (4096 per workgroup) on large products; 2 on small ones, where a tile's 16 dependent rounds
// would run on a handful of workgroups (the banded bench case: 28 workgroups x 31 us -> 223 x 8 us)
template <bool WRITE, int ITEMS>
__global__ __launch_bounds__(kThreads) void expand_filter_kernel(ExpandArgs g, uint32_t *__restrict__ tile_counts,
                                                                 const uint32_t *__restrict__ tile_base, uint64_t *__restrict__ out_keys,
                                                                 uint64_t *__restrict__ out_tasks)
{
    __shared__ uint32_t rel[kSpanMax + 1];
    __shared__ uint32_t range[2];
    __shared__ uint32_t lds4[4];
    constexpr uint32_t kTileE = (uint32_t)ITEMS * kThreads;
    const uint64_t t0 = (uint64_t)blockIdx.x * kTileE;
    const uint64_t t1 = (t0 + (uint64_t)kTileE < g.total) ? t0 + (uint64_t)kTileE : g.total;
    if (threadIdx.x == 0) range[0] = upper_bound_u64(g.first_pos, 0, (uint32_t)g.n_a + 1, t0) - 1;
    if (threadIdx.x == 64) range[1] = upper_bound_u64(g.first_pos, 0, (uint32_t)g.n_a + 1, t1 - 1) - 1;
    __syncthreads();
    const uint32_t a_lo = range[0], a_hi = range[1];
    const uint32_t span = a_hi - a_lo + 1;
    const bool staged = span <= (uint32_t)kSpanMax;
    if (staged) {
        for (uint32_t k = threadIdx.x; k <= span; k += kThreads) {
            uint64_t fp = g.first_pos[a_lo + k];
            rel[k] = fp <= t0 ? 0u : (fp - t0 < (uint64_t)kTileE ? (uint32_t)(fp - t0) : kTileE);
        }
    }
    __syncthreads();
    uint32_t carry = WRITE ? tile_base[blockIdx.x] : 0u;
    uint32_t kept_total = 0;
    for (int k = 0; k < ITEMS; k++) {
        uint64_t t = t0 + (uint64_t)k * kThreads + threadIdx.x;
        bool keep = false;
        uint32_t a = 0, b = 0;
        if (t < t1) {
            uint32_t lt = (uint32_t)(t - t0);
            if (staged) {
                // last staged block whose first candidate is <= lt
                uint32_t lo = 0, hi = span;
                while (lo < hi) {
                    uint32_t mid = lo + ((hi - lo + 1) >> 1);
                    if (rel[mid] <= lt) lo = mid;
                    else hi = mid - 1;
                }
                // (blocks with zero fan-out share a position with their successor; "last index <=" picks the
                // owner, which is the last of such a run)
                a = a_lo + lo;
            } else {
                a = upper_bound_u64(g.first_pos, a_lo, a_hi + 1, t) - 1;
            }
            uint64_t akey = g.a_keys[a];
            b = g.b_rowptr[key_col(akey)] + (uint32_t)(t - g.first_pos[a]);
            keep = !tile_product_empty(g.a_bmps[a], g.b_bmps[b]);  // multiplication_checker (:742-757)
        }
        if (WRITE) {
            uint32_t total;
            uint32_t ex = block_exclusive_sum<uint32_t>(keep ? 1u : 0u, lds4, total);
            if (keep) {
                uint64_t akey = g.a_keys[a];
                uint64_t ck = ((uint64_t)key_row(akey) << g.jbits) | (uint64_t)key_col(g.b_keys[b]);
                out_keys[carry + ex] = ck;
                out_tasks[carry + ex] = ((uint64_t)a << 32) | (uint64_t)b;
            }
            carry += total;
        } else {
            kept_total += keep ? 1u : 0u;
        }
    }
    if (!WRITE) {
        kept_total = wave_sum(kept_total);
        if (lane_id() == 0) lds4[wave_id()] = kept_total;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    }
}

// ---- T_3 + T_4 in ONE pass: decoupled look-back over the tiles -------------------------------------------------------------
// The count + write pair above evaluates every candidate twice.  Here a workgroup evaluates its tile once (the surviving pairs stay
// in registers), publishes the tile's survivor count in a status word {flag, count}, takes the sum of all earlier tiles by looking
// back over their status words (aggregate / inclusive-prefix protocol), and writes its survivors at that position.  Tile ids are
// handed out by an atomic counter, so a workgroup only ever waits for tiles that have already started.  Each status word is ONE
// 8-byte agent-scope atomic store / load ({flag, value} in a single granule: nothing else has to be ordered around it).  The
// output is allocated for the candidate count (the upper bound) instead of the survivor count.
constexpr uint64_t kTileAgg = 1ull << 62, kTilePrefix = 2ull << 62;

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void expand_filter_lookback_kernel(ExpandArgs g, unsigned long long *__restrict__ tile_state, uint32_t *__restrict__ tile_counter,
                                                                          uint32_t num_tiles, uint64_t *__restrict__ out_keys, uint64_t *__restrict__ out_tasks,
                                                                          uint32_t *__restrict__ n_tasks_host)
{
    __shared__ uint32_t rel[kSpanMax + 1];
    __shared__ uint32_t range[2];
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t s_tile, s_prefix;
    constexpr uint32_t kTileE = (uint32_t)ITEMS * kThreads;
    if (threadIdx.x == 0) s_tile = atomicAdd(tile_counter, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t t0 = (uint64_t)tile * kTileE;
    const uint64_t t1 = (t0 + (uint64_t)kTileE < g.total) ? t0 + (uint64_t)kTileE : g.total;
    if (threadIdx.x == 0) range[0] = upper_bound_u64(g.first_pos, 0, (uint32_t)g.n_a + 1, t0) - 1;
    if (threadIdx.x == 64) range[1] = upper_bound_u64(g.first_pos, 0, (uint32_t)g.n_a + 1, t1 - 1) - 1;
    __syncthreads();
    const uint32_t a_lo = range[0], a_hi = range[1];
    const uint32_t span = a_hi - a_lo + 1;
    const bool staged = span <= (uint32_t)kSpanMax;
    if (staged) {
        for (uint32_t k = threadIdx.x; k <= span; k += kThreads) {
            uint64_t fp = g.first_pos[a_lo + k];
            rel[k] = fp <= t0 ? 0u : (fp - t0 < (uint64_t)kTileE ? (uint32_t)(fp - t0) : kTileE);
        }
    }
    __syncthreads();
    // evaluate the tile once: surviving pairs stay in registers (~0 = dropped)
    uint64_t pair[ITEMS];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint64_t t = t0 + (uint64_t)k * kThreads + threadIdx.x;
        pair[k] = ~0ull;
        if (t < t1) {
            const uint32_t lt = (uint32_t)(t - t0);
            uint32_t a;
            if (staged) {
                uint32_t lo = 0, hi = span;
                while (lo < hi) {
                    uint32_t mid = lo + ((hi - lo + 1) >> 1);
                    if (rel[mid] <= lt) lo = mid;
                    else hi = mid - 1;
                }
                a = a_lo + lo;
            } else {
                a = upper_bound_u64(g.first_pos, a_lo, a_hi + 1, t) - 1;
            }
            const uint32_t b = g.b_rowptr[key_col(g.a_keys[a])] + (uint32_t)(t - g.first_pos[a]);
            if (!tile_product_empty(g.a_bmps[a], g.b_bmps[b])) {  // multiplication_checker (:742-757)
                pair[k] = ((uint64_t)a << 32) | (uint64_t)b;
                mine++;
            }
        }
    }
    // tile total -> status word -> look back
    const uint32_t wsum = wave_sum(mine);
    if (lane_id() == 0) lds4[wave_id()] = wsum;
    __syncthreads();
    const uint32_t tile_total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    if (wave_id() == 0) {
        const int lane = lane_id();
        if (lane == 0)
            __hip_atomic_store(&tile_state[tile], (tile == 0 ? kTilePrefix : kTileAgg) | (uint64_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t prefix = 0;
        int64_t look = (int64_t)tile - 1;  // newest tile not yet accounted for
        while (look >= 0) {
            const int64_t idx = look - lane;
            unsigned long long w = kTilePrefix;  // lanes before tile 0 read as "prefix 0"
            if (idx >= 0) {
                do {
                    w = __hip_atomic_load(&tile_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(w >> 62)) __builtin_amdgcn_s_sleep(1);
                } while (!(w >> 62));
            }
            const uint64_t is_prefix = __ballot((w >> 62) == 2ull);
            // lanes up to and including the first inclusive prefix count
            const int stop = is_prefix ? __builtin_ctzll(is_prefix) : 63;
            uint32_t v = lane <= stop ? (uint32_t)w : 0u;
            prefix += wave_sum(v);
            if (is_prefix) break;
            look -= 64;
        }
        if (lane == 0) {
            s_prefix = prefix;
            if (tile != 0)
                __hip_atomic_store(&tile_state[tile], kTilePrefix | (uint64_t)(prefix + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tile == num_tiles - 1) *n_tasks_host = prefix + tile_total;
        }
    }
    __syncthreads();
    // write the survivors in candidate order
    uint32_t carry = s_prefix;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool keep = pair[k] != ~0ull;
        uint32_t total;
        const uint32_t ex = block_exclusive_sum<uint32_t>(keep ? 1u : 0u, lds4, total);
        if (keep) {
            const uint32_t a = (uint32_t)(pair[k] >> 32), b = (uint32_t)pair[k];
            out_keys[carry + ex] = ((uint64_t)key_row(g.a_keys[a]) << g.jbits) | (uint64_t)key_col(g.b_keys[b]);
            out_tasks[carry + ex] = pair[k];
        }
        carry += total;
    }
}

struct CountIn {
    const uint32_t *p;
    uint64_t n;
    __device__ uint32_t operator()(uint64_t i) const { return i < n ? p[i] : 0u; }
};

// ---- T_6: run-length encode the sorted C keys -----------------------------------------------------------------
struct KeyHead {
    const uint64_t *sk;
    uint64_t n;
    __device__ uint32_t operator()(uint64_t t) const
    {
        if (t >= n) return 0;
        return (t == 0 || sk[t] != sk[t - 1]) ? 1u : 0u;
    }
};
struct EmitCBlocks {
    const uint64_t *sk;
    uint64_t n;
    int jbits;
    uint64_t *c_keys;
    uint32_t *task_begin;
    uint32_t *c_of_wave;  // C block of task 64 w, for the task-parallel bitmap pass (T_9)
    uint32_t *c_size;
    __device__ void operator()(uint64_t t, uint32_t ex, uint32_t is_head) const
    {
        if (t == n) {
            task_begin[ex] = (uint32_t)n;
            *c_size = ex;
            return;
        }
        const bool head = is_head != 0;  // KeyHead's value for this task
        if (head) {
            uint64_t k = sk[t];
            c_keys[ex] = key_make((uint32_t)(k >> jbits), (uint32_t)(k & ((1ull << jbits) - 1ull)));
            task_begin[ex] = (uint32_t)t;
        }
        if ((t & 63u) == 0) c_of_wave[t >> 6] = head ? ex : ex - 1u;
    }
};

// ---- T_9: bitmap of every C block = OR of the boolean products of its tasks ---------------------------------
// Task-parallel: a wave takes 64 consecutive tasks of the sorted list (coalesced reads, two bitmap gathers per lane),
// ORs the products of equal-key runs with a segmented scan across lanes, and the last lane of every run writes the C
// bitmap -- with a plain store when the run lies inside the wave, with an atomic OR when it crosses a wave boundary
// (hub C blocks with thousands of tasks become one atomic per wave instead of one serial loop).  c_bmps starts zeroed.
__global__ __launch_bounds__(kThreads) void c_bitmaps_kernel(const uint64_t *__restrict__ sk, const uint64_t *__restrict__ tasks, uint32_t n,
                                                             const uint32_t *__restrict__ c_of_wave, const uint64_t *__restrict__ a_bmps,
                                                             const uint64_t *__restrict__ b_bmps, unsigned long long *__restrict__ c_bmps)
{
    const int lane = lane_id();
    const uint32_t wv = blockIdx.x * 4 + wave_id();
    const uint32_t t = wv * 64u + (uint32_t)lane;
    if (wv * 64u >= n) return;
    const bool valid = t < n;
    const uint64_t key = valid ? sk[t] : ~0ull;
    uint64_t p = 0;
    if (valid) {
        const uint64_t tk = tasks[t];
        p = tile_product_bmp(a_bmps[tk >> 32], b_bmps[(uint32_t)tk]);  // bmp_calculator (:787-810)
    }
    // run boundaries from the neighbours' keys (the wave's edges look across to the adjacent tasks)
    uint64_t prev = __shfl_up(key, 1, kWave), next = __shfl_down(key, 1, kWave);
    if (lane == 0) prev = t > 0 ? sk[t - 1] : ~key;
    if (lane == 63) next = (valid && t + 1 < n) ? sk[t + 1] : ~key;
    const bool head = valid && key != prev, tail = valid && key != next;
    const uint64_t head_mask = __ballot(head);
    const uint64_t le = lanemask_lt() | (1ull << lane);
    // first lane of this lane's run inside the wave (lane 0 when the run began in an earlier wave)
    const int start = 63 - __clzll((long long)((head_mask | 1ull) & le));
    // segmented inclusive OR: lane l collects lanes [start, l]
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t q = __shfl_up(p, d, kWave);
        if (lane - d >= start) p |= q;
    }
    if (valid && (tail || lane == 63)) {
        const uint32_t c = c_of_wave[wv] + (uint32_t)__popcll(head_mask & le & ~1ull);
        // the whole run lies inside this wave: nobody else writes this C block
        if (tail && ((head_mask >> start) & 1ull)) c_bmps[c] = p;
        else atomicOr(&c_bmps[c], (unsigned long long)p);
    }
}

struct ZeroU64 {
    uint64_t *p;
    __device__ void operator()(uint64_t i) const { p[i] = 0ull; }
};
struct PopcIn {
    const uint64_t *bmps;
    uint64_t n;
    __device__ uint64_t operator()(uint64_t i) const { return i < n ? (uint64_t)__popcll(bmps[i]) : 0ull; }
};

// ---- T_7: block multiply-accumulate, vector-ALU kernel with the reference's V15 numerics ---------------------
// One wave per C block; lane l owns C(l/8, l%8).  Per task the wave expands the two tiles into LDS (lane l
// fetches tile position l, zero when its bit is clear) and runs the 8-step inner product.  fp16 inputs: every
// product is rounded to fp16 before the fp32 add, exactly as `__half * __half` does in multiplyV15 (:269-273).
template <typename T>
struct MacOps;
template <>
struct MacOps<float> {
    using Out = float;
    static __device__ __forceinline__ float step(float a, float b, float acc) { return __builtin_fmaf(a, b, acc); }
};
template <>
struct MacOps<_Float16> {
    using Out = float;
    static __device__ __forceinline__ float step(_Float16 a, _Float16 b, float acc)
    {
        _Float16 p = a * b;  // rounded to fp16
        return acc + (float)p;
    }
};
template <>
struct MacOps<double> {
    using Out = double;
    static __device__ __forceinline__ double step(double a, double b, double acc) { return __builtin_fma(a, b, acc); }
};

template <typename T>
__global__ __launch_bounds__(kThreads) void block_mac_valu_kernel(const uint64_t *__restrict__ tasks, const uint32_t *__restrict__ task_begin,
                                                                  const uint64_t *__restrict__ a_bmps, const uint64_t *__restrict__ a_offs,
                                                                  const T *__restrict__ a_vals, const uint64_t *__restrict__ b_bmps,
                                                                  const uint64_t *__restrict__ b_offs, const T *__restrict__ b_vals,
                                                                  const uint64_t *__restrict__ c_bmps, const uint64_t *__restrict__ c_offs,
                                                                  typename MacOps<T>::Out *__restrict__ c_vals, uint32_t c_size)
{
    using O = typename MacOps<T>::Out;
    __shared__ T tile_a[4][64];
    __shared__ T tile_b[4][64];
    const int w = wave_id(), lane = lane_id();
    const int i = lane >> 3, j = lane & 7;
    for (uint32_t c = blockIdx.x * 4 + w; c < c_size; c += gridDim.x * 4) {
        const uint32_t tb = task_begin[c], te = task_begin[c + 1];
        O acc = 0;
        for (uint32_t t = tb; t < te; t++) {
            const uint64_t tk = tasks[t];
            const uint32_t a = (uint32_t)(tk >> 32), b = (uint32_t)tk;
            const uint64_t bmp_a = a_bmps[a], bmp_b = b_bmps[b];
            T av = T(0), bv = T(0);
            if (tile_has(bmp_a, lane)) av = a_vals[a_offs[a] + tile_rank(bmp_a, lane)];
            if (tile_has(bmp_b, lane)) bv = b_vals[b_offs[b] + tile_rank(bmp_b, lane)];
            __builtin_amdgcn_wave_barrier();
            tile_a[w][lane] = av;  // A(i,k) at i*8+k
            tile_b[w][lane] = bv;  // B stored column-major in the tile: B(k,j) at j*8+k
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; k++) acc = MacOps<T>::step(tile_a[w][i * 8 + k], tile_b[w][j * 8 + k], acc);
        }
        const uint64_t bmp_c = c_bmps[c];
        if (tile_has(bmp_c, lane)) c_vals[c_offs[c] + tile_rank(bmp_c, lane)] = acc;
    }
}

// ---- T_7: block multiply-accumulate on the matrix cores (fp16 inputs, fp32 accumulate) ---------------------
// v_mfma_f32_16x16x16_f16 computes a 16x16x16 product per wave.  Two C blocks are packed block-diagonally
// (block 0 -> rows/cols 0-7, block 1 -> rows/cols 8-15) and two tasks of each block are packed along K
// (k 0-7 and 8-15): four 8x8x8 block products per instruction (the packing idea of multiplyV14, :294-417).
// Operand layout (gfx950): lane l supplies A[row = l&15][k = 4*(l>>4) .. +3] and B[k = 4*(l>>4) .. +3][col = l&15];
// result lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3.  Every lane therefore needs FOUR CONSECUTIVE
// k of one tile row (A, row-major tile) or of one tile column (B, column-major tile): one nibble of the bitmap
// and up to four consecutive stored values -- straight from global memory, no LDS staging, no shuffles.
constexpr int kPairsPerWave = 4;  // C-block pairs a wave works on at once (independent load chains in flight)

__global__ __launch_bounds__(kThreads) void block_mac_mfma_f16_kernel(const uint64_t *__restrict__ tasks, const uint32_t *__restrict__ task_begin,
                                                                      const uint64_t *__restrict__ a_bmps, const uint64_t *__restrict__ a_offs,
                                                                      const _Float16 *__restrict__ a_vals, const uint64_t *__restrict__ b_bmps,
                                                                      const uint64_t *__restrict__ b_offs, const _Float16 *__restrict__ b_vals,
                                                                      const uint64_t *__restrict__ c_bmps, const uint64_t *__restrict__ c_offs,
                                                                      float *__restrict__ c_vals, uint32_t c_size, uint32_t a_bytes, uint32_t b_bytes)
{
    const int w = wave_id(), lane = lane_id();
    const int line = lane & 7;          // tile row (A operand) / tile column (B operand)
    const int which = (lane >> 3) & 1;  // which of the two packed C blocks this lane feeds (rows/cols 8-15 -> 1)
    const int kq = lane >> 4;           // k quarter: 0,1 -> first task of the pair, 2,3 -> second task
    const int slot = kq >> 1, khalf = kq & 1;
    const rsrc_t ra = make_rsrc(a_vals, a_bytes), rb = make_rsrc(b_vals, b_bytes);
    const uint32_t pairs = (c_size + 1) / 2;
    const uint32_t groups = (pairs + kPairsPerWave - 1) / kPairsPerWave;
    for (uint32_t g = blockIdx.x * 4 + w; g < groups; g += gridDim.x * 4) {
        uint32_t tb[kPairsPerWave], te[kPairsPerWave];
        uint32_t steps = 0;
#pragma unroll
        for (int p = 0; p < kPairsPerWave; p++) {
            const uint32_t c = (g * kPairsPerWave + p) * 2 + which;  // the C block this lane loads operands for
            tb[p] = 0; te[p] = 0;
            if (c < c_size) { tb[p] = task_begin[c]; te[p] = task_begin[c + 1]; }
            steps = max(steps, te[p] - tb[p]);
        }
        // both blocks of every pair advance two tasks per step; the wave runs until the longest list is exhausted
        steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(steps));
        steps = (steps + 1) / 2;
        float4_t acc[kPairsPerWave];
#pragma unroll
        for (int p = 0; p < kPairsPerWave; p++) acc[p] = float4_t{0.f, 0.f, 0.f, 0.f};
        for (uint32_t s = 0; s < steps; s++) {
            half4_t fa[kPairsPerWave], fb[kPairsPerWave];
            uint64_t tk[kPairsPerWave];
            bool live[kPairsPerWave];
#pragma unroll
            for (int p = 0; p < kPairsPerWave; p++) {
                const uint32_t t = tb[p] + 2 * s + slot;
                live[p] = t < te[p];
                tk[p] = live[p] ? tasks[t] : 0ull;
            }
#pragma unroll
            for (int p = 0; p < kPairsPerWave; p++) {
                const uint32_t a = (uint32_t)(tk[p] >> 32), b = (uint32_t)tk[p];
                fa[p] = load_nibble(a_bmps[a], ra, (uint32_t)a_offs[a] * 2u, line, khalf, live[p]);
                fb[p] = load_nibble(b_bmps[b], rb, (uint32_t)b_offs[b] * 2u, line, khalf, live[p]);
            }
#pragma unroll
            for (int p = 0; p < kPairsPerWave; p++) acc[p] = __builtin_amdgcn_mfma_f32_16x16x16f16(fa[p], fb[p], acc[p], 0, 0, 0);
        }
        // result: lane holds D[4*(lane>>4)+r][lane&15]; block 0 lives in rows 0-7 x cols 0-7, block 1 in 8-15 x 8-15
        const int col = lane & 15, rq = lane >> 4;
        const int blk = col >> 3;
        if ((rq >> 1) == blk) {
#pragma unroll
            for (int p = 0; p < kPairsPerWave; p++) {
                const uint32_t cc = (g * kPairsPerWave + p) * 2 + blk;
                if (cc < c_size) {
                    const uint64_t bmp_c = c_bmps[cc];
                    const uint64_t off = c_offs[cc];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int pos = ((rq & 1) * 4 + r) * 8 + (col & 7);
                        if (tile_has(bmp_c, pos)) c_vals[off + tile_rank(bmp_c, pos)] = acc[p][r];
                    }
                }
            }
        }
    }
}

// ---- T_7, task-parallel operand fetch, software-pipelined over C-block groups ------------------------------------
// The kernel above walks each C block's task list with the whole wave, so a wave has one dependent chain
// (task_begin -> task -> bitmaps/offsets -> values) per C-block pair in flight and the stage is bound by memory round trips
// (profiles/r01c_spgemm_cage_like_summary.md: 59 % of wave cycles parked on s_waitcnt at 4.5 waves/SIMD).  Here a wave
// owns kGroupC consecutive C blocks (kGroupC/2 MFMA pairs) per step and turns the chain sideways:
//   * the tasks of those blocks are one contiguous range of the sorted task list, so 64 LANES fetch 64 tasks and their
//     four bitmap/offset words at once and park them in LDS; the MFMA operand lanes then pick their task's words from
//     LDS and issue the value gathers of four pairs back to back;
//   * the four levels of the chain belong to four DIFFERENT groups in any one iteration (group k: task_begin, k-1: tasks,
//     k-2: bitmaps/offsets -> LDS, k-3: values + MFMA + store), so each iteration waits for one round trip, not four.
// A group with more than 64 tasks (hub C blocks) streams its remaining 64-task windows through the same LDS slot unpipelined.

template <int kGroupC>
struct MacMeta {
    uint64_t abmp[64], bbmp[64], cbmp[kGroupC], coff[kGroupC];
    uint32_t aoff[64], boff[64];
};

// a_meta / b_meta: the matrices' packed block records {bitmap lo, bitmap hi, value offset, 0} (matrix.h: block_meta) -- one
// 16-byte gather per operand block instead of a bitmap gather and an offset gather
#define BMSP_MAC_ARGS                                                                                                              \
    const uint64_t *__restrict__ tasks, const uint32_t *__restrict__ task_begin, const uint32_t *__restrict__ a_meta,                  \
        uint32_t a_meta_bytes, const _Float16 *__restrict__ a_vals, const uint32_t *__restrict__ b_meta, uint32_t b_meta_bytes,        \
        const _Float16 *__restrict__ b_vals, const uint64_t *__restrict__ c_bmps, const uint64_t *__restrict__ c_offs,                 \
        float *__restrict__ c_vals, uint32_t c_size, uint32_t a_bytes, uint32_t b_bytes
#define BMSP_MAC_PASS tasks, task_begin, a_meta, a_meta_bytes, a_vals, b_meta, b_meta_bytes, b_vals, c_bmps, c_offs, c_vals, c_size, a_bytes, b_bytes

template <int kGroupC>
__device__ __forceinline__ void block_mac_group_body(BMSP_MAC_ARGS)
{
    typedef MacMeta<kGroupC> MacMeta;
    constexpr int kPairs = kGroupC / 2, kBatch = kPairs < 4 ? kPairs : 4;  // pairs whose value gathers are issued back to back
    __shared__ MacMeta s_meta[4][2];
    __shared__ float s_out[4][kGroupC][64];  // finished C tiles, position-major, before the compacting store
    __shared__ uint64_t s_sel[16];
    const int w = wave_id(), lane = lane_id();
    const int line = lane & 7;          // tile row (A operand) / tile column (B operand)
    const int which = (lane >> 3) & 1;  // which C block of the pair this lane feeds
    const int kq = lane >> 4;
    const int slot = kq >> 1, khalf = kq & 1;
    const NibbleLane nl = make_nibble_lane(line, khalf);
    if (threadIdx.x < 16) s_sel[threadIdx.x] = nibble_selector(threadIdx.x);
    __syncthreads();
    const rsrc_t ra = make_rsrc(a_vals, a_bytes), rb = make_rsrc(b_vals, b_bytes);
    const rsrc_t rma = make_rsrc(a_meta, a_meta_bytes), rmb = make_rsrc(b_meta, b_meta_bytes);
    const uint32_t groups = (c_size + kGroupC - 1) / kGroupC;
    const uint32_t g0 = blockIdx.x * 4 + w, stride = gridDim.x * 4;
    const uint32_t mine = g0 < groups ? (groups - g0 + stride - 1) / stride : 0u;
    // task_begin words of the groups in stages 1..3 (lane l <= 16 holds task_begin[16 g + l]); 0 = no tasks
    uint32_t tbv1 = 0, tbv2 = 0, tbv3 = 0;
    uint64_t tk2 = 0;  // the task lane `lane` fetched for the group now entering stage 2
    for (uint32_t k = 0; k < mine + 3; k++) {
        // stage 0: task_begin of group k
        uint32_t tbv0 = 0;
        if (k < mine) tbv0 = task_begin[min((g0 + k * stride) * kGroupC + (uint32_t)min(lane, kGroupC), c_size)];
        // stage 1: first 64 tasks of group k-1
        uint64_t tk1 = 0;
        {
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)tbv1, 0), te = (uint32_t)__builtin_amdgcn_readlane((int)tbv1, kGroupC);
            if (tb + lane < min(te, tb + 64u)) tk1 = tasks[tb + lane];
        }
        // stage 2: bitmaps / value offsets of those tasks for group k-2, C block bitmaps / offsets of the group
        uint64_t m_abmp = 0, m_bbmp = 0, m_cbmp = 0, m_coff = 0;
        uint32_t m_aoff = 0, m_boff = 0;
        const bool have2 = k >= 2 && k - 2 < mine;
        {
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)tbv2, 0), te = (uint32_t)__builtin_amdgcn_readlane((int)tbv2, kGroupC);
            if (tb + lane < min(te, tb + 64u)) {
                load_block_meta(rma, (uint32_t)(tk2 >> 32), m_abmp, m_aoff);
                load_block_meta(rmb, (uint32_t)tk2, m_bbmp, m_boff);
                m_aoff *= 2u; m_boff *= 2u;
            }
            if (have2 && lane < kGroupC) {
                const uint32_t cc = min((g0 + (k - 2) * stride) * kGroupC + (uint32_t)lane, c_size - 1);
                m_cbmp = c_bmps[cc]; m_coff = c_offs[cc];
            }
        }
        // stage 3: values, MFMA, store for group k-3
        if (k >= 3) {
            MacMeta &M = s_meta[w][(k - 3) & 1];
            const uint32_t c0 = (g0 + (k - 3) * stride) * kGroupC;
            uint32_t sb[kGroupC + 1];
#pragma unroll
            for (int i = 0; i <= kGroupC; i++) sb[i] = (uint32_t)__builtin_amdgcn_readlane((int)tbv3, i);
            const uint32_t tb = sb[0], te = sb[kGroupC];
            float4_t acc[kGroupC / 2];
#pragma unroll
            for (int p = 0; p < kGroupC / 2; p++) acc[p] = float4_t{0.f, 0.f, 0.f, 0.f};
            for (uint32_t lo = tb; lo < te; lo += 64) {
                const uint32_t hi = min(lo + 64u, te);
                if (lo != tb) {  // hub group: later windows, fetched in place
                    __builtin_amdgcn_wave_barrier();
                    if (lo + lane < hi) {
                        const uint64_t tk = tasks[lo + lane];
                        uint64_t ab, bb;
                        uint32_t ao, bo;
                        load_block_meta(rma, (uint32_t)(tk >> 32), ab, ao);
                        load_block_meta(rmb, (uint32_t)tk, bb, bo);
                        M.abmp[lane] = ab; M.aoff[lane] = ao * 2u;
                        M.bbmp[lane] = bb; M.boff[lane] = bo * 2u;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // the part of every C block's list that lies in this 64-task window; two tasks per MFMA step
                uint32_t steps[kGroupC / 2], max_steps = 0;
#pragma unroll
                for (int p = 0; p < kGroupC / 2; p++) {
                    const uint32_t b0 = max(sb[2 * p], lo), e0 = min(sb[2 * p + 1], hi);
                    const uint32_t b1 = max(sb[2 * p + 1], lo), e1 = min(sb[2 * p + 2], hi);
                    const uint32_t n0 = e0 > b0 ? e0 - b0 : 0u, n1 = e1 > b1 ? e1 - b1 : 0u;
                    steps[p] = (max(n0, n1) + 1) / 2;
                    max_steps = max(max_steps, steps[p]);
                }
                for (uint32_t s = 0; s < max_steps; s++) {
#pragma unroll
                    for (int h = 0; h < kPairs / kBatch; h++) {
                        half4_t fa[kBatch], fb[kBatch];
#pragma unroll
                        for (int q = 0; q < kBatch; q++) {
                            const int p = kBatch * h + q;
                            if (s < steps[p]) {
                                const uint32_t bq = max(which ? sb[2 * p + 1] : sb[2 * p], lo), eq = min(which ? sb[2 * p + 2] : sb[2 * p + 1], hi);
                                const uint32_t t = bq + 2 * s + slot;
                                const bool live = t < eq;
                                const uint32_t i = live ? t - lo : 0u;
                                fa[q] = load_nibble_wide(M.abmp[i], ra, M.aoff[i], nl, s_sel, live);
                                fb[q] = load_nibble_wide(M.bbmp[i], rb, M.boff[i], nl, s_sel, live);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < kBatch; q++) {
                            const int p = kBatch * h + q;
                            if (s < steps[p]) acc[p] = __builtin_amdgcn_mfma_f32_16x16x16f16(fa[q], fb[q], acc[p], 0, 0, 0);
                        }
                    }
                }
            }
            // result: lane holds D[4*(lane>>4)+r][lane&15]; block 0 of a pair lives in rows/cols 0-7, block 1 in 8-15.  The
            // tiles go through LDS into position-per-lane form: lane = tile position, so the rank of a position inside the
            // C bitmap is one mbcnt against the bit-reversed (wave-uniform) bitmap and each C block is one contiguous store.
            const int col = lane & 15, rq = lane >> 4;
            const int blk = col >> 3;
            if ((rq >> 1) == blk) {
#pragma unroll
                for (int p = 0; p < kGroupC / 2; p++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) s_out[w][2 * p + blk][((rq & 1) * 4 + r) * 8 + (col & 7)] = acc[p][r];
                }
            }
            const uint64_t my_cbmp = M.cbmp[lane & (kGroupC - 1)], my_coff = M.coff[lane & (kGroupC - 1)];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < kGroupC; j++) {
                if (c0 + j < c_size) {
                    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_cbmp, j);
                    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_cbmp >> 32), j);
                    const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_coff >> 32), j) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_coff, j);
                    // position `lane` is bit 63-lane of the bitmap = bit `lane` of its reversal
                    const uint32_t rlo = __builtin_bitreverse32(bhi), rhi = __builtin_bitreverse32(blo);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(rhi, __builtin_amdgcn_mbcnt_lo(rlo, 0u));
                    const uint32_t word = lane < 32 ? rlo : rhi;
                    if ((word >> (lane & 31)) & 1u) (c_vals + off)[rank] = s_out[w][j][lane];
                }
            }
        }
        // hand the stage-2 words to the LDS slot stage 3 will read next iteration (the other slot was read above)
        if (have2) {
            MacMeta &N = s_meta[w][(k - 2) & 1];
            N.abmp[lane] = m_abmp; N.aoff[lane] = m_aoff;
            N.bbmp[lane] = m_bbmp; N.boff[lane] = m_boff;
            if (lane < kGroupC) { N.cbmp[lane] = m_cbmp; N.coff[lane] = m_coff; }
        }
        __builtin_amdgcn_wave_barrier();
        tbv3 = tbv2; tbv2 = tbv1; tbv1 = tbv0;
        tk2 = tk1;
    }
}

// Group size 4 (two MFMA pairs per step, 68 VGPRs, 7 waves/SIMD) measured best of {16, 8, 4, 2} x occupancy caps on the three
// generator cases (block-MAC log): larger groups fetch more per round trip but halve the resident waves.
constexpr int kGroupC = 4;
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) void block_mac_mfma_f16_group_kernel(BMSP_MAC_ARGS)
{
    block_mac_group_body<kGroupC>(BMSP_MAC_PASS);
}

// ---- T_7, vector-ALU numerics on the group schedule ---------------------------------------------------------------
// Same software pipeline as the MFMA group kernel (task_begin / tasks / bitmaps+offsets of the next groups are in flight
// while the current group computes), V15 numerics of block_mac_valu_kernel: lane l owns C(l/8, l%8) and adds the eight
// products of a task in k order, tasks in list order.  Per task the two tiles are expanded position-per-lane -- the
// task's bitmap is wave-uniform, so a lane's value index is one mbcnt against the reversed bitmap -- and staged through
// LDS; value loads of kValuBatch consecutive tasks (across C-block boundaries) are issued before the first is consumed.
constexpr int kValuGroupC = 16;  // (16, 4) measured best of G in {4,8,16,32} x U in {2,4,8} on the generator cases
constexpr int kValuBatch = 4;

template <typename T>
struct ValuLoad;
template <>
struct ValuLoad<float> {
    static __device__ __forceinline__ float ld(rsrc_t r, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
};
template <>
struct ValuLoad<_Float16> {
    static __device__ __forceinline__ _Float16 ld(rsrc_t r, uint32_t off) { return ld_half(r, off); }
};
template <>
struct ValuLoad<double> {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ double ld(rsrc_t r, uint32_t off) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0)); }
};

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}
// position `lane` of a tile is bit 63-lane of its bitmap = bit `lane` of the reversed bitmap; rank = set positions before it
__device__ __forceinline__ bool tile_lane(uint64_t bmp_uniform, int lane, uint32_t &rank)
{
    const uint32_t rlo = __builtin_bitreverse32((uint32_t)(bmp_uniform >> 32)), rhi = __builtin_bitreverse32((uint32_t)bmp_uniform);
    rank = __builtin_amdgcn_mbcnt_hi(rhi, __builtin_amdgcn_mbcnt_lo(rlo, 0u));
    return ((lane < 32 ? rlo : rhi) >> (lane & 31)) & 1u;
}

// DENSE (fp16 operands that carry the dense copy of their tiles, the one the MFMA kernels use): a tile is 128 contiguous bytes in position
// order, so the element gathers, their rank arithmetic and the block records go away -- one 16-byte load per lane brings the A and B tiles
// of FOUR tasks (lane = task u = l >> 4, operand (l >> 3) & 1, line l & 7) and one ds_write_b128 parks them; the k loop is unchanged.
template <typename T, int G, int U, bool DENSE = false>
__global__ __launch_bounds__(kThreads) void block_mac_valu_group_kernel(const uint64_t *__restrict__ tasks, const uint32_t *__restrict__ task_begin,
                                                                        const uint32_t *__restrict__ a_meta, uint32_t a_meta_bytes,
                                                                        const T *__restrict__ a_vals, const uint32_t *__restrict__ b_meta,
                                                                        uint32_t b_meta_bytes, const T *__restrict__ b_vals,
                                                                        const uint64_t *__restrict__ c_bmps, const uint64_t *__restrict__ c_offs,
                                                                        typename MacOps<T>::Out *__restrict__ c_vals, uint32_t c_size, uint32_t a_bytes,
                                                                        uint32_t b_bytes, const u32x4_t *__restrict__ a_dense = nullptr,
                                                                        const u32x4_t *__restrict__ b_dense = nullptr)
{
    using O = typename MacOps<T>::Out;
    static_assert(!DENSE || (U == 4 && (sizeof(T) == 2 || sizeof(T) == 4)), "the dense staging maps 64 lanes onto (tasks) x 2 operands x (16-byte lines of a tile)");
    __shared__ __attribute__((aligned(16))) T tile_a[4][U][64];
    __shared__ __attribute__((aligned(16))) T tile_b[4][U][64];
    const int w = wave_id(), lane = lane_id();
    const int i = lane >> 3, j = lane & 7;
    const rsrc_t ra = make_rsrc(a_vals, a_bytes), rb = make_rsrc(b_vals, b_bytes);
    const rsrc_t rma = make_rsrc(a_meta, a_meta_bytes), rmb = make_rsrc(b_meta, b_meta_bytes);
    const uint32_t groups = (c_size + G - 1) / G;
    const uint32_t g0 = blockIdx.x * 4 + w, stride = gridDim.x * 4;
    const uint32_t mine = g0 < groups ? (groups - g0 + stride - 1) / stride : 0u;
    uint32_t tbv1 = 0, tbv2 = 0, tbv3 = 0;  // lane l <= G holds task_begin[G g + l] of the group in stage 1 / 2 / 3
    uint64_t tk2 = 0;
    // stage-3 inputs, fetched one iteration earlier: lane l holds the words of task tb + l, lane l < G those of C block l
    uint64_t w_abmp = 0, w_bbmp = 0, w_cbmp = 0, w_coff = 0;
    uint32_t w_aoff = 0, w_boff = 0;
    for (uint32_t k = 0; k < mine + 3; k++) {
        uint32_t tbv0 = 0;
        if (k < mine) tbv0 = task_begin[min((g0 + k * stride) * G + (uint32_t)min(lane, G), c_size)];
        uint64_t tk1 = 0;
        {
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)tbv1, 0), te = (uint32_t)__builtin_amdgcn_readlane((int)tbv1, G);
            if (tb + lane < min(te, tb + 64u)) tk1 = tasks[tb + lane];
        }
        uint64_t m_abmp = 0, m_bbmp = 0, m_cbmp = 0, m_coff = 0;
        uint32_t m_aoff = 0, m_boff = 0;
        {
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)tbv2, 0), te = (uint32_t)__builtin_amdgcn_readlane((int)tbv2, G);
            if (DENSE) {
                m_abmp = tk2;  // the task word itself: tile indices are all the dense staging needs
            } else if (tb + lane < min(te, tb + 64u)) {
                load_block_meta(rma, (uint32_t)(tk2 >> 32), m_abmp, m_aoff);
                load_block_meta(rmb, (uint32_t)tk2, m_bbmp, m_boff);
                m_aoff *= (uint32_t)sizeof(T); m_boff *= (uint32_t)sizeof(T);
            }
            if (k >= 2 && k - 2 < mine && lane < G) {
                const uint32_t cc = min((g0 + (k - 2) * stride) * G + (uint32_t)lane, c_size - 1);
                m_cbmp = c_bmps[cc]; m_coff = c_offs[cc];
            }
        }
        if (k >= 3) {
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)tbv3, 0), te = (uint32_t)__builtin_amdgcn_readlane((int)tbv3, G);
            uint32_t c_cur = 0, cur_end = (uint32_t)__builtin_amdgcn_readlane((int)tbv3, 1), win_lo = tb;
            O acc = 0;
            for (uint32_t t = tb; t < te; t += U) {
                if (t >= win_lo + 64u) {  // hub group: next 64-task window, fetched in place
                    win_lo = t;
                    w_abmp = 0; w_bbmp = 0; w_aoff = 0; w_boff = 0;
                    if (t + lane < te) {
                        const uint64_t tk = tasks[t + lane];
                        if (DENSE) {
                            w_abmp = tk;
                        } else {
                            load_block_meta(rma, (uint32_t)(tk >> 32), w_abmp, w_aoff);
                            load_block_meta(rmb, (uint32_t)tk, w_bbmp, w_boff);
                            w_aoff *= (uint32_t)sizeof(T); w_boff *= (uint32_t)sizeof(T);
                        }
                    }
                }
                if constexpr (DENSE) {
                    // a tile is 64 / EPL lines of 16 bytes: 8 (fp16) or 16 (fp32); one load instruction brings 64 / (2 LPT) tasks
                    constexpr int EPL = 16 / (int)sizeof(T), LPT = 64 / EPL, TPLD = 64 / (2 * LPT), ROUNDS = U / TPLD;
                    const int op = (lane / LPT) & 1, line = lane % LPT;
                    u32x4_t v[ROUNDS];
                    bool on[ROUNDS];
#pragma unroll
                    for (int q = 0; q < ROUNDS; q++) {
                        const uint32_t du = (uint32_t)(q * TPLD + lane / (2 * LPT));
                        const uint32_t src = min(t + du - win_lo, 63u);
                        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)w_abmp, (int)src, kWave), hi = (uint32_t)__shfl((int)(uint32_t)(w_abmp >> 32), (int)src, kWave);
                        on[q] = t + du < te;
                        v[q] = u32x4_t{0u, 0u, 0u, 0u};
                        if (on[q]) v[q] = (op ? b_dense + (size_t)lo * LPT : a_dense + (size_t)hi * LPT)[line];
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < ROUNDS; q++) {
                        const int du = q * TPLD + lane / (2 * LPT);
                        if (on[q]) *(u32x4_t *)((op ? tile_b[w][du] : tile_a[w][du]) + line * EPL) = v[q];  // A(i,k) at i*8+k, B(k,j) at j*8+k: the copies' own order
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                T av[U], bv[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t l = min(t + u - win_lo, 63u);
                    const bool on = t + u < te;
                    uint32_t rka, rkb;
                    const bool ha = tile_lane(readlane_u64(w_abmp, l), lane, rka) && on;
                    const bool hb = tile_lane(readlane_u64(w_bbmp, l), lane, rkb) && on;
                    const uint32_t oa = (uint32_t)__builtin_amdgcn_readlane((int)w_aoff, (int)l), ob = (uint32_t)__builtin_amdgcn_readlane((int)w_boff, (int)l);
                    av[u] = ValuLoad<T>::ld(ra, ha ? oa + rka * (uint32_t)sizeof(T) : kOob);
                    bv[u] = ValuLoad<T>::ld(rb, hb ? ob + rkb * (uint32_t)sizeof(T) : kOob);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < U; u++) {
                    tile_a[w][u][lane] = av[u];  // A(i,k) at i*8+k
                    tile_b[w][u][lane] = bv[u];  // B column-major in the tile: B(k,j) at j*8+k
                }
                __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (t + u < te) {
                        if (t + u >= cur_end) {  // the list of C block c_cur is exhausted: compacting store, next block
                            uint32_t rk;
                            if (tile_lane(readlane_u64(w_cbmp, c_cur), lane, rk)) (c_vals + readlane_u64(w_coff, c_cur))[rk] = acc;
                            acc = 0;
                            c_cur++;
                            cur_end = (uint32_t)__builtin_amdgcn_readlane((int)tbv3, (int)(c_cur + 1));
                        }
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) acc = MacOps<T>::step(tile_a[w][u][i * 8 + kk], tile_b[w][u][j * 8 + kk], acc);
                    }
                }
            }
            if (te > tb) {
                uint32_t rk;
                if (tile_lane(readlane_u64(w_cbmp, c_cur), lane, rk)) (c_vals + readlane_u64(w_coff, c_cur))[rk] = acc;
            }
        }
        w_abmp = m_abmp; w_bbmp = m_bbmp; w_cbmp = m_cbmp; w_coff = m_coff;
        w_aoff = m_aoff; w_boff = m_boff;
        tbv3 = tbv2; tbv2 = tbv1; tbv1 = tbv0;
        tk2 = tk1;
    }
}

template <typename T>
void launch_mac_valu(const uint64_t *tasks, const uint32_t *task_begin, bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C,
                     hipStream_t st)
{
    uint32_t cs = (uint32_t)C->block_num;
    if (!cs) return;
    const uint64_t a_bytes = (uint64_t)A->values_extent() * sizeof(T), b_bytes = (uint64_t)B->values_extent() * sizeof(T);
    if (a_bytes < (1ull << 32) && b_bytes < (1ull << 32) && A->block_num < (1ll << 28) && B->block_num < (1ll << 28)) {
        ensure_block_meta(A, st);
        ensure_block_meta(B, st);
        const uint32_t groups = (cs + kValuGroupC - 1) / kValuGroupC;
        uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)groups + 3) / 4, 256ull * 64);
        if constexpr (sizeof(T) == 2 || sizeof(T) == 4) {
            // fp16 / fp32: stage the tiles from their dense copies (64 elements per block: 128 / 256 B; BMSP_MAC_VALU_DENSE=0 keeps the element
            // gathers, and so does a copy above 4 GiB)
            const char *de = getenv("BMSP_MAC_VALU_DENSE");
            const uint64_t per_block = 64 * sizeof(T);
            // fp16 copies exist anyway (the MFMA kernels' operands).  An fp32 copy is 256 B per block: taken when the tiles are at least a
            // quarter full (the copy is then <= 4x the values; FEM-like 3.7 values per tile: 17x the memory for -10 % of T_7, not taken)
            const bool worth = sizeof(T) == 2 || (A->nnz >= 16 * A->block_num && B->nnz >= 16 * B->block_num);
            const bool dense = de ? de[0] == '1'
                                  : (worth && (uint64_t)A->block_num * per_block <= (4ull << 30) && (uint64_t)B->block_num * per_block <= (4ull << 30));
            if (dense) {
                ensure_dense_tiles(A, st);
                ensure_dense_tiles(B, st);
                hipLaunchKernelGGL((block_mac_valu_group_kernel<T, kValuGroupC, kValuBatch, true>), dim3(grid), dim3(kThreads), 0, st, tasks, task_begin,
                                   A->block_meta, (uint32_t)(A->block_num * 16), (const T *)A->values, B->block_meta, (uint32_t)(B->block_num * 16),
                                   (const T *)B->values, C->bmps, C->offsets, (typename MacOps<T>::Out *)C->values, cs, (uint32_t)a_bytes,
                                   (uint32_t)b_bytes, (const u32x4_t *)A->dense_tiles, (const u32x4_t *)B->dense_tiles);
                BMSP_CHECK_LAUNCH();
                return;
            }
        }
        hipLaunchKernelGGL((block_mac_valu_group_kernel<T, kValuGroupC, kValuBatch>), dim3(grid), dim3(kThreads), 0, st, tasks, task_begin, A->block_meta,
                           (uint32_t)(A->block_num * 16), (const T *)A->values, B->block_meta, (uint32_t)(B->block_num * 16), (const T *)B->values,
                           C->bmps, C->offsets, (typename MacOps<T>::Out *)C->values, cs, (uint32_t)a_bytes, (uint32_t)b_bytes);
    } else {  // value arrays beyond the 4 GiB a buffer descriptor addresses: pointer-based kernel
        uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)cs + 3) / 4, 256ull * 64);
        hipLaunchKernelGGL((block_mac_valu_kernel<T>), dim3(grid), dim3(kThreads), 0, st, tasks, task_begin, A->bmps, A->offsets,
                           (const T *)A->values, B->bmps, B->offsets, (const T *)B->values, C->bmps, C->offsets,
                           (typename MacOps<T>::Out *)C->values, cs);
    }
    BMSP_CHECK_LAUNCH();
}

void print_stage(bool verbose, const char *name, double us)
{
    // the reference's VERBOSE lines (src/bmSparse_SPGEMM.cu:852 ..): "T_1: <n> μs "
    if (verbose) printf("%s: %lld \xce\xbcs \n", name, (long long)(us + 0.5));
}

}  // namespace

// T_7 from a sorted task list: the block multiply-accumulate kernel of the tc_version and value type (shared by the whole product and by
// bmsp_spgemm_numeric on a C that kept its list)
static void launch_block_mac(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, const uint64_t *tasks_sorted, uint64_t n_tasks, const uint32_t *task_begin,
                             const uint32_t *c_of_wave, uint64_t total, int tc_version, bool mfma, bmsp_spgemm_stats *S, hipStream_t st)
{
    const uint32_t c_size = (uint32_t)C->block_num;
    if (mfma) {
        if ((uint64_t)A->values_extent() * 2 + 16 >= (1ull << 32) || (uint64_t)B->values_extent() * 2 + 16 >= (1ull << 32))
            fail(BMSP_ERR_LIMIT, "MFMA block-MAC addresses operand values through 4 GiB buffer descriptors; use tc_version 5");
        // the group kernel's 12-byte value loads may run past the last stored value: arrays from this library's
        // allocator carry that slack (runtime.h), borrowed arrays (bmsp_matrix_from_arrays, ownership 2) may not
        const uint32_t a_bytes = (uint32_t)(A->values_extent() * 2), b_bytes = (uint32_t)(B->values_extent() * 2);
        const bool old_mac = getenv("BMSP_MAC_OLD") != nullptr;  // experiment switch: the r1 16x16x16 group kernel
        if (tc_version == 4 && !old_mac && mac_mfma32_supported(A, B)) {
            // many tasks per C tile and most candidate pairs alive: the strip kernel (operand reuse; reads A, B, C, not the task list)
            if (mac_strip_eligible(A, B, C, total, n_tasks, st)) {
                S->mac_variant = launch_mac_strip(A, B, C, tc_version, st);
            } else {
                S->mac_variant = launch_mac_mfma32(tasks_sorted, n_tasks, task_begin, c_of_wave, A, B, C, st);
            }
        } else if (tc_version == 4 && pool_owns(A->values) && pool_owns(B->values) && A->block_num < (1ll << 28) && B->block_num < (1ll << 28)) {
            ensure_block_meta(A, st);
            ensure_block_meta(B, st);
            uint32_t groups = (c_size + kGroupC - 1) / kGroupC;
            uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)groups + 3) / 4, 256ull * 64);
            hipLaunchKernelGGL(block_mac_mfma_f16_group_kernel, dim3(grid), dim3(kThreads), 0, st, tasks_sorted, task_begin, A->block_meta,
                               (uint32_t)(A->block_num * 16), (const _Float16 *)A->values, B->block_meta, (uint32_t)(B->block_num * 16),
                               (const _Float16 *)B->values, C->bmps, C->offsets, (float *)C->values, c_size, a_bytes + 16u, b_bytes + 16u);
        } else {
            uint32_t groups = ((c_size + 1) / 2 + kPairsPerWave - 1) / kPairsPerWave;
            uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)groups + 3) / 4, 256ull * 64);
            hipLaunchKernelGGL(block_mac_mfma_f16_kernel, dim3(grid), dim3(kThreads), 0, st, tasks_sorted, task_begin, A->bmps, A->offsets,
                               (const _Float16 *)A->values, B->bmps, B->offsets, (const _Float16 *)B->values, C->bmps, C->offsets,
                               (float *)C->values, c_size, a_bytes, b_bytes);
        }
        BMSP_CHECK_LAUNCH();
        S->mac_kernel = tc_version;
    } else if (A->dtype != BMSP_F64 && !getenv("BMSP_MAC_VALU_DENSE") && !getenv("BMSP_MAC_F32MFMA") && mac_rowsparse_applies(A, B, tc_version, st) &&
               mac_rowsparse_fits_c(C, st)) {
        // V15 numerics on operands of nearly empty tiles: the products of stored values only (blockmac_rowsparse.hip); the task list is not read
        launch_mac_rowsparse(A, B, C, st);
        S->mac_variant = BMSP_MAC_ROWSPARSE;
        S->mac_kernel = 5;
    } else {
        if (A->dtype == BMSP_F32 && launch_mac_f32_mfma(tasks_sorted, n_tasks, task_begin, c_of_wave, A, B, C, st)) S->mac_variant = BMSP_MAC_F32MFMA;
        else if (A->dtype == BMSP_F32) launch_mac_valu<float>(tasks_sorted, task_begin, A, B, C, st);
        else if (A->dtype == BMSP_F16) launch_mac_valu<_Float16>(tasks_sorted, task_begin, A, B, C, st);
        else launch_mac_valu<double>(tasks_sorted, task_begin, A, B, C, st);
        S->mac_kernel = 5;
    }
}

void spgemm(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **Cout, int mode, int tc_version, int verbose, hipStream_t st,
            bmsp_spgemm_stats *stats, bool structure_only)
{
    if (!A || !B || !Cout) fail(BMSP_ERR_INVALID, "null argument");
    if (A->transposed) fail(BMSP_ERR_INVALID, "A must be built with transposed=0");
    if (!B->transposed) fail(BMSP_ERR_INVALID, "B must be built with transposed=1 (column-major tiles)");
    if (A->num_cols != B->num_rows) fail(BMSP_ERR_INVALID, "shape mismatch: A is %dx%d, B is %dx%d", A->num_rows, A->num_cols, B->num_rows, B->num_cols);
    if (A->dtype != B->dtype) fail(BMSP_ERR_INVALID, "A and B must have the same value type");
    if (mode < 0 || mode > 2) fail(BMSP_ERR_INVALID, "sort mode must be 0, 1 or 2");
    if (tc_version < 1 || tc_version > 5) fail(BMSP_ERR_INVALID, "tc_version must be 1..5");
    const bool mfma = tc_version != 5 && A->dtype == BMSP_F16;

    bmsp_spgemm_stats local{};
    bmsp_spgemm_stats *S = stats ? stats : &local;
    *S = bmsp_spgemm_stats{};
    const bool timing = verbose || stats;
    StageTimer tm(st, timing);
    tm.mark(-1);

    // T_1: blocks per block-row of B (cached dense pointer; the per-matrix row maxima are cached with it)
    ensure_rowptr(B, st);
    ensure_row_stats(A, st);
    ensure_row_stats(B, st);
    tm.mark(1);

    // T_2 + T_3 (first half): fan-out per A block and its exclusive scan
    const uint64_t n_a = (uint64_t)A->block_num;
    DevBuf<uint64_t> first_pos;
    uint64_t total = 0;
    auto run_t2 = [&]() {
        first_pos.alloc(n_a + 1);
        HostScalar<uint64_t> total_h;
        device_exclusive_scan<uint64_t>(FanOut{A->keys, B->rowptr, n_a, (uint32_t)B->num_block_rows()}, PtrOutTotal<uint64_t>{first_pos.p, n_a, total_h.dev()},
                                        n_a + 1, st);
        total = total_h.wait(st);
        tm.mark(2);
        S->task_list_size = (int64_t)total;
        if (total >= (1ull << 32))  // bmsp_spgemm retries in block-row panels (shard.hip: spgemm_paneled)
            throw TaskRangeExceeded(BMSP_ERR_LIMIT, std::to_string(total) + " candidate block pairs exceed the 32-bit range of one task list");
    };

    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> C(new bmsp_matrix_s(), free_matrix);
    C->num_rows = A->num_rows; C->num_cols = B->num_cols;  // :1171-1172
    C->dtype = A->dtype == BMSP_F64 ? BMSP_F64 : BMSP_F32;  // OUTPUT_TYPE float (:51)
    C->transposed = 0;
    // value offsets and nnz from C's bitmaps, C's value array (the tail of T_9)
    auto finish_structure = [&]() -> uint64_t {
        const uint32_t c_size = (uint32_t)C->block_num;
        uint64_t c_nnz = 0;
        if (C->offsets) {  // the row-merge passes write the offsets themselves (scan over block-rows + scan inside the block-row)
            C->values = pool_alloc(dtype_size(C->dtype) * (size_t)(C->nnz ? C->nnz : 1));
            return (uint64_t)C->nnz;
        }
        C->offsets = (uint64_t *)pool_alloc(8 * ((size_t)c_size + 1));
        if (c_size) {
            HostScalar<uint64_t> c_nnz_h;
            device_exclusive_scan<uint64_t>(PopcIn{C->bmps, c_size}, PtrOutTotal<uint64_t>{C->offsets, c_size, c_nnz_h.dev()}, (uint64_t)c_size + 1, st);
            c_nnz = c_nnz_h.wait(st);
        } else {
            BMSP_HIP(hipMemsetAsync(C->offsets, 0, 8, st));
        }
        C->nnz = (int64_t)c_nnz;
        C->values = pool_alloc(dtype_size(C->dtype) * (size_t)(c_nnz ? c_nnz : 1));
        return c_nnz;
    };
    const double host_t0 = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3;
    auto finish = [&]() {
        BMSP_HIP(hipStreamSynchronize(st));  // reference: cudaDeviceSynchronize (:1158)
        segsort_check_violation();
        if (getenv("BMSP_HOST_TIMES")) {
            fprintf(stderr, "[host times] spgemm entered %.0f us before the first mark\n", tm.n ? tm.host_us[0] - host_t0 : 0.0);
            tm.print_host_times("spgemm marks (stage: host us since the previous mark)");
        }
        S->t_us[0] = tm.collect(S->t_us);
        S->t_us[5] += S->t_us[8];  // the reference's T_5 includes its "Segmented sort" sub-timer (:1009-1024)
        if (verbose) {
            // the reference's VERBOSE lines, in its order (src/bmSparse_SPGEMM.cu:849-1164)
            print_stage(true, "T_1", S->t_us[1]);
            print_stage(true, "T_2", S->t_us[2]);
            printf("Task list size: %llu\n", (unsigned long long)total);
            print_stage(true, "T_3", S->t_us[3]);
            printf("Bmp reduction: %lld\n", (long long)S->bmp_reduction);
            print_stage(true, "T_4", S->t_us[4]);
            if (S->sort_path == 1) print_stage(true, "Segmented sort", S->t_us[8]);
            print_stage(true, "T_5", S->t_us[5]);
            print_stage(true, "T_6", S->t_us[6]);
            print_stage(true, "T_9", S->t_us[9]);
            print_stage(true, "T_7", S->t_us[7]);
        }
        S->c_blocks = C->block_num;
        S->c_nnz = C->nnz;
        // which structures this C is the product of (row-panel views are not stamped: bmsp_spgemm_numeric then checks the structure itself)
        const bool a_view = A->ownership == 2 && (A->view_block_begin || A->view_values_end);
        if (!a_view) { C->sp_a_hash = ensure_struct_hash(A, st); C->sp_b_hash = ensure_struct_hash(B, st); }
        *Cout = C.release();
    };

    // Row-merge paths (rowmerge.hip): C's structure formed block-row by block-row in LDS instead of expand - sort - compress.  Taken when
    // the sort mode is left to the library.  (1) Strip mode -- one pass, no task list, numeric stage by the strip kernel -- when the
    // operands fit that kernel and every block-row of C holds at most mac_strip_row_cap() tiles; (2) task-list mode -- a count pass and a
    // fill pass that also writes the sorted task list, numeric stage by the task-list kernels of the tc_version -- for block-rows of C
    // of up to ~900 tiles; (3) the pipeline below otherwise (hub rows).  What a pair of operands turned out to need is remembered on A's
    // handle, so that only the first product of a pair pays for a pass that did not fit (cage-like: 280 us).  BMSP_SPGEMM_ROWMERGE=0
    // switches both off, =1 takes them under an explicit sort mode too, =2 skips strip mode.
    uint64_t n_tasks = 0;
    uint32_t c_size = 0;
    DevBuf<uint64_t> k0, k1, v0, v1, rm_tasks;
    DevBuf<uint32_t> task_begin, c_of_wave;
    const uint64_t *tasks_sorted = nullptr;
    bool have_tasks = false;
    {
        const char *rme = getenv("BMSP_SPGEMM_ROWMERGE");
        const bool rm_force = rme && rme[0] != '0', rm_off = rme && rme[0] == '0', rm_no_strip = rme && rme[0] == '2';
        const bool known = A->rm_partner_uid == B->uid;
        const int hint = known ? A->rm_partner_mode : 0;  // 0 unknown, 1 strip mode fits, 2 task-list mode, 3 pipeline
        auto remember = [&](int m) {
            if (A->rm_partner_uid != B->uid) A->rm_partner_cw_hash = 0;
            A->rm_partner_uid = B->uid; A->rm_partner_blocks = B->block_num; A->rm_partner_mode = m;
        };
        // numeric stages that work from C's structure alone: the K = 32 MFMA strip kernel (tc_version 4, fp16) and its fp32 form (V15's
        // summation order on v_mfma_f32_16x16x4_f32: any tc_version, as fp32 operands always take V15's numerics)
        // ... and the row-sparse kernel: V15 numerics (fp32 operands, or fp16 under tc_version 5 -- the reference's default configuration) on
        // operands of nearly empty tiles
        const bool rm_numeric = (tc_version == 4 && mfma && !getenv("BMSP_MAC_OLD")) || (A->dtype == BMSP_F32 && !getenv("BMSP_MAC_VALU_DENSE") && !getenv("BMSP_MAC_F32MFMA")) ||
                                (A->dtype == BMSP_F16 && tc_version == 5 && !getenv("BMSP_MAC_VALU_DENSE") && mac_rowsparse_applies(A, B, tc_version, st));
        const char *sf = getenv("BMSP_MAC_STRIP");
        // BMSP_SPGEMM_ROWWINDOW=0: never the column-window passes (rowwindow.hip); =1: always, whatever the operands look like (tests)
        const char *we = getenv("BMSP_SPGEMM_ROWWINDOW");
        const bool win_off = we && we[0] == '0', win_force = we && we[0] == '1';
        const bool strip_allowed = rm_numeric && !rm_no_strip && !(sf && sf[0] == '0') && !win_force;
        // strip mode done: value array, numeric stage (or zero values for bmsp_spgemm_symbolic), statistics
        auto finish_strip = [&](uint64_t surv) {
            tm.mark(3);
            remember(1);
            S->task_list_size = (int64_t)total;
            S->surviving_tasks = (int64_t)surv;
            S->bmp_reduction = (int64_t)(total - surv);
            S->sort_path = BMSP_SORT_PATH_ROWMERGE;
            S->sort_long = BMSP_ROWMERGE_STRIP;
            finish_structure();
            tm.mark(9);
            if (C->block_num && !structure_only) {
                S->mac_variant = launch_mac_strip(A, B, C.get(), tc_version, st);
                S->mac_kernel = mfma ? tc_version : 5;
            } else if (C->nnz) {
                BMSP_HIP(hipMemsetAsync(C->values, 0, dtype_size(C->dtype) * (size_t)C->nnz, st));  // bmsp_spgemm_symbolic: structure only
            }
            tm.mark(7);
            finish();
        };
        auto drop_structure = [&]() {
            pool_free(C->keys); pool_free(C->bmps); pool_free(C->rowptr); pool_free(C->offsets);
            C->keys = nullptr; C->bmps = nullptr; C->rowptr = nullptr; C->offsets = nullptr; C->nnz = 0;
            C->rowptr_rows = 0; C->max_row_blocks = -1; C->block_num = 0;
        };
        bool strip_tried = false;
        // A pair of operands known to fit strip mode: the pass runs WITHOUT T_2 (its scan, and the wait for its total, only serve the task
        // list); it counts the candidate pairs itself.
        // (also for a pair never seen whose row maxima bound every block-row of C by the strip kernel's capacity)
        const bool surely_fits = A->max_row_blocks >= 0 && B->max_row_blocks >= 0 &&
                                 (uint64_t)A->max_row_blocks * (uint64_t)B->max_row_blocks <= (uint64_t)mac_strip_row_cap();
        if (((known && hint == 1) || (hint == 0 && surely_fits)) && n_a && strip_allowed && (mode == BMSP_SORT_AUTO || rm_force) && !rm_off &&
            (uint64_t)A->num_block_rows() * (uint64_t)mac_strip_row_cap() * 12 <= (4ull << 30) &&  // (beyond: T_2 first, scratch sized by the candidate pairs)
            mac_structure_numeric_ok(A, B, tc_version, st)) {
            uint64_t surv = 0;
            strip_tried = true;
            if (rowmerge_symbolic(A, B, C.get(), nullptr, 0, mac_strip_row_cap(), &surv, &total, st)) {
                if (C->block_num == 0 || mac_strip_fits_c(C.get(), st)) {
                    finish_strip(surv);
                    return;
                }
                drop_structure();
            }
            tm.mark(-1);
        }
        run_t2();
        const bool rm_on = !rm_off && total && (mode == BMSP_SORT_AUTO || rm_force);
        bool try_strip = rm_on && strip_allowed && !strip_tried && hint != 2 && hint != 3 && hint != 4;
        if (try_strip) try_strip = mac_structure_numeric_ok(A, B, tc_version, st);
        if (try_strip) {
            uint64_t surv = 0, cand = 0;
            if (rowmerge_symbolic(A, B, C.get(), first_pos.p, total, mac_strip_row_cap(), &surv, &cand, st)) {
                if (C->block_num == 0 || mac_strip_fits_c(C.get(), st)) {
                    finish_strip(surv);
                    return;
                }
                // C exists but a strip of it exceeds the kernel's column list: drop it
                drop_structure();
            }
            tm.mark(-1);
        }
        if (rm_on && hint != 3) {
            // task-list mode (a wave per block-row, hash table of ~900 C tiles); operands with hub block-rows: column windows (a workgroup
            // per block-row and window of block columns, dense tables: rowwindow.hip); neither applies: the pipeline
            int got = 0;
            if (hint != 4 && !win_force && rowmerge_tasklist(A, B, C.get(), first_pos.p, total, rm_tasks, task_begin, c_of_wave, &n_tasks, st)) got = 2;
            else if (!win_off && rowmerge_windowed(A, B, C.get(), first_pos.p, total, rm_tasks, task_begin, c_of_wave, &n_tasks, st)) got = 4;
            if (got) {
                remember(got);
                have_tasks = true;
                tasks_sorted = rm_tasks.p;
                c_size = (uint32_t)C->block_num;
                tm.mark(3);
                S->surviving_tasks = (int64_t)n_tasks;
                S->bmp_reduction = (int64_t)(total - n_tasks);
                S->sort_path = got == 2 ? BMSP_SORT_PATH_ROWMERGE : BMSP_SORT_PATH_ROWWINDOW;
                if (got == 2) S->sort_long = BMSP_ROWMERGE_TASKLIST;
                finish_structure();
                tm.mark(9);
            } else {
                remember(3);
                tm.mark(-1);
            }
        }
    }

    if (!have_tasks) {
    // T_3 + T_4: expansion fused with the bitmap filter
    const int jbits = std::max(1, ceil_log2_u64((uint64_t)B->num_block_cols()));
    const int ibits = std::max(1, ceil_log2_u64((uint64_t)A->num_block_rows()));
    const bool small_product = total <= (2u << 20);
    const uint32_t tile_e = small_product ? 2u * kThreads : (uint32_t)kTile;
    const uint32_t tiles = (uint32_t)((total + tile_e - 1) / tile_e);
    ExpandArgs ea{first_pos.p, A->keys, A->bmps, B->keys, B->bmps, B->rowptr, n_a, total, jbits};
    DevBuf<uint32_t> tile_counts((size_t)tiles + 1);
    // The single-pass form (decoupled look-back, candidate-sized output) is kept as an option: measured on MI355X it costs what the
    // count + write pair costs (cage-like 224 vs 106 + 128 us, FEM-like 358 vs 147 + 210 us) and loses on small products (full-tile
    // banded 69 vs 49 us): a workgroup holds its slot through the look-back and the sixteen ordered block scans of the write.
    const bool lookback = total && total <= (512ull << 20) && getenv("BMSP_EXPAND_LOOKBACK") != nullptr;
    if (lookback) {
        tm.mark(3);
        k0.alloc(total); v0.alloc(total);
        DevBuf<unsigned long long> state((size_t)tiles + 1);
        BMSP_HIP(hipMemsetAsync(state.p, 0, 8 * ((size_t)tiles + 1), st));
        uint32_t *counter = (uint32_t *)(state.p + tiles);
        HostScalar<uint32_t> n_tasks_h;
        if (small_product)
            hipLaunchKernelGGL((expand_filter_lookback_kernel<2>), dim3(tiles), dim3(kThreads), 0, st, ea, state.p, counter, tiles, k0.p, v0.p, n_tasks_h.dev());
        else
            hipLaunchKernelGGL((expand_filter_lookback_kernel<kItems>), dim3(tiles), dim3(kThreads), 0, st, ea, state.p, counter, tiles, k0.p, v0.p, n_tasks_h.dev());
        BMSP_CHECK_LAUNCH();
        n_tasks = n_tasks_h.wait(st);
        k1.alloc(n_tasks); v1.alloc(n_tasks);
        tm.mark(4);
    } else if (total) {
        if (small_product)
            hipLaunchKernelGGL((expand_filter_kernel<false, 2>), dim3(tiles), dim3(kThreads), 0, st, ea, tile_counts.p, (const uint32_t *)nullptr,
                               (uint64_t *)nullptr, (uint64_t *)nullptr);
        else
            hipLaunchKernelGGL((expand_filter_kernel<false, kItems>), dim3(tiles), dim3(kThreads), 0, st, ea, tile_counts.p, (const uint32_t *)nullptr,
                               (uint64_t *)nullptr, (uint64_t *)nullptr);
        BMSP_CHECK_LAUNCH();
        DevBuf<uint32_t> tile_base((size_t)tiles + 1);
        HostScalar<uint32_t> n_tasks_h;
        device_exclusive_scan<uint32_t>(CountIn{tile_counts.p, tiles}, PtrOutTotal<uint32_t>{tile_base.p, tiles, n_tasks_h.dev()}, (uint64_t)tiles + 1, st);
        n_tasks = n_tasks_h.wait(st);
        tm.mark(3);
        k0.alloc(n_tasks); k1.alloc(n_tasks); v0.alloc(n_tasks); v1.alloc(n_tasks);
        if (small_product)
            hipLaunchKernelGGL((expand_filter_kernel<true, 2>), dim3(tiles), dim3(kThreads), 0, st, ea, (uint32_t *)nullptr, (const uint32_t *)tile_base.p, k0.p,
                               v0.p);
        else
            hipLaunchKernelGGL((expand_filter_kernel<true, kItems>), dim3(tiles), dim3(kThreads), 0, st, ea, (uint32_t *)nullptr, (const uint32_t *)tile_base.p,
                               k0.p, v0.p);
        BMSP_CHECK_LAUNCH();
        tm.mark(4);
    }
    S->surviving_tasks = (int64_t)n_tasks;
    S->bmp_reduction = (int64_t)(total - n_tasks);

    // T_5: group the tasks by C key
    PingPong<uint64_t> kk{k0.p, k1.p}, vv{v0.p, v1.p};
    // The tasks leave the expansion grouped by block-row of A (A's blocks are key-ordered), so only the column bits need
    // sorting inside each block-row segment.  The segmented sort keeps a whole segment in LDS (segsort.hip); it wins when
    // segments are long enough to fill a wave (cage-like rows: 289 vs 410 us; banded, 9 tasks per block-row: 92 vs 121 us) and -- since
    // round 4, with counting passes on the column bits for the hub rows of power-law operands -- whatever their longest is.  AUTO
    // decides on the average segment length; the reference decides on the task count alone (:963).
    const uint64_t a_block_rows = (uint64_t)A->num_block_rows();
    const uint64_t avg_seg = n_tasks / (a_block_rows ? a_block_rows : 1);
    // a block-row of C collects at most (most blocks in a block-row of A) x (most blocks in a block-row of B) tasks: when that fits a
    // wave's register sort, T_5 runs without a single read-back (per-matrix maxima, cached: ensure_row_stats)
    const uint64_t seg_bound = (uint64_t)std::max<int64_t>(A->max_row_blocks, 0) * (uint64_t)std::max<int64_t>(B->max_row_blocks, 0);
    // Segments beyond a wave's 4096 words: pieces + merge passes while two passes do (dense-tile ceiling, 4225 tasks per block-row: 2.9
    // vs 3.8 ms for the global radix sort), stable counting passes on the column bits beyond that (segsort.hip).  Before the counting
    // passes hub rows lost to the global sort (R-MAT 2^16, segments of 10^5 tasks, merge passes: 4.5 vs 2.8 ms) and AUTO kept products
    // of more than 2048 tasks per block-row away from the segmented path; with them it wins there too (R-MAT 2^16 x 8: T_5 1.70 vs 2.78 ms,
    // 2^20 x 2: 9.8 vs 20.2 ms).  BMSP_SEG_AVG_MAX restores a bound on the average for A/B runs.
    const char *sme = getenv("BMSP_SEG_AVG_MAX");
    const uint64_t seg_avg_max = sme ? (uint64_t)atoll(sme) : ~0ull;
    const bool try_segmented = mode == BMSP_SORT_SEGMENTED ||
                               (mode == BMSP_SORT_AUTO && avg_seg >= 4 && (avg_seg <= seg_avg_max || (seg_bound > 0 && seg_bound <= 16384)));
    S->sort_path = 0;
    if (n_tasks) {
        if (try_segmented && segsort_tasks_by_column(kk, vv, n_tasks, jbits, st, seg_bound, a_block_rows, &S->sort_long)) {
            S->sort_path = 1;
            tm.mark(8);
        } else {
            device_radix_sort_pairs<uint64_t>(kk, vv, n_tasks, 0, ibits + jbits, st);
        }
    }
    tm.mark(5);

    // T_6: C's block keys and the task range of every C block.  One scan: the keys and task ranges are emitted into
    // task-sized scratch and C's own key array is cut to size once the block count is known (a counting scan first, as the
    // reference's reduce_by_key does internally, would read the sorted keys twice more).
    HostScalar<uint32_t> csize_h;
    DevBuf<uint64_t> c_keys_scratch((size_t)n_tasks);
    task_begin.alloc((size_t)n_tasks + 1);
    c_of_wave.alloc((size_t)(n_tasks / 64 + 1));
    if (n_tasks) {
        device_exclusive_scan<uint32_t>(KeyHead{kk.cur, n_tasks}, EmitCBlocks{kk.cur, n_tasks, jbits, c_keys_scratch.p, task_begin.p, c_of_wave.p, csize_h.dev()},
                                        n_tasks + 1, st);
        c_size = csize_h.wait(st);
    }
    C->block_num = c_size;
    C->keys = (uint64_t *)pool_alloc(8 * (size_t)(c_size ? c_size : 1));
    C->bmps = (uint64_t *)pool_alloc(8 * (size_t)(c_size ? c_size : 1));
    if (c_size) BMSP_HIP(hipMemcpyAsync(C->keys, c_keys_scratch.p, 8 * (size_t)c_size, hipMemcpyDeviceToDevice, st));
    tm.mark(6);

    // T_9: C bitmaps, value offsets, nnz
    if (c_size) {
        device_for_each(ZeroU64{C->bmps}, (uint64_t)c_size, st);  // one launch (hipMemsetAsync splits into two fill kernels)
        hipLaunchKernelGGL(c_bitmaps_kernel, dim3((uint32_t)((n_tasks + 255) / 256)), dim3(kThreads), 0, st, kk.cur, vv.cur, (uint32_t)n_tasks, c_of_wave.p,
                           A->bmps, B->bmps, (unsigned long long *)C->bmps);
        BMSP_CHECK_LAUNCH();
    }
    finish_structure();
    tm.mark(9);
    tasks_sorted = vv.cur;
    }  // (expand - sort - compress)

    // T_7: block multiply-accumulate
    if (structure_only) {
        // bmsp_spgemm_symbolic: zero values, and C keeps the sorted task list so that bmsp_spgemm_numeric can run T_7 alone from it
        if (C->nnz) BMSP_HIP(hipMemsetAsync(C->values, 0, dtype_size(C->dtype) * (size_t)C->nnz, st));
        if (c_size && n_tasks) {
            C->sp_tasks = have_tasks ? rm_tasks.take() : (tasks_sorted == v0.p ? v0.take() : v1.take());
            C->sp_task_begin = task_begin.take();
            C->sp_c_of_wave = c_of_wave.take();
            C->sp_n_tasks = n_tasks; C->sp_candidates = total; C->sp_a_blocks = A->block_num; C->sp_b_blocks = B->block_num;
        }
    } else if (c_size) {
        launch_block_mac(A, B, C.get(), tasks_sorted, n_tasks, task_begin.p, c_of_wave.p, total, tc_version, mfma, S, st);
    }
    tm.mark(7);
    finish();
}

namespace {
struct StructDiff {
    const uint64_t *k0, *b0, *k1, *b1;
    uint32_t *flag;
    __device__ void operator()(uint64_t i) const
    {
        if (k0[i] != k1[i] || b0[i] != b1[i]) *flag = 1u;
    }
};
}  // namespace

// The numeric stage alone (bmsp_spgemm_numeric): C already holds the structure of A x B -- from bmsp_spgemm or bmsp_spgemm_symbolic on
// operands of the same structure -- and receives the values of A x B in place.  Where a strip kernel applies (it needs C's structure and
// the operands, no task list) that is T_7 and nothing else; otherwise the whole product runs and its values are copied over after its
// structure was checked against C's.
void spgemm_numeric(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s *C, int tc_version, hipStream_t st, bmsp_spgemm_stats *stats)
{
    if (!A || !B || !C) fail(BMSP_ERR_INVALID, "null argument");
    if (A->transposed || !B->transposed || C->transposed) fail(BMSP_ERR_INVALID, "layouts: A and C normal, B transposed");
    if (A->num_cols != B->num_rows || C->num_rows != A->num_rows || C->num_cols != B->num_cols) fail(BMSP_ERR_INVALID, "shape mismatch");
    if (A->dtype != B->dtype || C->dtype != (A->dtype == BMSP_F64 ? BMSP_F64 : BMSP_F32)) fail(BMSP_ERR_INVALID, "value types: A, B alike; C fp32 (fp64 for fp64 operands)");
    if (tc_version < 1 || tc_version > 5) fail(BMSP_ERR_INVALID, "tc_version must be 1..5");
    if (C->ownership == 2 && C->view_block_begin) fail(BMSP_ERR_UNSUPPORTED, "C must not be a row-panel view");
    bmsp_spgemm_stats local{};
    bmsp_spgemm_stats *S = stats ? stats : &local;
    *S = bmsp_spgemm_stats{};
    const bool mfma = tc_version != 5 && A->dtype == BMSP_F16;
    const bool strip_numeric = (tc_version == 4 && mfma && !getenv("BMSP_MAC_OLD")) || A->dtype == BMSP_F32 || (A->dtype == BMSP_F16 && tc_version == 5);
    const char *sf = getenv("BMSP_MAC_STRIP");
    // The kernels below write C's values from the operands' structure and trust C to hold the product's: they run only for a C stamped
    // with THESE operands' fingerprints (a product of bmsp_spgemm / _symbolic on operands of the same structure).  A stamped C of other
    // operands is refused; an unstamped one (adopted arrays, a product of panel views) goes through the checked path at the end.
    const bool stamped = C->sp_a_hash != 0 && C->sp_b_hash != 0;
    if (stamped && (C->sp_a_hash != ensure_struct_hash(A, st) || C->sp_b_hash != ensure_struct_hash(B, st)))
        fail(BMSP_ERR_INVALID, "C holds the structure of a product of other operands (its stamp does not match A and B)");
    if (stamped && C->block_num && strip_numeric && !(sf && sf[0] == '0') && mac_structure_numeric_ok(A, B, tc_version, st) && mac_strip_fits_c(C, st)) {
        StageTimer tm(st, true);
        tm.mark(-1);
        const int variant = launch_mac_strip(A, B, C, tc_version, st);
        tm.mark(7);
        BMSP_HIP(hipStreamSynchronize(st));
        S->t_us[0] = tm.collect(S->t_us);
        S->mac_variant = variant;
        S->mac_kernel = mfma ? tc_version : 5;
        S->c_blocks = C->block_num; S->c_nnz = C->nnz;
        return;
    }
    if (stamped && C->block_num && C->sp_tasks && C->sp_a_blocks == A->block_num && C->sp_b_blocks == B->block_num) {
        // C kept the product's sorted task list (bmsp_spgemm_symbolic): the block-MAC kernel of the tc_version runs from it
        StageTimer tm(st, true);
        tm.mark(-1);
        launch_block_mac(A, B, C, C->sp_tasks, C->sp_n_tasks, C->sp_task_begin, C->sp_c_of_wave, C->sp_candidates, tc_version, mfma, S, st);
        tm.mark(7);
        BMSP_HIP(hipStreamSynchronize(st));
        S->t_us[0] = tm.collect(S->t_us);
        S->surviving_tasks = (int64_t)C->sp_n_tasks; S->task_list_size = (int64_t)C->sp_candidates;
        S->c_blocks = C->block_num; S->c_nnz = C->nnz;
        return;
    }
    bmsp_matrix_s *full_raw = nullptr;
    spgemm(A, B, &full_raw, BMSP_SORT_AUTO, tc_version, 0, st, S);
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> full(full_raw, free_matrix);
    bool same = full->block_num == C->block_num && full->nnz == C->nnz;
    if (same && C->block_num) {
        DevBuf<uint32_t> flag(1);
        BMSP_HIP(hipMemsetAsync(flag.p, 0, 4, st));
        device_for_each(StructDiff{full->keys, full->bmps, C->keys, C->bmps, flag.p}, (uint64_t)C->block_num, st);
        same = read_back(flag.p, st) == 0u;
    }
    if (!same) fail(BMSP_ERR_INVALID, "C does not hold the structure of A x B (%lld blocks / %lld values expected)", (long long)full->block_num, (long long)full->nnz);
    if (C->nnz) BMSP_HIP(hipMemcpyAsync(C->values, full->values, dtype_size(C->dtype) * (size_t)C->nnz, hipMemcpyDeviceToDevice, st));
    BMSP_HIP(hipStreamSynchronize(st));
}

}  // namespace bmsp

BMSP_DEFINE_WARM(spgemm)
