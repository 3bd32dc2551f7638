// spmm.hip -- Y = A * X for a block of k vectors on the bmSparse format.
//
// Reference: none that runs.  bmSparse_SpMV takes a `batched` flag (src/bmSparse_SPMV.cu:191) and carries an unfinished
// multi-tile kernel (spmv_kernel_new, :84-150); the multi-vector product is the operation that flag points at and what
// CUSP's bytes_per_spmv_block accounts for (cusp/performance/spmv/bytes_per_spmv.h:42-50).  Numerics follow the SpMV:
// products in the accumulator type (float, double for F64), fused multiply-add, a row's tiles in key order.
//
// Layout: X is row-major num_cols x k (leading dimension ldx), Y row-major num_rows x k (ldy): the k values a stored
// element needs are contiguous, so lane j of a wave reads X[col][j] and the gather of one element is ONE coalesced
// request -- the tile metadata (24 B per tile) and the value are read once and amortised over k products.
//
// Work decomposition: the SpMV's cached sweep plan (spmv_plan.h).  One wave per item and per chunk of KK vectors;
// the 64 lanes are 64/KK tile slots x KK vectors: slot s walks tiles s, s+S, ... of a block-row, the slots' partial rows
// are folded with xor-shuffles and lane group 0 stores the block-row.  Hub block-rows are cut into 256-tile items by the
// plan; their partial rows go through a carry slot and the last wave to arrive folds them in item order (same
// write-through store + agent-scope counter hand-off as the SpMV, so the result does not depend on arrival order).
#include "spmv_plan.h"
#include <cstdlib>
#include "prims.hip.h"

namespace bmsp {
namespace {

__device__ __forceinline__ float fma_acc(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_acc(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int KK>
__global__ __launch_bounds__(kThreads) void spmm_kernel(const SweepItem *__restrict__ items, uint32_t num_items, const uint32_t *__restrict__ rowptr,
                                                        const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bmps,
                                                        const uint64_t *__restrict__ offsets, const T *__restrict__ values,
                                                        const T *__restrict__ X, typename Acc<T>::type *__restrict__ Y,
                                                        typename Acc<T>::type *__restrict__ carry, uint32_t *__restrict__ counters,
                                                        uint32_t num_rows, uint32_t num_cols, uint32_t k, uint64_t ldx, uint64_t ldy)
{
    using A = typename Acc<T>::type;
    constexpr int S = 64 / KK;  // tile slots per wave
    const int w = wave_id(), lane = lane_id();
    const uint32_t item_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
    if (item_id >= num_items) return;
    const SweepItem it = items[item_id];
    const uint32_t j0 = (uint32_t)lane % KK, s = (uint32_t)lane / KK;
    const uint32_t j = blockIdx.y * KK + j0;
    const bool jok = j < k;
    const bool is_long = it.num_items != 0;
    A acc[8];
    for (uint32_t br = it.row_begin; br < it.row_end; br++) {
        const uint32_t lo = is_long ? it.blk_begin : rowptr[br], hi = is_long ? it.blk_end : rowptr[br + 1];
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = A(0);
        for (uint32_t b = lo + s; b < hi; b += S) {
            const uint64_t bm = bmps[b];
            const T *vp = values + offsets[b];
            const uint32_t xb = key_col(keys[b]) * 8u;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t byte = tile_byte(bm, r);
                while (byte) {
                    const int c = __clz((int)byte) - 24;  // leading set bit of an 8-bit value -> column
                    byte &= ~(0x80u >> c);
                    const A a = (A)(*vp++);
                    const uint32_t col = xb + (uint32_t)c;
                    const A xv = (col < num_cols && jok) ? (A)X[(uint64_t)col * ldx + j] : A(0);
                    acc[r] = fma_acc(a, xv, acc[r]);
                }
            }
        }
        if (S > 1) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int d = KK; d < 64; d <<= 1) acc[r] += __shfl_xor(acc[r], d, kWave);
            }
        }
        if (!is_long && s == 0 && jok) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t row = br * 8u + (uint32_t)r;
                if (row < num_rows) Y[(uint64_t)row * ldy + j] = acc[r];
            }
        }
    }
    if (!is_long) return;
    // hub block-row: park the partial rows, the last arriver of this (row, vector chunk) folds them in item order
    const size_t slot = ((size_t)item_id * gridDim.y + blockIdx.y) * 8 * KK;
    if (s == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) __hip_atomic_store(&carry[slot + (size_t)r * KK + j0], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    uint32_t *cnt = counters + (size_t)it.long_idx * gridDim.y + blockIdx.y;
    if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != it.num_items - 1) return;
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = A(0);
    for (uint32_t c = s; c < it.num_items; c += S) {
        const size_t src = ((size_t)(it.first_item + c) * gridDim.y + blockIdx.y) * 8 * KK;
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] += __hip_atomic_load(&carry[src + (size_t)r * KK + j0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (S > 1) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int d = KK; d < 64; d <<= 1) acc[r] += __shfl_xor(acc[r], d, kWave);
        }
    }
    if (s == 0 && jok) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t row = it.row_begin * 8u + (uint32_t)r;
            if (row < num_rows) Y[(uint64_t)row * ldy + j] = acc[r];
        }
    }
}

// ---- k > 16: one lane per vector, tiles walked by the whole wave ----------------------------------------------------
// The slot kernel above keeps 64/KK tiles in flight per wave, which is one for KK = 64: every tile then costs two dependent
// round trips (its words, then its values and X rows).  Here the tile words are fetched lane-parallel, 64 tiles per batch
// (three coalesced streams, as in the SpMV sweep); each lane decodes its tile's stored elements into a flat ELEMENT STREAM in
// LDS (column, value index, row; positions from a wave prefix sum of the popcounts), and the wave then walks the stream sixteen
// elements at a time: sixteen X rows (lane j reads X[col][j], one coalesced request per element) and sixteen values are
// requested before the first product, whatever tiles the elements came from.  Block-rows arrive in order, so eight row accumulators per
// lane suffice; they are stored whenever the stream moves to the next block-row.
template <typename A>
__device__ __forceinline__ void acc_row(A (&acc)[8], uint32_t r, A a, A x)
{
    switch (r) {  // r is wave-uniform: a scalar branch, each arm one FMA on a fixed register
    case 0: acc[0] = fma_acc(a, x, acc[0]); break;
    case 1: acc[1] = fma_acc(a, x, acc[1]); break;
    case 2: acc[2] = fma_acc(a, x, acc[2]); break;
    case 3: acc[3] = fma_acc(a, x, acc[3]); break;
    case 4: acc[4] = fma_acc(a, x, acc[4]); break;
    case 5: acc[5] = fma_acc(a, x, acc[5]); break;
    case 6: acc[6] = fma_acc(a, x, acc[6]); break;
    default: acc[7] = fma_acc(a, x, acc[7]); break;
    }
}

constexpr uint32_t kStreamCap = 512;  // stored elements decoded per pass of the element stream

struct StreamEntry {  // 16 bytes: one ds_read_b128
    uint32_t col, vidx, rowid, pad;  // column of X, index into values, (block-row << 3) | tile row
};

template <typename T>
__global__ __launch_bounds__(kThreads) void spmm_wide_kernel(const SweepItem *__restrict__ items, uint32_t num_items, const uint64_t *__restrict__ keys,
                                                             const uint64_t *__restrict__ bmps, const uint64_t *__restrict__ offsets,
                                                             const T *__restrict__ values, const T *__restrict__ X,
                                                             typename Acc<T>::type *__restrict__ Y, typename Acc<T>::type *__restrict__ carry,
                                                             uint32_t *__restrict__ counters, uint32_t num_rows, uint32_t num_cols, uint32_t k,
                                                             uint64_t ldx, uint64_t ldy)
{
    using A = typename Acc<T>::type;
    constexpr int KK = 64;
    __shared__ StreamEntry s_entries[4][kStreamCap];
    const int w = wave_id(), lane = lane_id();
    const uint32_t item_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
    if (item_id >= num_items) return;
    StreamEntry *ent = s_entries[w];
    const SweepItem it = items[item_id];
    const uint32_t j = blockIdx.y * KK + (uint32_t)lane;
    const bool jok = j < k;
    const bool is_long = it.num_items != 0;
    A acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = A(0);
    uint32_t cur_row = it.row_begin;
    // stores block-row cur_row and zero-fills the block-rows up to (not including) next_row -- short items own their rows
    auto flush_to = [&](uint32_t next_row) {
        if (!is_long && jok) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t row = cur_row * 8u + (uint32_t)r;
                if (row < num_rows) Y[(uint64_t)row * ldy + j] = acc[r];
            }
            for (uint32_t br = cur_row + 1; br < next_row; br++) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t row = br * 8u + (uint32_t)r;
                    if (row < num_rows) Y[(uint64_t)row * ldy + j] = A(0);
                }
            }
        }
    };
    for (uint32_t base = it.blk_begin; base < it.blk_end; base += 64) {
        // lane-per-tile: the three streamed words, and the tile's place in the element stream
        const uint32_t b = base + (uint32_t)lane;
        uint64_t bm = 0, key = (uint64_t)it.row_begin << 32;
        uint32_t off = 0;
        if (b < it.blk_end) { bm = bmps[b]; key = keys[b]; off = (uint32_t)offsets[b]; }
        const uint32_t cnt = (uint32_t)__popcll(bm);
        uint32_t incl = cnt;  // inclusive prefix sum over lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t q = (uint32_t)__shfl_up((int)incl, d, kWave);
            if (lane >= d) incl += q;
        }
        const uint32_t excl = incl - cnt;
        uint32_t first_lane = 0;  // tiles [first_lane, ...) still to be streamed
        while (first_lane < 64u) {
            // the longest run of tiles from first_lane whose elements fit the staging buffer (one tile always fits)
            const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)first_lane);
            const uint64_t fits = __ballot(incl - start <= kStreamCap && (uint32_t)lane >= first_lane);
            const uint32_t end_lane = first_lane + (uint32_t)__popcll(fits);  // fits is a contiguous run (incl is monotone)
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(end_lane - 1)) - start;
            __builtin_amdgcn_wave_barrier();
            if ((uint32_t)lane >= first_lane && (uint32_t)lane < end_lane) {
                uint64_t m = bm;
                uint32_t e = excl - start, v = off;
                const uint32_t rbase = key_row(key) << 3, cbase = key_col(key) * 8u;
                while (m) {
                    const uint32_t p = (uint32_t)__clzll((long long)m);
                    m &= ~(0x8000000000000000ull >> p);
                    ent[e++] = StreamEntry{cbase + (p & 7u), v++, rbase | (p >> 3), 0u};
                }
            }
            __builtin_amdgcn_wave_barrier();
            // the stream: kChunk X rows and values requested before the first product.  (Overlapping the requests of the next
            // chunk with the products of this one was measured slower: with the row-change stores in the product phase the
            // compiler's wait-count analysis falls back to draining every load, and the code doubles.)
            constexpr int kChunk = 16;
            for (uint32_t e0 = 0; e0 < total; e0 += kChunk) {
                A av[kChunk], xv[kChunk];
                uint32_t rid[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    if (e0 + u < total) {
                        const StreamEntry en = ent[e0 + u];  // same address in every lane: an LDS broadcast
                        rid[u] = en.rowid;
                        av[u] = (A)values[en.vidx];
                        xv[u] = (en.col < num_cols && jok) ? (A)X[(uint64_t)en.col * ldx + j] : A(0);
                    }
                }
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    if (e0 + u < total) {
                        const uint32_t rowid = (uint32_t)__builtin_amdgcn_readfirstlane((int)rid[u]);
                        if ((rowid >> 3) != cur_row) {
                            flush_to(rowid >> 3);
                            cur_row = rowid >> 3;
#pragma unroll
                            for (int r = 0; r < 8; r++) acc[r] = A(0);
                        }
                        acc_row<A>(acc, rowid & 7u, av[u], xv[u]);
                    }
                }
            }
            first_lane = end_lane;
        }
    }
    flush_to(it.row_end);
    if (!is_long) return;
    // hub block-row: park the partial rows, the last arriver of this (row, vector chunk) folds them in item order
    const size_t slot = ((size_t)item_id * gridDim.y + blockIdx.y) * 8 * KK;
#pragma unroll
    for (int r = 0; r < 8; r++) __hip_atomic_store(&carry[slot + (size_t)r * KK + lane], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    uint32_t *cnt_p = counters + (size_t)it.long_idx * gridDim.y + blockIdx.y;
    if (lane == 0) ticket = __hip_atomic_fetch_add(cnt_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != it.num_items - 1) return;
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = A(0);
    for (uint32_t c = 0; c < it.num_items; c++) {
        const size_t src = ((size_t)(it.first_item + c) * gridDim.y + blockIdx.y) * 8 * KK;
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] += __hip_atomic_load(&carry[src + (size_t)r * KK + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (jok) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t row = it.row_begin * 8u + (uint32_t)r;
            if (row < num_rows) Y[(uint64_t)row * ldy + j] = acc[r];
        }
    }
}

// ---- k <= 8 on matrices that carry the SpMV position cache: the value-stream walk of spmv.hip, KK products per stored value ---------
// One wave (= one workgroup) per plan item and chunk of KK vectors.  The tile phase and the batch cut are those of spmv_vstream_kernel
// (the cached entries number their tile slots inside exactly these batches: <= 128 tiles holding <= 512 values); the value phase loads, per
// stored value, its entry, its value (both coalesced) and KK consecutive X entries of its column (row-major X: one 16-byte request per
// four vectors when X allows it), and adds the KK products into the item's u tile (16 block-rows x 8 rows x KK) with LDS float adds.
// The slot kernel above walks a tile's elements serially per lane group; here every lane of every request has an element.
typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t spmm_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
constexpr uint32_t kSmTiles = 128, kSmVals = 512;  // the batch cut of the position cache (spmv.hip: kVsTiles, kVsVals)

template <typename T, int KK>
__global__ __launch_bounds__(64) void spmm_vstream_kernel(const SweepItem *__restrict__ items, const uint64_t *__restrict__ keys,
                                                          const uint64_t *__restrict__ offsets, const T *__restrict__ values,
                                                          const uint16_t *__restrict__ pos, uint32_t pos_base, uint32_t pos_count,
                                                          uint32_t values_bytes, const T *__restrict__ X, typename Acc<T>::type *__restrict__ Y,
                                                          typename Acc<T>::type *__restrict__ carry, uint32_t *__restrict__ counters,
                                                          uint32_t num_rows, uint32_t num_cols, uint32_t k, uint64_t ldx, uint64_t ldy, int x_vec)
{
    using A = typename Acc<T>::type;
    constexpr uint32_t kOobS = 0xffffffffu;
    __shared__ A tile[kItemRows * 8 * KK];
    __shared__ uint32_t tinfo[kSmTiles];
    const int lane = lane_id();
    const uint32_t item_id = blockIdx.x, j0 = blockIdx.y * KK;
    const SweepItem it = items[item_id];
    const __amdgpu_buffer_rsrc_t rp = spmm_rsrc(pos, pos_count * 2u), rv = spmm_rsrc(values, values_bytes);
    for (uint32_t e = (uint32_t)lane; e < kItemRows * 8 * KK; e += 64) tile[e] = A(0);

    for (uint32_t base = it.blk_begin; base < it.blk_end;) {
        const uint32_t b0 = base + (uint32_t)lane, b1 = b0 + 64u, bend = min(base + kSmTiles, it.blk_end);
        uint64_t k0 = (uint64_t)it.row_begin << 32, k1 = k0;
        uint32_t e0 = kOobS, e1 = kOobS;
        const uint32_t v_first = (uint32_t)offsets[base];
        if (b0 < bend) { k0 = keys[b0]; e0 = (uint32_t)offsets[b0 + 1] - v_first; }
        if (b1 < bend) { k1 = keys[b1]; e1 = (uint32_t)offsets[b1 + 1] - v_first; }
        const bool ok0 = b0 < bend && e0 <= kSmVals, ok1 = b1 < bend && e1 <= kSmVals;
        const uint32_t nb = (uint32_t)__popcll(__ballot(ok0)) + (uint32_t)__popcll(__ballot(ok1));
        const uint32_t nvals = nb <= 64u ? (uint32_t)__builtin_amdgcn_readlane((int)e0, (int)(nb - 1u)) : (uint32_t)__builtin_amdgcn_readlane((int)e1, (int)(nb - 65u));
        tinfo[lane] = key_col(k0) | ((key_row(k0) - it.row_begin) << 28);
        tinfo[64 + lane] = key_col(k1) | ((key_row(k1) - it.row_begin) << 28);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c0 = 0; c0 < nvals; c0 += 64u) {
            const uint32_t idx = c0 + (uint32_t)lane;
            const bool on = idx < nvals;
            const uint32_t e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rp, on ? (v_first - pos_base + idx) * 2u : kOobS, 0, 0);
            A av;
            if (sizeof(T) == 4) av = (A)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, on ? (v_first + idx) * 4u : kOobS, 0, 0));
            else av = on ? (A)values[v_first + idx] : A(0);
            const uint32_t ti = tinfo[e >> 6], p = e & 63u;
            const uint32_t row = (ti >> 28) * 8u + (p >> 3), col = (ti & 0x0fffffffu) * 8u + (p & 7u);
            const bool live = on && col < num_cols;
            A xv[KK];
            const T *xr = X + (uint64_t)col * ldx + j0;
            if (sizeof(T) == 4 && x_vec) {
                // X is 16-byte aligned with ldx a multiple of 4 floats, and the chunk's vectors all exist: one request per four vectors
                const float4 *xq = reinterpret_cast<const float4 *>(xr);
#pragma unroll
                for (int q = 0; q < KK / 4; q++) {
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (live) w = xq[q];
                    xv[4 * q + 0] = (A)w.x;
                    xv[4 * q + 1] = (A)w.y;
                    xv[4 * q + 2] = (A)w.z;
                    xv[4 * q + 3] = (A)w.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < KK; q++) xv[q] = (live && j0 + (uint32_t)q < k) ? (A)xr[q] : A(0);
            }
            if (on) {
#pragma unroll
                for (int q = 0; q < KK; q++) __hip_atomic_fetch_add(&tile[row * KK + (uint32_t)q], av * xv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        __builtin_amdgcn_wave_barrier();
        base += nb;
    }

    if (it.num_items == 0) {
        const uint32_t n_out = (it.row_end - it.row_begin) * 8u * KK, out0 = it.row_begin * 8u;
        for (uint32_t e = (uint32_t)lane; e < n_out; e += 64) {
            const uint32_t row = out0 + e / KK, j = j0 + e % KK;
            if (row < num_rows && j < k) Y[(uint64_t)row * ldy + j] = tile[e];
        }
        return;
    }
    // hub block-row: park the 8 x KK partial sums, the last arriver of this (row, vector chunk) folds them in item order
    const size_t slot = ((size_t)item_id * gridDim.y + blockIdx.y) * 8 * KK;
    for (uint32_t e = (uint32_t)lane; e < 8u * KK; e += 64) __hip_atomic_store(&carry[slot + e], tile[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    uint32_t *cnt = counters + (size_t)it.long_idx * gridDim.y + blockIdx.y;
    if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != it.num_items - 1) return;
    for (uint32_t e = (uint32_t)lane; e < 8u * KK; e += 64) {
        A sum = A(0);
        for (uint32_t c = 0; c < it.num_items; c++)
            sum += __hip_atomic_load(&carry[((size_t)(it.first_item + c) * gridDim.y + blockIdx.y) * 8 * KK + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t row = it.row_begin * 8u + e / KK, j = j0 + e % KK;
        if (row < num_rows && j < k) Y[(uint64_t)row * ldy + j] = sum;
    }
}

template <typename T, int KK>
void launch_vstream(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st)
{
    using Ac = typename Acc<T>::type;
    const uint32_t n_items = (uint32_t)A->spmv_num_chunks;
    const uint32_t chunks = (uint32_t)((k + KK - 1) / KK);
    DevBuf<Ac> carry(A->spmv_plan_long ? (size_t)n_items * chunks * 8 * KK : 1);
    DevBuf<uint32_t> counters((size_t)(A->spmv_plan_long ? A->spmv_plan_long : 1) * chunks);
    BMSP_HIP(hipMemsetAsync(counters.p, 0, 4 * counters.n, st));
    const int x_vec = sizeof(T) == 4 && ((uintptr_t)X & 15u) == 0 && ldx % 4 == 0 && k % KK == 0;
    hipLaunchKernelGGL((spmm_vstream_kernel<T, KK>), dim3(n_items, chunks), dim3(64), 0, st, plan_items(A), A->keys, A->offsets, (const T *)A->values,
                       A->spmv_pos, (uint32_t)A->spmv_pos_base, (uint32_t)A->spmv_pos_count, (uint32_t)((size_t)A->values_extent() * sizeof(T)),
                       (const T *)X, (Ac *)Y, carry.p, counters.p, (uint32_t)A->num_rows, (uint32_t)A->num_cols, (uint32_t)k, (uint64_t)ldx,
                       (uint64_t)ldy, x_vec);
    BMSP_CHECK_LAUNCH();
    if (A->spmv_plan_long) BMSP_HIP(hipStreamSynchronize(st));  // the carry slots go back to the pool on return
}

template <typename T>
void launch_wide(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st)
{
    using Ac = typename Acc<T>::type;
    const uint32_t n_items = (uint32_t)A->spmv_num_chunks;
    const uint32_t chunks = (uint32_t)((k + 63) / 64);
    DevBuf<Ac> carry(A->spmv_plan_long ? (size_t)n_items * chunks * 8 * 64 : 1);
    DevBuf<uint32_t> counters((size_t)(A->spmv_plan_long ? A->spmv_plan_long : 1) * chunks);
    BMSP_HIP(hipMemsetAsync(counters.p, 0, 4 * counters.n, st));
    hipLaunchKernelGGL((spmm_wide_kernel<T>), dim3((n_items + 3) / 4, chunks), dim3(kThreads), 0, st, plan_items(A), n_items, A->keys, A->bmps, A->offsets,
                       (const T *)A->values, (const T *)X, (Ac *)Y, carry.p, counters.p, (uint32_t)A->num_rows, (uint32_t)A->num_cols, (uint32_t)k,
                       (uint64_t)ldx, (uint64_t)ldy);
    BMSP_CHECK_LAUNCH();
    if (A->spmv_plan_long) BMSP_HIP(hipStreamSynchronize(st));  // the carry slots go back to the pool on return
}

template <typename T, int KK>
void launch_kk(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st)
{
    using Ac = typename Acc<T>::type;
    const uint32_t n_items = (uint32_t)A->spmv_num_chunks;
    const uint32_t chunks = (uint32_t)((k + KK - 1) / KK);
    // the plan's own carry / counters serve the single-vector sweep; a k-wide product needs k-wide slots
    DevBuf<Ac> carry(A->spmv_plan_long ? (size_t)n_items * chunks * 8 * KK : 1);
    DevBuf<uint32_t> counters((size_t)(A->spmv_plan_long ? A->spmv_plan_long : 1) * chunks);
    BMSP_HIP(hipMemsetAsync(counters.p, 0, 4 * counters.n, st));
    hipLaunchKernelGGL((spmm_kernel<T, KK>), dim3((n_items + 3) / 4, chunks), dim3(kThreads), 0, st, plan_items(A), n_items, A->rowptr, A->keys, A->bmps,
                       A->offsets, (const T *)A->values, (const T *)X, (Ac *)Y, carry.p, counters.p, (uint32_t)A->num_rows, (uint32_t)A->num_cols,
                       (uint32_t)k, (uint64_t)ldx, (uint64_t)ldy);
    BMSP_CHECK_LAUNCH();
    if (A->spmv_plan_long) BMSP_HIP(hipStreamSynchronize(st));  // the carry slots go back to the pool on return
}

template <typename T>
void launch(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st)
{
    // measured on the webbase-1M-like case: the time is set by tiles x vector chunks (each wave walks its tiles
    // serially), so the widest lane group that k fills wins
    // k <= 8 on a matrix that carries the SpMV position cache (sparse tiles): the value-stream walk
    if (A->spmv_pos && k <= 8 && !getenv("BMSP_SPMM_NO_VSTREAM")) {
        if (k <= 4) launch_vstream<T, 4>(A, X, ldx, Y, ldy, k, st);
        else launch_vstream<T, 8>(A, X, ldx, Y, ldy, k, st);
        return;
    }
    if (k <= 4) launch_kk<T, 4>(A, X, ldx, Y, ldy, k, st);
    else if (k <= 16) launch_kk<T, 16>(A, X, ldx, Y, ldy, k, st);
    else if ((uint64_t)A->values_extent() < (1ull << 32)) launch_wide<T>(A, X, ldx, Y, ldy, k, st);  // 32-bit value indices in the stream
    else launch_kk<T, 64>(A, X, ldx, Y, ldy, k, st);
}

}  // namespace

void spmm(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st)
{
    if (A->transposed) fail(BMSP_ERR_INVALID, "SpMM needs a matrix built with transposed=0");
    if (k < 1) fail(BMSP_ERR_INVALID, "k must be >= 1");
    if (ldx < k || ldy < k) fail(BMSP_ERR_INVALID, "leading dimensions must be >= k");
    if (A->num_rows == 0) return;
    if (k == 1 && ldx == 1 && ldy == 1) {  // one contiguous vector: the SpMV itself
        spmv(A, X, Y, BMSP_SPMV_DEFAULT, st);
        return;
    }
    ensure_rowptr(A, st);
    build_plan(A, st);
    prepare_spmv(A, st);  // + the position cache, for matrices the value-stream kernels take
    switch (A->dtype) {
    case BMSP_F32: launch<float>(A, X, ldx, Y, ldy, k, st); break;
    case BMSP_F16: launch<_Float16>(A, X, ldx, Y, ldy, k, st); break;
    case BMSP_F64: launch<double>(A, X, ldx, Y, ldy, k, st); break;
    default: fail(BMSP_ERR_INVALID, "unknown dtype");
    }
}

}  // namespace bmsp

BMSP_DEFINE_WARM(spmm)
