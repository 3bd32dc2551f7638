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
    if (k <= 4) launch_kk<T, 4>(A, X, ldx, Y, ldy, k, st);
    else if (k <= 16) launch_kk<T, 16>(A, X, ldx, Y, ldy, k, st);
    else launch_kk<T, 64>(A, X, ldx, Y, ldy, k, st);
}

}  // namespace

void spmm(bmsp_matrix_s *A, const void *X, int64_t ldx, void *Y, int64_t ldy, int k, hipStream_t st)
{
    if (A->transposed) fail(BMSP_ERR_INVALID, "SpMM needs a matrix built with transposed=0");
    if (k < 1) fail(BMSP_ERR_INVALID, "k must be >= 1");
    if (ldx < k || ldy < k) fail(BMSP_ERR_INVALID, "leading dimensions must be >= k");
    if (A->num_rows == 0) return;
    ensure_rowptr(A, st);
    build_plan(A, st);
    switch (A->dtype) {
    case BMSP_F32: launch<float>(A, X, ldx, Y, ldy, k, st); break;
    case BMSP_F16: launch<_Float16>(A, X, ldx, Y, ldy, k, st); break;
    case BMSP_F64: launch<double>(A, X, ldx, Y, ldy, k, st); break;
    default: fail(BMSP_ERR_INVALID, "unknown dtype");
    }
}

}  // namespace bmsp
