// spmv.hip -- u = A * v on the bmSparse format, gfx950 / wave64.
//
// Reference: bmSparse_SpMV<VI,VO>, src/bmSparse_SPMV.cu:191-230 with spmv_kernel (:153-189, one 64-thread
// block per block-row, one lane per tile ELEMENT, serial loop over the row's tiles) and spmv_kernel_new
// (:84-150, the "batched" path: several tiles per step, wide lane reduction).
//
// MI355X design: the unit of work is a tile ROW (one byte of the bitmap).  A lane owns row r of a tile:
// it takes byte r of the bitmap, ranks it with one popcount of the bits in front of it, and walks the set
// bits of that byte, multiplying consecutive values by the matching x entries.  Tiles are 1-8 % dense on
// graph matrices, so a lane per element (the reference's mapping) leaves >90 % of a wave64 idle, while a
// lane per tile row keeps the value loads of a tile contiguous across the 8 lanes of a group.
//   variant 0: an 8-lane group sweeps one block-row, no cross-lane traffic at all;
//   variant 1: a whole wave sweeps one block-row 8 tiles at a time and folds the eight partial rows with
//              xor-shuffles (wavefront reduction) -- for block-rows with many tiles.
// The dense block-row pointer is built once per matrix (builder.hip), not per call as the reference does.
#include "matrix.h"
#include "prims.hip.h"

namespace bmsp {
namespace {

template <typename T>
struct Acc { using type = float; };
template <>
struct Acc<double> { using type = double; };

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
        A xv = col < num_cols ? (A)x[col] : A(0);  // ragged last block column (reference reads past the end)
        acc = __builtin_fma(a, xv, acc);
    }
    return acc;
}

template <typename T, int LANES_PER_ROW>
__global__ __launch_bounds__(kThreads) void spmv_sweep_kernel(const uint32_t *__restrict__ rowptr, const uint64_t *__restrict__ keys,
                                                              const uint64_t *__restrict__ bmps, const uint64_t *__restrict__ offsets,
                                                              const T *__restrict__ values, const T *__restrict__ x,
                                                              typename Acc<T>::type *__restrict__ y, uint32_t num_rows,
                                                              uint32_t num_cols, uint32_t num_block_rows)
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
    if (g == 0 && row < num_rows) y[row] = acc;
}

template <typename T>
void launch(bmsp_matrix_s *A, const void *v, void *u, int variant, hipStream_t st)
{
    using Ac = typename Acc<T>::type;
    uint32_t nbr = (uint32_t)A->num_block_rows();
    if (nbr == 0) return;
    if (variant == BMSP_SPMV_BATCHED) {
        hipLaunchKernelGGL((spmv_sweep_kernel<T, 64>), dim3((nbr + 3) / 4), dim3(kThreads), 0, st, A->rowptr, A->keys, A->bmps,
                           A->offsets, (const T *)A->values, (const T *)v, (Ac *)u, (uint32_t)A->num_rows,
                           (uint32_t)A->num_cols, nbr);
    } else {
        hipLaunchKernelGGL((spmv_sweep_kernel<T, 8>), dim3((nbr + 31) / 32), dim3(kThreads), 0, st, A->rowptr, A->keys, A->bmps,
                           A->offsets, (const T *)A->values, (const T *)v, (Ac *)u, (uint32_t)A->num_rows,
                           (uint32_t)A->num_cols, nbr);
    }
    BMSP_CHECK_LAUNCH();
}

}  // namespace

void spmv(bmsp_matrix_s *A, const void *v, void *u, int variant, hipStream_t st)
{
    if (A->transposed) fail(BMSP_ERR_INVALID, "SpMV needs a matrix built with transposed=0");
    if (!v || !u) fail(BMSP_ERR_INVALID, "null vector");
    ensure_rowptr(A, st);
    switch (A->dtype) {
    case BMSP_F32: launch<float>(A, v, u, variant, st); break;
    case BMSP_F16: launch<_Float16>(A, v, u, variant, st); break;
    case BMSP_F64: launch<double>(A, v, u, variant, st); break;
    }
}

}  // namespace bmsp
