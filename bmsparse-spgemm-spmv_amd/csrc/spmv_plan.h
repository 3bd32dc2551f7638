// spmv_plan.h -- the cached sweep plan shared by the SpMV and SpMM kernels (see spmv.hip for how it is built and used).
#ifndef BMSP_SPMV_PLAN_H_
#define BMSP_SPMV_PLAN_H_
#include "matrix.h"

namespace bmsp {

template <typename T>
struct Acc { using type = float; };
template <>
struct Acc<double> { using type = double; };

constexpr uint32_t kItemTiles = 256;  // tile budget of an item (short items hold < 2x this, long-row items exactly this)
constexpr uint32_t kBatch = 128;      // tiles a wave loads at once (two per lane)
constexpr uint32_t kItemRows = 16;    // block-rows per item window (u tile = 16 x 8 accumulators)

struct SweepItem {      // 32 bytes, read with scalar loads
    uint32_t row_begin, row_end;  // block-rows [row_begin, row_end)
    uint32_t blk_begin, blk_end;  // tiles [blk_begin, blk_end)
    uint32_t first_item;          // long rows: index of the row's first item; short items: offset of the first value past the item
    uint32_t num_items;           // long rows: number of items of the row; 0 = short item
    uint32_t long_idx;            // long rows: arrival counter index
    uint32_t val_begin;           // offset of the item's first stored value (low 32 bits)
};

// builds (once) and caches the plan in A->spmv_chunks: 64-byte header | items | counters | carry
void build_plan(bmsp_matrix_s *A, hipStream_t st);
inline const SweepItem *plan_items(const bmsp_matrix_s *A) { return (const SweepItem *)((const char *)A->spmv_chunks + 64); }

}  // namespace bmsp
#endif
