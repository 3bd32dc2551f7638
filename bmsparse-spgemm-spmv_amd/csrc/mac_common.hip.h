// mac_common.hip.h -- device helpers shared by the block multiply-accumulate kernels (spgemm.hip, blockmac32.hip):
// buffer-resource loads, the packed block record, and the nibble decode of a compact tile line.
#ifndef BMSP_MAC_COMMON_HIP_H_
#define BMSP_MAC_COMMON_HIP_H_
#include "matrix.h"
#include "prims.hip.h"

namespace bmsp {
namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// buffer (SRSRC) loads: 32-bit byte offsets against a wave-uniform descriptor; an out-of-range offset reads 0, which
// turns the "only if the bit is set" gathers of a tile nibble into straight-line code
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t kOob = 0xffffffffu;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ _Float16 ld_half(rsrc_t r, uint32_t byte_off)
{
    return __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0));
}

// tile line `line` (row of A / column of B), k = 4*khalf .. 4*khalf+3  -> positions line*8 + 4*khalf + q; the stored ones
// are consecutive in the value array starting at rank(first position)
__device__ __forceinline__ half4_t load_nibble(uint64_t bmp, rsrc_t vals, uint32_t tile_byte_off, int line, int khalf, bool live)
{
    const int p0 = line * 8 + khalf * 4;
    const uint32_t nib = live ? (uint32_t)(bmp >> (60 - p0)) & 0xfu : 0u;  // bit 3 = k+0 ... bit 0 = k+3
    uint32_t off = tile_byte_off + (uint32_t)tile_rank(bmp, p0) * 2u;
    half4_t r;
    r[0] = ld_half(vals, (nib & 8u) ? off : kOob); off += (nib & 8u) ? 2u : 0u;
    r[1] = ld_half(vals, (nib & 4u) ? off : kOob); off += (nib & 4u) ? 2u : 0u;
    r[2] = ld_half(vals, (nib & 2u) ? off : kOob); off += (nib & 2u) ? 2u : 0u;
    r[3] = ld_half(vals, (nib & 1u) ? off : kOob);
    return r;
}

// Wide variant: the (up to four) stored values of a nibble are consecutive halves, so ONE 12-byte load from the enclosing
// aligned dwords fetches them all; a funnel shift drops the odd leading half and v_perm_b32 routes value `rank` to lane slot
// `q` (or zero) with a selector looked up by nibble pattern.  ~10 VALU + 3 VMEM fewer than load_nibble per operand.
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));

struct NibbleLane {   // per-lane constants of the operand position p0 = line*8 + khalf*4
    uint32_t use_hi;  // the nibble lives in the high word of the bitmap
    uint32_t shift;   // its shift inside that word
    uint32_t hi_mask, lo_mask;  // bitmap bits of positions < p0
};
__device__ __forceinline__ NibbleLane make_nibble_lane(int line, int khalf)
{
    const uint32_t p0 = (uint32_t)(line * 8 + khalf * 4);
    NibbleLane n;
    n.use_hi = p0 < 32u;
    n.shift = (60u - p0) & 31u;
    n.hi_mask = p0 >= 32u ? 0xffffffffu : (p0 == 0u ? 0u : 0xffffffffu << (32u - p0));
    n.lo_mask = p0 > 32u ? 0xffffffffu << (64u - p0) : 0u;
    return n;
}
// selector pair for nibble pattern `nib` (bit 3 = first position): output half q takes stored value popc(bits before q)
__device__ __forceinline__ uint64_t nibble_selector(uint32_t nib)
{
    uint64_t sel = 0;
    uint32_t rank = 0;
    for (int q = 0; q < 4; q++) {
        const bool has = (nib >> (3 - q)) & 1u;
        const uint64_t two = has ? (uint64_t)((2u * rank) | ((2u * rank + 1u) << 8)) : 0x0c0cull;
        sel |= two << (16 * q);
        rank += has;
    }
    return sel;
}
__device__ __forceinline__ half4_t load_nibble_wide(uint64_t bmp, rsrc_t vals, uint32_t tile_byte_off, const NibbleLane &n, const uint64_t *sel_table, bool live)
{
    const uint32_t hi = (uint32_t)(bmp >> 32), lo = (uint32_t)bmp;
    const uint32_t nib = live ? ((n.use_hi ? hi : lo) >> n.shift) & 0xfu : 0u;
    const uint32_t rank = (uint32_t)__builtin_popcount(hi & n.hi_mask) + (uint32_t)__builtin_popcount(lo & n.lo_mask);
    const uint32_t addr = tile_byte_off + rank * 2u;
    const u32x3_t d = __builtin_amdgcn_raw_buffer_load_b96(vals, nib ? (addr & ~3u) : kOob, 0, 0);
    const uint64_t sel = sel_table[nib];
    const uint32_t sh = (addr & 2u) * 8u;
    const uint32_t v01 = __builtin_amdgcn_alignbit(d[1], d[0], sh), v23 = __builtin_amdgcn_alignbit(d[2], d[1], sh);
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t r;
    r[0] = __builtin_amdgcn_perm(v23, v01, (uint32_t)sel);
    r[1] = __builtin_amdgcn_perm(v23, v01, (uint32_t)(sel >> 32));
    return __builtin_bit_cast(half4_t, r);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_block_meta(rsrc_t r, uint32_t block, uint64_t &bmp, uint32_t &off_elems)
{
    const u32x4_t m = __builtin_amdgcn_raw_buffer_load_b128(r, block << 4, 0, 0);
    bmp = ((uint64_t)m[1] << 32) | m[0];
    off_elems = m[2];
}


}  // namespace
}  // namespace bmsp
#endif
