// bmsp_bits.h -- bit-level definitions of the bmSparse tile format, usable from host and device code.
//
// Format (reference: src/bmSpMatrix.cu:76-101, src/bmSparse_SPMV.cu:72-82):
//   block key   = (block_row << 32) | block_col
//   bitmap      = 64 bits, bit (63 - pos) set when tile position pos holds a value,
//                 pos = 8*r + c (normal) or 8*c + r (transposed build)
//   values      = compacted per tile in ascending pos; element pos lives at offsets[b] + rank(bmp, pos)
#ifndef BMSP_BITS_H_
#define BMSP_BITS_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define BMSP_HD __host__ __device__ __forceinline__
#else
#define BMSP_HD inline
#endif

namespace bmsp {

BMSP_HD int popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

BMSP_HD uint64_t key_make(uint32_t brow, uint32_t bcol) { return ((uint64_t)brow << 32) | (uint64_t)bcol; }
BMSP_HD uint32_t key_row(uint64_t k) { return (uint32_t)(k >> 32); }
BMSP_HD uint32_t key_col(uint64_t k) { return (uint32_t)(k & 0xffffffffull); }

// number of stored values in front of tile position p (p in [0,64)).  The reference computes
// popc(bmp >> (64 - p)), which is undefined at p == 0 (src/bmSparse_SPMV.cu:77); defined here as 0.
BMSP_HD int tile_rank(uint64_t bmp, int p) { return p ? popc64(bmp >> (64 - p)) : 0; }
BMSP_HD bool tile_has(uint64_t bmp, int p) { return (bmp >> (63 - p)) & 1ull; }

// byte i (0 = most significant) of a bitmap: row i of a normal tile, column i of a transposed tile
BMSP_HD uint32_t tile_byte(uint64_t bmp, int i) { return (uint32_t)(bmp >> (56 - 8 * i)) & 0xffu; }

// OR of the eight bytes: for a normal tile bit (7-k) says "column k is non-empty";
// for a transposed tile bit (7-k) says "row k is non-empty".
BMSP_HD uint32_t tile_or_bytes(uint64_t bmp)
{
    uint64_t x = bmp | (bmp >> 32);
    x |= x >> 16;
    x |= x >> 8;
    return (uint32_t)x & 0xffu;
}

// true when the 8x8 boolean product of A (normal) and B (transposed layout) is empty.
// Equivalent to the reference's multiplication_checker (src/bmSparse_SPGEMM.cu:742-757): some (i,j) has
// rowA_i & colB_j != 0  <=>  some k is used by a column of A and by a row of B.
BMSP_HD bool tile_product_empty(uint64_t a, uint64_t bt) { return (tile_or_bytes(a) & tile_or_bytes(bt)) == 0; }

// bitmap of the boolean product C = A * B, A normal layout, B transposed layout, C normal layout.
// Same result as the reference's bmp_calculator (src/bmSparse_SPGEMM.cu:787-810), computed as eight
// rank-1 updates: for each k, (column k of A) x (row k of B).
BMSP_HD uint64_t tile_product_bmp(uint64_t a, uint64_t bt)
{
    const uint64_t lsb = 0x0101010101010101ull;
    uint64_t res = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t ak = (a >> (7 - k)) & lsb;   // byte i == 1 when A(i,k) stored
        uint64_t bk = (bt >> (7 - k)) & lsb;  // byte j == 1 when B(k,j) stored
        // gather byte j's flag into bit (7-j) of one byte (byte 0 is the most significant byte)
        uint64_t row = (bk * 0x0102040810204080ull) >> 56;
        res |= (ak * 0xffull) & (row * lsb);
    }
    return res;
}

// 8x8 bit-matrix transpose of a tile bitmap (byte r of the word, top byte first, = row r; bit 7-c of the byte = column c): the row-major
// form of a tile stored column-major (a `transposed` build, src/bmSpMatrix.cu:85-98) and vice versa.
BMSP_HD uint64_t tile_transpose(uint64_t x)
{
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}

// bmp_calculator (src/bmSparse_SPGEMM.cu:787-810) from A's bitmap (row-major) and B's ROW-major bitmap br = tile_transpose(bt):
// C = OR_k (rows i with A(i,k)) x (row k of B); equals tile_product_bmp(a, bt).  On the device: per k, bit (7-k) of every byte of A is
// moved to the byte's sign bit and the sign bits are spread over their bytes by v_perm_b32 (selectors 8..11 = sign of bytes 1 / 3 of S1,
// 1 / 3 of S0), row k of B is replicated by another v_perm_b32: 11 full-rate instructions per k against ~25 (two 64-bit multiplies) for
// tile_product_bmp -- the row-merge passes evaluate it for every surviving pair (bmsp_selftest_tile_product checks it on the hardware).
BMSP_HD uint64_t tile_product_rm(uint64_t a, uint64_t br)
{
    const uint32_t ah = (uint32_t)(a >> 32), al = (uint32_t)a, bh = (uint32_t)(br >> 32), bl = (uint32_t)br;
    uint32_t ch = 0, cl = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t sh = ah << k, sl = al << k;
        const uint32_t mh = __builtin_amdgcn_perm(sh, sh << 8, 0x0B090A08u);
        const uint32_t ml = __builtin_amdgcn_perm(sl, sl << 8, 0x0B090A08u);
        const uint32_t rk = __builtin_amdgcn_perm(0u, k < 4 ? bh : bl, 0x01010101u * (uint32_t)(3 - (k & 3)));
#else
        const uint32_t mh = ((ah >> (7 - k)) & 0x01010101u) * 0xffu, ml = ((al >> (7 - k)) & 0x01010101u) * 0xffu;
        const uint32_t rk = (((k < 4 ? bh : bl) >> (8 * (3 - (k & 3)))) & 0xffu) * 0x01010101u;
#endif
        ch |= mh & rk;
        cl |= ml & rk;
    }
    return ((uint64_t)ch << 32) | (uint64_t)cl;
}

#if defined(__HIPCC__)
// bmp_calculator for one lane's B tile (row-major bitmap, halves bh : bl) against an A tile that is the same for the whole wave: ah : al
// its bitmap, cols = tile_or_bytes(a) (bit 7-k: column k in use), all three in scalar registers.  C = OR over the columns k the A tile
// uses of (rows i with A(i,k)) x (row k of B): the masks are scalar arithmetic, a lane spends one v_perm_b32 and two v_and_or_b32 per k.
__device__ __forceinline__ uint64_t tile_product_scalar_a(uint32_t ah, uint32_t al, uint32_t cols, uint32_t bh, uint32_t bl)
{
    uint32_t ch = 0, cl = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if ((cols >> (7 - k)) & 1u) {  // (wave-uniform)
            const uint32_t mh = ((ah >> (7 - k)) & 0x01010101u) * 0xffu, ml = ((al >> (7 - k)) & 0x01010101u) * 0xffu;
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t rk = __builtin_amdgcn_perm(0u, k < 4 ? bh : bl, 0x01010101u * (uint32_t)(3 - (k & 3)));
#else
            const uint32_t rk = (((k < 4 ? bh : bl) >> (8 * (3 - (k & 3)))) & 0xffu) * 0x01010101u;  // (the host pass of the compiler only needs it to parse)
#endif
            ch |= mh & rk;
            cl |= ml & rk;
        }
    }
    return ((uint64_t)ch << 32) | (uint64_t)cl;
}
#endif

// double -> IEEE binary16 bits, round to nearest even in one step (what `(half)double` does in the
// reference, src/bmSpMatrix.cu:141; include/half.hpp:373-374).
BMSP_HD uint16_t f64_to_f16_bits(double x)
{
    union { double d; uint64_t u; } cv;
    cv.d = x;
    uint64_t u = cv.u;
    uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
    int64_t exp = (int64_t)((u >> 52) & 0x7ff);
    uint64_t man = u & 0xfffffffffffffull;
    if (exp == 0x7ff) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (uint16_t)(man >> 42)) : 0));
    if (exp == 0) return sign;
    int64_t e = exp - 1023;
    if (e > 15) return (uint16_t)(sign | 0x7c00u);
    uint64_t full = man | (1ull << 52);
    int shift = 42;
    uint32_t hexp = 0;
    if (e >= -14) hexp = (uint32_t)(e + 15);
    else {
        shift = 42 + (int)(-14 - e);
        if (shift > 54) return sign;
    }
    uint64_t kept = full >> shift;
    uint64_t rem = full & ((1ull << shift) - 1);
    uint64_t halfway = 1ull << (shift - 1);
    if (rem > halfway || (rem == halfway && (kept & 1))) kept++;
    uint32_t bits = hexp ? ((hexp - 1) << 10) + (uint32_t)kept : (uint32_t)kept;
    if (bits >= 0x7c00u) bits = 0x7c00u;
    return (uint16_t)(sign | bits);
}

BMSP_HD int ceil_log2_u64(uint64_t n)
{  // number of bits needed to represent values in [0, n)
    int b = 0;
    while (b < 64 && (n > (1ull << b))) b++;
    return b;
}

}  // namespace bmsp
#endif
