// host check of the bit-level helpers the HIP kernels use (csrc/bmsp_bits.h) against brute force.
#include "../bmsparse-spgemm-spmv_amd/csrc/bmsp_bits.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

static uint64_t brute_product(uint64_t a, uint64_t bt)
{
    uint64_t r = 0;
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
            for (int k = 0; k < 8; k++)
                if (((a >> (63 - (i * 8 + k))) & 1) && ((bt >> (63 - (j * 8 + k))) & 1)) r |= 1ull << (63 - (i * 8 + j));
    return r;
}

int main()
{
    std::mt19937_64 rng(7);
    for (int it = 0; it < 200000; it++) {
        uint64_t a = rng(), b = rng();
        int da = it % 5, db = (it / 5) % 5;  // thin the bitmaps out
        for (int k = 0; k < da; k++) a &= rng();
        for (int k = 0; k < db; k++) b &= rng();
        uint64_t e = brute_product(a, b);
        if (bmsp::tile_product_bmp(a, b) != e) { std::printf("FAIL product %016llx %016llx\n", (unsigned long long)a, (unsigned long long)b); return 1; }
        if (bmsp::tile_product_empty(a, b) != (e == 0)) { std::printf("FAIL empty\n"); return 1; }
        if (bmsp::tile_product_rm(a, bmsp::tile_transpose(b)) != e) { std::printf("FAIL row-major product\n"); return 1; }
        if (bmsp::tile_transpose(bmsp::tile_transpose(a)) != a) { std::printf("FAIL transpose\n"); return 1; }
        for (int q = 0; q < 4; q++) {  // transpose: position 8r + c <-> 8c + r
            const int r = (int)(rng() % 8), c = (int)(rng() % 8);
            if (bmsp::tile_has(bmsp::tile_transpose(a), 8 * c + r) != bmsp::tile_has(a, 8 * r + c)) { std::printf("FAIL transpose bit\n"); return 1; }
        }
        int p = (int)(rng() % 64);
        int rank = 0;
        for (int q = 0; q < p; q++) rank += (int)((a >> (63 - q)) & 1);
        if (bmsp::tile_rank(a, p) != rank) { std::printf("FAIL rank\n"); return 1; }
        uint32_t orb = 0;
        for (int i = 0; i < 8; i++) orb |= bmsp::tile_byte(a, i);
        if (bmsp::tile_or_bytes(a) != orb) { std::printf("FAIL or_bytes\n"); return 1; }
    }
    // f64 -> f16 against the compiler's own conversion where available (gcc >= 12 / clang have _Float16)
#if defined(__FLT16_MANT_DIG__)
    for (int it = 0; it < 2000000; it++) {
        uint64_t u = rng();
        double d;
        if (it & 1) { // restrict exponent to the interesting range
            int e = (int)(rng() % 48) - 30;
            d = std::ldexp(1.0 + (double)(rng() >> 11) / 9007199254740992.0, e) * ((u & 1) ? -1 : 1);
        } else std::memcpy(&d, &u, 8);
        if (d != d) continue;
        _Float16 h = (_Float16)d;
        uint16_t hb; std::memcpy(&hb, &h, 2);
        if (bmsp::f64_to_f16_bits(d) != hb) { std::printf("FAIL f16 %a -> %04x vs %04x\n", d, bmsp::f64_to_f16_bits(d), hb); return 1; }
    }
    std::printf("OK (with _Float16 cross-check)\n");
#else
    std::printf("OK\n");
#endif
    for (uint64_t n : {0ull, 1ull, 2ull, 3ull, 8ull, 9ull, 1ull << 20, (1ull << 20) + 1})
        if (n > 1 && !((1ull << bmsp::ceil_log2_u64(n)) >= n && (1ull << (bmsp::ceil_log2_u64(n) - 1)) < n)) { std::printf("FAIL log2\n"); return 1; }
    return 0;
}
