// ref_half_check.cpp -- test infrastructure.  Compiled against the reference's include/half.hpp
// (half_float 2.2.0, round-to-nearest) where it lies under /root/reference; emits, for every input
// double on stdin (one hex-float or decimal per line), the fp16 bit pattern half.hpp produces, and
// with "mul a b" lines the fp16 product of two fp16 operands.  Used by tests/golden/make_golden.py
// to generate tests/golden/half_rounding.json, which pins oracle/bmsp_oracle.c's fp16 helpers.
#include <half.hpp>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>

static uint16_t bits(half_float::half h) { uint16_t b; std::memcpy(&b, &h, 2); return b; }

int main() {
    char line[256];
    while (std::fgets(line, sizeof line, stdin)) {
        if (!std::strncmp(line, "mul", 3)) {
            unsigned a, b;
            if (std::sscanf(line + 3, "%x %x", &a, &b) != 2) continue;
            half_float::half ha, hb; uint16_t ua = (uint16_t)a, ub = (uint16_t)b;
            std::memcpy(&ha, &ua, 2); std::memcpy(&hb, &ub, 2);
            half_float::half p = ha * hb;
            std::printf("%04x\n", bits(p));
        } else {
            double d = std::strtod(line, nullptr);
            half_float::half h = half_float::half_cast<half_float::half, std::round_to_nearest>(d);
            std::printf("%04x\n", bits(h));
        }
    }
    return 0;
}
