#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
static long long ref_fix32(float v) {   // the definition in use: __double2ll_rn(clamp((double)v * 2^32))
    double s = (double)v * 4294967296.0;
    s = s > 9.0e18 ? 9.0e18 : s;
    s = s < -9.0e18 ? -9.0e18 : s;
    return llrint(s);   // round to nearest even (default rounding mode)
}
static long long int_fix32(float v) {
    uint32_t u; std::memcpy(&u, &v, 4);
    const int e = (int)((u >> 23) & 0xffu);
    const unsigned long long m = e ? ((u & 0x7fffffu) | 0x800000u) : (u & 0x7fffffu);
    const int sh = (e ? e - 127 : -126) + 9;            // value * 2^32 = m * 2^sh
    unsigned long long mag;
    if (sh >= 0) {
        mag = sh > 39 ? 9000000000000000000ull : (m << sh);
        if (mag > 9000000000000000000ull) mag = 9000000000000000000ull;
    } else {
        const int r = -sh;
        if (r > 25) mag = 0;
        else {
            const unsigned long long q = m >> r, rem = m & ((1ull << r) - 1), half = 1ull << (r - 1);
            mag = q + ((rem > half || (rem == half && (q & 1ull))) ? 1ull : 0ull);
        }
    }
    return (u >> 31) ? -(long long)mag : (long long)mag;
}
int main() {
    std::mt19937_64 g(1);
    long bad = 0;
    for (long i = 0; i < 400000000L; i++) {
        uint32_t u = (uint32_t)g();
        if (((u >> 23) & 0xff) == 0xff) continue;
        float v; std::memcpy(&v, &u, 4);
        if (ref_fix32(v) != int_fix32(v)) { if (bad < 5) std::printf("mismatch %a: %lld vs %lld\n", v, ref_fix32(v), int_fix32(v)); bad++; }
    }
    // exhaustive over the exponents around the rounding region and the clamp
    for (int e = 80; e <= 200; e++) for (uint32_t mant = 0; mant < (1u << 23); mant += 1) {
        if ((mant & 0xfff) && e > 110 && e < 150) continue;   // thin out where nothing rounds
        for (int sgn = 0; sgn < 2; sgn++) {
            uint32_t u = ((uint32_t)sgn << 31) | ((uint32_t)e << 23) | mant; float v; std::memcpy(&v, &u, 4);
            if (ref_fix32(v) != int_fix32(v)) { if (bad < 5) std::printf("mismatch %a\n", v); bad++; }
        }
    }
    std::printf("bad=%ld\n", bad);
    return bad != 0;
}
