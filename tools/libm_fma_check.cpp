// tools/libm_fma_check.cpp -- which multiply-adds of logf's double polynomial may be fused without changing a single float result?
// Domain: every normalised mantissa log10f hands to logf (exponent fields 0x7e and 0x7f: 2^24 values).  pv_libm.h fuses all five on
// the strength of this check.   g++ -O2 -ffp-contract=off -mfma -std=c++17 tools/libm_fma_check.cpp -o /tmp/libm_fma_check && /tmp/libm_fma_check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
static const double T[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
static inline float fb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
template <int MASK>
static float lm(uint32_t ix, int* kkOut) {
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int kk = (int)tmp >> 23;
    *kkOut = kk;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)fb(iz);
    volatile double t;
    double r, y0, y;
    if (MASK & 1) r = fma(z, invc, -1.0); else { t = z * invc; r = t - 1.0; }
    if (MASK & 2) y0 = fma((double)kk, 0x1.62e42fefa39efp-1, logc); else { t = (double)kk * 0x1.62e42fefa39efp-1; y0 = logc + t; }
    const double r2 = r * r;
    if (MASK & 4) y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2); else { t = 0x1.5575b0be00b6ap-2 * r; y = t + -0x1.ffffef20a4123p-2; }
    if (MASK & 8) y = fma(-0x1.00ea348b88334p-2, r2, y); else { t = -0x1.00ea348b88334p-2 * r2; y = t + y; }
    volatile double u = y0 + r;
    if (MASK & 16) y = fma(y, r2, u); else { t = y * r2; y = t + u; }
    return (float)y;
}
template <int MASK>
static void run() {
    long bad = 0; int kmin = 9, kmax = -9;
    for (uint32_t ix = 0x3f000000u; ix < 0x40000000u; ++ix) {
        int k0, k1;
        const float a = lm<0>(ix, &k0), b = lm<MASK>(ix, &k1);
        if (bf(a) != bf(b)) ++bad;
        if (k0 < kmin) kmin = k0; if (k0 > kmax) kmax = k0;
    }
    printf("fused mask %2d: %ld of %u mantissas differ (kk in [%d, %d])\n", MASK, bad, 0x40000000u - 0x3f000000u, kmin, kmax);
}
int main() {
    run<1>(); run<2>(); run<4>(); run<8>(); run<16>(); run<31>(); run<30>(); run<28>();
    return 0;
}
